"""One autograd node for a whole training render on the MI355X fp32 path (the `self.training` branch of
NeRFRenderer.run_cuda, nerf/renderer.py:318-353 of the reference):

    near_far_from_aabb -> march_rays_train -> network (fused_network) -> composite_rays_train
    -> image + (1 - weights_sum) * bg_color,  depth normalised to [0,1]

Every kernel is the library's; what the node removes is the autograd / dispatcher traffic between them (four Function
nodes, a dozen elementwise launches), which is what bounds a 4096-ray step once the kernels are fast.  Values are those
of the op-by-op route; `tests/test_gpu_training.py` compares the two.
"""
import contextlib

import torch
from torch.autograd import Function

from . import _lib as L
from . import fused_network as fnet
from . import raymarching as _rm
from .backends import _raymarching as _rb

ENABLED = True


def supported(model, rays_o, rays_d, bg_color, dt_gamma):
    if not (ENABLED and fnet.ENABLED and model.training and model.cuda_ray and model.bg_radius <= 0
            and rays_o.is_cuda and rays_o.dtype == torch.float32 and rays_d.dtype == torch.float32
            and not torch.is_autocast_enabled() and not rays_o.requires_grad and not rays_d.requires_grad
            and _rm._DEVICE == "cuda" and torch.is_grad_enabled()):
        return False
    if isinstance(bg_color, torch.Tensor):
        # the composite kernels take a scalar, one RGB or one RGB per ray, fp32 and dense; anything else the reference's
        # `image + (1 - ws)[:, None] * bg_color` would broadcast goes the op-by-op route
        if (bg_color.requires_grad or not bg_color.is_cuda or bg_color.dtype != torch.float32
                or not bg_color.is_contiguous() or bg_color.numel() not in (1, 3, 3 * rays_o.view(-1, 3).shape[0])):
            return False
    probe = rays_o.view(-1, 3)
    return fnet.supported(model, probe, probe)


_BOX_FOR = {}        # device index -> key of the bitfield the library's occupied box was last computed for
_BOX_TOKENS = iter(range(1, 1 << 62))


def occupied_box_flag(model):
    """-> the marcher flag (4) that enables the occupied-box test, after making sure the library's box belongs to this
    model's current bitfield: recomputed on the current stream when another model used the box last, or when this
    model's bitfield changed storage, torch version (copy_, load_state_dict) or the package's raw-write epoch
    (packbits, update_extra_state).  The model carries a unique token, so a new model whose bitfield happens to land
    on a freed one's address is never mistaken for it."""
    bf = model._buffers["density_bitfield"]
    token = model.__dict__.get("_occ_box_token")
    if token is None:
        token = model.__dict__["_occ_box_token"] = next(_BOX_TOKENS)
    key = (token, bf.data_ptr(), bf._version, _rm.BITFIELD_EPOCH[0], int(model.cascade), int(model.grid_size),
           float(model.bound))
    if _BOX_FOR.get(bf.device.index) != key:
        _rb.occupied_box_update(bf, model.cascade, model.grid_size, model.bound)
        _BOX_FOR[bf.device.index] = key
    return 4


class _Stage(dict):
    """A march stage's buffers.  When its kernels were issued on a launch stream, the buffers (allocated from the
    consumer stream's pool, see march_stage) must not be recycled before that stream is done with them: a stage that
    is consumed hands its "ready" event to the consumer, which waits for it; one that is dropped unconsumed -- the
    rays never came, the bitfield changed, the model was deleted -- makes the current stream wait here, before its
    tensors go back to the pool."""

    def __del__(self):
        ready = self.get("ready")
        if ready is not None:
            try:
                torch.cuda.current_stream().wait_event(ready)
            except Exception:           # interpreter shutdown
                pass


def march_stage(model, rays_o, rays_d, counter, mean_count, perturb, force_all_rays, dt_gamma, max_steps,
                background=False, defer=False, launch_stream=None, after_signal=False):
    """near_far_from_aabb + march_rays_train: everything of a training render that does not read the parameters.
    Returns the sample buffers; a data-parallel harness runs it for the NEXT batch while the gradient all-reduce of
    the current step is in flight (TrainHarness.prefetch_march).

    While no sample budget exists (`mean_count <= 0`: the first update_extra_state window; or `force_all_rays`) the
    reference's wrapper allocates and zero-fills N * max_steps rows, marches, reads the count back and crops
    (raymarching/raymarching.py:195-228).  The marcher here counts before it writes, so only the count pass runs first;
    its total comes back through pinned memory and the write pass goes into buffers of exactly the cropped size
    (same rows, same `rays` / `counter`: the drop rule still sees min(cropped size, N * max_steps)).  With `defer` the
    read-back is left to finish_march(): a stage issued ahead of its step has long finished by then, so the wait is
    free and the cold window runs the same launch sequence as the budgeted steady state -- write pass included, into
    rows reserved from the previous render's count (kept when the count that comes back fits).

    `launch_stream`: the kernels are issued on that stream (ordered after everything queued so far on the current
    one) while every buffer is allocated HERE, from the current stream's pool: the consumer is the current stream,
    which waits for the stage's event before it reads and frees in its own order -- nothing has to be
    `record_stream`-ed, and no buffer's release puts an event record (a barrier packet, ~15 us of idle queue) between
    two kernels of the training step."""
    N = rays_o.shape[0]
    dev = rays_o.device
    budgeted = not (force_all_rays or mean_count <= 0)
    nears = torch.empty(N, dtype=torch.float32, device=dev)
    fars = torch.empty(N, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    if budgeted:
        M = mean_count + (128 - mean_count % 128)            # raymarching.py:186-189 (align = 128)
        # budgeted buffers: the write pass zero-fills the rows no ray writes, so no torch.zeros passes over them
        xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
        dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
        deltas = torch.empty(M, 2, dtype=torch.float32, device=dev)
    spec_rows = 0
    if not budgeted and defer and not force_all_rays:
        # no budget yet, but the previous render's count is known: its rows + 1/8 are reserved here and the write pass
        # follows the count pass on the same stream, unseen by the host; finish_march() keeps the result when the count
        # it reads back fits (consecutive batches differ by a few per cent) and repeats the write pass when not
        last = int(getattr(model, "_cold_rows", 0))
        if last > 0:
            spec_rows = min(last + last // 8 + (128 - (last + last // 8) % 128), N * max_steps)
            xyzs = torch.empty(spec_rows, 3, dtype=torch.float32, device=dev)
            dirs = torch.empty(spec_rows, 3, dtype=torch.float32, device=dev)
            deltas = torch.empty(spec_rows, 2, dtype=torch.float32, device=dev)
    bufs = model._buffers                            # (nn.Module.__getattr__ is a slow path)
    bitfield = bufs["density_bitfield"]
    pre = _Stage(nears=nears, fars=fars, rays=rays, counter=counter)
    box = occupied_box_flag(model)
    if launch_stream is not None:
        if after_signal == "ordered":
            pass        # a stage issued just before on the same launch stream has waited already
        elif after_signal:
            # the current stream's last launch (the MLP backward's reduce) carries a completion signal: waiting for it
            # orders the stage after everything queued so far without an event record in the current stream
            L.check(L.lib().enerf_stream_wait_mlp32_signal(launch_stream.cuda_stream), "stream_wait_mlp32_signal")
        else:
            launch_stream.wait_stream(torch.cuda.current_stream())
    with (torch.cuda.stream(launch_stream) if launch_stream is not None else contextlib.nullcontext()):
        _rb.near_far_from_aabb(rays_o, rays_d, bufs["aabb_train"], N, model.min_near, nears, fars)
        if not budgeted:
            _rb.march_rays_train_count(rays_o, rays_d, bitfield, model.bound, dt_gamma, max_steps, N,
                                       model.cascade, model.grid_size, nears, fars, rays, counter, perturb,
                                       box | (2 if background else 0) | 8)
            total = torch.empty(2, dtype=torch.int32, pin_memory=True)
            total.copy_(counter, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            pre["pending"] = (done, total, rays_o, rays_d, perturb, dt_gamma, max_steps)
            if spec_rows:
                _rb.march_rays_train_write(rays_o, rays_d, bitfield, model.bound, dt_gamma, max_steps, N, model.cascade,
                                           model.grid_size, spec_rows, nears, fars, xyzs, dirs, deltas, rays, counter,
                                           perturb, 1)
                pre["speculative"] = (spec_rows, xyzs, dirs, deltas)
        else:
            _rb.march_rays_train_ex(rays_o, rays_d, bitfield, model.bound, dt_gamma, max_steps, N,
                                    model.cascade, model.grid_size, M, nears, fars, xyzs, dirs, deltas, rays, counter,
                                    perturb, box | (3 if background else 1) | 8)
            pre.update(xyzs=xyzs, dirs=dirs, deltas=deltas, M=M)
        if launch_stream is not None:
            pre["ready"] = torch.cuda.Event()
            pre["ready"].record(launch_stream)
            model._last_march_event = (pre["ready"], launch_stream)
    if not budgeted and not defer:
        finish_march(model, pre)
    return pre


def finish_march(model, pre):
    """Second half of an unbudgeted march_stage, on the current stream: wait for the count (host side; the pinned
    total is the only thing read), size the sample buffers from it, run the write pass."""
    pending = pre.pop("pending", None)
    if pending is None:
        return pre
    done, total, rays_o, rays_d, perturb, dt_gamma, max_steps = pending
    done.synchronize()
    N = rays_o.shape[0]
    dev = rays_o.device
    m = int(total[0])
    M = min(m + (128 - m % 128), N * max_steps)              # the reference's crop of its N * max_steps buffers
    model._cold_rows = M
    spec = pre.pop("speculative", None)
    if spec is not None and M <= spec[0] and m < N * max_steps:
        # the write pass already ran behind the count pass into buffers at least this large: the rows are the same
        # (nothing is dropped below N * max_steps samples, rows past the count are zero-filled either way)
        pre.update(xyzs=spec[1][:M], dirs=spec[2][:M], deltas=spec[3][:M], M=M)
        return pre
    xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.empty(M, 2, dtype=torch.float32, device=dev)
    _rb.march_rays_train_write(rays_o, rays_d, model._buffers["density_bitfield"], model.bound, dt_gamma, max_steps, N,
                               model.cascade, model.grid_size, M, pre["nears"], pre["fars"], xyzs, dirs, deltas,
                               pre["rays"], pre["counter"], perturb, 1)
    pre.update(xyzs=xyzs, dirs=dirs, deltas=deltas, M=M)
    return pre


class _FusedRenderTrain(Function):
    @staticmethod
    def forward(ctx, rays_o, rays_d, model, bg_color, counter, mean_count, perturb, force_all_rays, dt_gamma,
                max_steps, pre, embeddings, *weights):
        N = rays_o.shape[0]
        dev = rays_o.device
        if pre is None:
            pre = march_stage(model, rays_o, rays_d, counter, mean_count, perturb, force_all_rays, dt_gamma, max_steps)
        nears, fars, xyzs, dirs, deltas, rays, M = (pre[k] for k in ("nears", "fars", "xyzs", "dirs", "deltas", "rays",
                                                                     "M"))

        train = any(p.requires_grad for p in (embeddings,) + weights)
        sigma, rgb, sv = fnet.nerf_forward(xyzs, dirs, fnet.network_cfg(model), train, embeddings, fnet.encoder_offsets(model),
                                           *weights)
        scale = float(model.density_scale)
        sigmas = sigma if scale == 1.0 else sigma * scale
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        depth = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        # composite + `image + (1 - weights_sum).unsqueeze(-1) * bg_color` in one launch
        _rb.composite_rays_train_forward_blend(sigmas, rgb, deltas, rays, M, N, weights_sum, depth, image, bg_color,
                                               out_image)
        out_depth = torch.clamp(depth - nears, min=0) / (fars - nears)
        if train:
            ctx.sv = sv
            ctx.rest = (sigmas, rgb, deltas, rays, weights_sum, image, bg_color, M, N, scale)
        ctx.mark_non_differentiable(out_depth)
        return out_depth, out_image

    @staticmethod
    def backward(ctx, _g_depth, g_image):
        sigmas, rgb, deltas, rays, weights_sum, image, bg_color, M, N, scale = ctx.rest
        g_image = g_image.reshape(N, 3).float().contiguous()
        # out_image = image + (1 - weights_sum)[:, None] * bg  ->  d/d(weights_sum) = -(g . bg)
        g_ws = -(g_image * bg_color).sum(-1) if isinstance(bg_color, torch.Tensor) else -(g_image.sum(-1) * bg_color)
        g_ws = g_ws.reshape(N).contiguous()
        g_sigmas = torch.zeros_like(sigmas)
        g_rgbs = torch.zeros_like(rgb)
        _rb.composite_rays_train_backward(g_ws, g_image, sigmas, rgb, deltas, rays, weights_sum, image, M, N, g_sigmas,
                                          g_rgbs)
        g = fnet.nerf_backward(ctx.sv, g_sigmas, g_rgbs, sigma_scale=scale)
        return (None,) * 11 + g


def _budget(model):
    mean_count = int(model.mean_count)
    quantum = int(getattr(model, "sample_budget_quantum", 0))
    if quantum > 0 and mean_count > 0:
        mean_count = (mean_count + quantum - 1) // quantum * quantum       # fewer distinct shapes (never fewer slots)
    return mean_count


def _next_counter(model):
    """The render's slot of the step-counter ring (raymarching.py:197-199).  Not zeroed here: march_stage tells the
    marcher to start from (0, 0) (flag bit 3) -- a fill launch per step in the middle of the training stream, ~14 us of
    it with the idle queue around it, for eight bytes."""
    # graph replay needs the counter at a fixed address; the harness copies it into the step_counter ring afterwards
    model._last_march_event = None              # (set again by whoever marches on a side stream: see early_mean_count)
    counter = getattr(model, "graph_counter", None)
    if counter is None:
        model.last_counter_slot = model.local_step % 16
        counter = model._buffers["step_counter"][model.last_counter_slot]
        model.local_step += 1
    return counter


def early_mean_count(model):
    """The sample budget update_extra_state is about to compute (mean of the window's step counters,
    nerf/renderer.py:550-552), read BEFORE the update is queued: -> (mean_count or None, total_step), or None when it
    cannot be had without waiting for the training stream.

    update_extra_state's read-back sits behind ~2 ms of sweep kernels, and the host -- which needs the budget to size and
    queue the step that follows -- used to wait for it with an empty queue behind (0.15-0.3 ms of idle device per
    update).  The window's counters are final much earlier: every march of the window has been issued, and when the last
    one ran on the side stream (the normal case: each step marches its successor's rays there) a copy queued behind it
    on that stream completes while the training stream is still a step away from the update.  The copy waits for that
    march only; the value is the same sum of the same ring slots the update kernel forms."""
    total_step = min(16, int(model.local_step))
    if total_step == 0:
        return (None, 0)                           # nothing to average: mean_count keeps its value
    staged = model.__dict__.pop("_ring_copy", None)
    if staged is not None and staged[1] == int(model.local_step) and not getattr(model, "_premarched", None):
        # stage_ring_copy: the ring was copied on the training stream in front of the window's last step (whose march-free
        # call has long been queued behind it); nothing has marched since
        staged[0].synchronize()
        counted = int(model._ring_host[:total_step, 0].to(torch.int64).sum())
        return (int(counted / total_step), total_step)
    last = getattr(model, "_last_march_event", None)
    stash = getattr(model, "_premarched", None)
    if last is None or stash:                      # the last march ran on the training stream / an unconsumed stage
        return None
    _ready, stream = last
    host = getattr(model, "_ring_host", None)
    if host is None:
        host = model._ring_host = torch.empty(16, 2, dtype=torch.int32, pin_memory=True)
    with torch.cuda.stream(stream):
        host.copy_(model._buffers["step_counter"], non_blocking=True)
        done = torch.cuda.Event()
        done.record(stream)
    done.synchronize()
    counted = int(host[:total_step, 0].to(torch.int64).sum())
    return (int(counted / total_step), total_step)


def stage_ring_copy(model):
    """The window's step counters to pinned host memory, on the CURRENT stream, now.  Where the marches run on the training
    stream itself (the one-call step carries the next batch's march in its optimizer launch: csrc/train_step.hip), early_mean_count
    has no side stream to read the ring behind: TrainHarness calls this in front of the window's LAST step -- which marches
    nothing, the update that follows voids any stage -- so that the copy is complete a whole step before the update needs it."""
    host = getattr(model, "_ring_host", None)
    if host is None:
        host = model._ring_host = torch.empty(16, 2, dtype=torch.int32, pin_memory=True)
    host.copy_(model._buffers["step_counter"], non_blocking=True)
    done = torch.cuda.Event()
    done.record()
    model._ring_copy = (done, int(model.local_step))


def prefetch_march(model, rays_o, rays_d, perturb=True, dt_gamma=0, max_steps=1024, stream=None, background=True,
                   after_signal=False):
    """Run the parameter-independent stage of the NEXT training render now (e.g. under a gradient all-reduce).  The
    result is picked up by the next render_train call on the same ray tensors; anything that changes what the stage
    reads (update_extra_state: bitfield, sample budget) must not happen in between -- the caller's responsibility.

    With `stream` the stage is issued on that HIP stream, ordered after everything queued so far on the current one
    (so it never overlaps a march of the current stream: the marcher's chunk-log workspace is shared), and runs
    concurrently with what the current stream does next (`background`: beside compute kernels, where the count pass
    runs one wavefront per SIMD; False beside collectives, where nothing competes for registers) -- the marcher is
    latency / VALU bound and hides under the
    MFMA- and HBM-bound backward kernels.  The consumer waits on the stage's event."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    key = (rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], bool(perturb), float(dt_gamma), int(max_steps))
    stash = getattr(model, "_premarched", None)
    if not isinstance(stash, dict):
        stash = model._premarched = {}
    if any("pending" in p for p in stash.values()):
        # an unbudgeted stage is waiting for its write pass, and the marcher's chunk log (count -> write) is one
        # per-process buffer: a second count now would overwrite it.  The second render of an event step marches inline.
        return
    if stream is None:
        pre = march_stage(model, rays_o, rays_d, _next_counter(model), _budget(model), bool(perturb), False,
                          float(dt_gamma), int(max_steps), defer=True)
    else:
        pre = march_stage(model, rays_o, rays_d, _next_counter(model), _budget(model), bool(perturb), False,
                          float(dt_gamma), int(max_steps), background=background, defer=True, launch_stream=stream,
                          after_signal=after_signal)
    pre["slot"] = getattr(model, "last_counter_slot", None)
    stash[key] = pre                                 # (an event step stashes both of its renders)


def premarch_count(model, rays_o, rays_d, perturb=True, dt_gamma=0, max_steps=1024):
    """near/far + the marcher's count pass and scan for the coming training render, on the current stream, before the
    sample budget is known on the host: TrainHarness queues it between density_update.update_begin and update_end, so
    the device has work while the host waits for the update's read-back (neither pass reads the budget: rays, counter
    and chunk log do not depend on M).  The render picks the stage up and runs the write pass with the budget it then
    has -- the same launches as march_rays_train_ex, in the same order."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, dev = rays_o.shape[0], rays_o.device
    key = (rays_o.data_ptr(), rays_d.data_ptr(), N, bool(perturb), float(dt_gamma), int(max_steps))
    nears = torch.empty(N, dtype=torch.float32, device=dev)
    fars = torch.empty(N, dtype=torch.float32, device=dev)
    rays = torch.empty(N, 3, dtype=torch.int32, device=dev)
    counter = _next_counter(model)
    bufs = model._buffers
    _rb.near_far_from_aabb(rays_o, rays_d, bufs["aabb_train"], N, model.min_near, nears, fars)
    _rb.march_rays_train_count(rays_o, rays_d, bufs["density_bitfield"], model.bound, float(dt_gamma), int(max_steps), N,
                               model.cascade, model.grid_size, nears, fars, rays, counter, bool(perturb),
                               occupied_box_flag(model) | 8)
    model._premarched = {key: dict(nears=nears, fars=fars, rays=rays, counter=counter,
                                   counted=(rays_o, rays_d, bool(perturb), float(dt_gamma), int(max_steps)),
                                   slot=getattr(model, "last_counter_slot", None))}


def _finish_counted(model, pre):
    """Write pass of a premarch_count stage, with the budget the host has by now (march_stage's budgeted branch)."""
    rays_o, rays_d, perturb, dt_gamma, max_steps = pre.pop("counted")
    mean_count = _budget(model)
    N, dev = rays_o.shape[0], rays_o.device
    if mean_count <= 0:                         # no budget after all (first window): the reference's crop of N * max_steps
        m = int(pre["counter"][0].item())
        M = min(m + (128 - m % 128), N * max_steps)
        model._cold_rows = M
    else:
        M = mean_count + (128 - mean_count % 128)
    xyzs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(M, 3, dtype=torch.float32, device=dev)
    deltas = torch.empty(M, 2, dtype=torch.float32, device=dev)
    _rb.march_rays_train_write(rays_o, rays_d, model._buffers["density_bitfield"], model.bound, dt_gamma, max_steps, N,
                               model.cascade, model.grid_size, M, pre["nears"], pre["fars"], xyzs, dirs, deltas,
                               pre["rays"], pre["counter"], perturb, 1)
    pre.update(xyzs=xyzs, dirs=dirs, deltas=deltas, M=M)
    return pre


def render_train(model, rays_o, rays_d, bg_color, perturb, force_all_rays, dt_gamma, max_steps):
    """-> depth [N], image [N,3] (+ the step counter bookkeeping of run_cuda)."""
    pre = _take_premarched(model, rays_o, rays_d, perturb, dt_gamma, max_steps)
    if force_all_rays:
        pre = None
    if pre is not None:
        model.rendered_counter_slot = pre["slot"]
    counter = None
    if pre is None:
        counter = _next_counter(model)
        model.rendered_counter_slot = getattr(model, "last_counter_slot", None)
    params = fnet.network_params(model)
    return _FusedRenderTrain.apply(rays_o, rays_d, model, bg_color, counter, _budget(model), bool(perturb),
                                   bool(force_all_rays), float(dt_gamma), int(max_steps), pre, *params)


def _check_cold_stage(model, pre, rays_o, rays_d, perturb, dt_gamma, max_steps):
    """A stage marched by the one-call step of the cold window into `cap` reserved rows, budgeted-style, is valid iff
    nothing was dropped, i.e. iff the count that came back (behind the march, on its stream) fits.  Waits for that count
    (the only host wait of such a step), records the rows for later reservations, repairs the stage when it does not fit:
    count, scan and ray table do not depend on the rows (same counter slot, nothing marched since), so the write pass
    alone is repeated into buffers of the exact size -- what finish_march does for a speculative write pass that did not
    fit.  -> the stage to use."""
    done, host, cap = pre.pop("cold_check")[:3]
    # the count arrives in pinned memory behind the march (an 8-byte copy on its stream): the host watches the two words
    # it pre-set to -1 instead of sleeping on the event -- the wake-up of an event wait is 10-20 us, in which the device,
    # which has nothing queued yet, idles (the cold window's steps: 11 of the driver's 20)
    # The watch is bounded by a few march times (2 ms; a 4096-ray march is ~0.13 ms) and only tried where the count is
    # stored by the kernel itself (MIRROR_COUNT): a count that arrives by an asynchronous copy is ordered by the event
    # anyway.  A watch that runs out -- pinned memory that is not host-coherent while the kernel runs
    # (HIP_HOST_COHERENT=0) -- is remembered on the model and every later step goes straight to the event.
    hv = host.numpy()
    if hv[0] < 0 or hv[1] < 0:
        if MIRROR_COUNT and not getattr(model, "_cold_watch_failed", False):
            import time as _time
            t_end = _time.perf_counter() + 0.002
            while (hv[0] < 0 or hv[1] < 0) and _time.perf_counter() < t_end:
                pass
            if hv[0] < 0 or hv[1] < 0:
                model._cold_watch_failed = True
        if hv[0] < 0 or hv[1] < 0:
            done.synchronize()
    m = int(hv[0])
    N = rays_o.shape[0]
    rows = min(m + (128 - m % 128), N * int(max_steps))
    model._cold_rows = rows
    if rows <= cap:
        return pre
    dev = rays_o.device
    xyzs = torch.empty(rows, 3, dtype=torch.float32, device=dev)
    dirs = torch.empty(rows, 3, dtype=torch.float32, device=dev)
    deltas = torch.empty(rows, 2, dtype=torch.float32, device=dev)
    _rb.march_rays_train_write(rays_o, rays_d, model._buffers["density_bitfield"], model.bound, float(dt_gamma),
                               int(max_steps), N, model.cascade, model.grid_size, rows, pre["nears"], pre["fars"], xyzs,
                               dirs, deltas, pre["rays"], pre["counter"], bool(perturb), 1)
    out = dict(pre)
    out.update(xyzs=xyzs, dirs=dirs, deltas=deltas, M=rows)
    return out


def _take_premarched(model, rays_o, rays_d, perturb, dt_gamma, max_steps, defer_cold_check=False):
    stash = getattr(model, "_premarched", None)
    if not stash:
        return None
    key = (rays_o.data_ptr(), rays_d.data_ptr(), rays_o.shape[0], bool(perturb), float(dt_gamma), int(max_steps))
    pre = stash.pop(key, None)
    if pre is None:
        stash.clear()                                # marched for rays that are not coming: drop, march afresh
        return None
    if "cold_check" in pre:
        ready = pre.pop("ready", None)
        if ready is not None:
            torch.cuda.current_stream().wait_event(ready)
        if defer_cold_check:
            return pre                               # the one-call step checks right before its call (host work first)
        return _check_cold_stage(model, pre, rays_o, rays_d, perturb, dt_gamma, max_steps)
    if "counted" in pre:
        return _finish_counted(model, pre)
    ready = pre.pop("ready", None)
    if ready is not None:                       # marched on a side stream, into buffers of this stream's pool (see
        torch.cuda.current_stream().wait_event(ready)      # march_stage): order this stream after it, nothing else
    return finish_march(model, pre)              # unbudgeted stage: the write pass runs here, sized from the count


FUSED_COMPOSITE = True        # train_step_mse: compositing forward + MSE backward as one launch
MIRROR_COUNT = True           # cold window: the march's count goes straight to pinned host memory (enerf_march_mirror_count)
import os as _os
SKIP_PADDING_ROWS = _os.environ.get("ENERF_SKIP_PADDING_ROWS", "1") != "0"      # raw renders: the MLP kernels skip the sample budget's unfilled rows (device-side count)


def render_train_raw(model, rays_o, rays_d, bg_color=1, perturb=True, dt_gamma=0, max_steps=1024, composite=True):
    """Forward half of a training render without autograd: -> (image [N,3] blended with bg_color, ctx).  Feed the
    gradient of whatever loss was computed on `image` to backward_raw(ctx, ...).  No depth.  composite=False leaves
    the compositing to backward_raw(target=...), which then runs it fused with its backward (the image is valid once
    that call is queued)."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N = rays_o.shape[0]
    dev = rays_o.device
    with torch.no_grad():
        pre = _take_premarched(model, rays_o, rays_d, perturb, dt_gamma, max_steps)
        if pre is not None:
            model.rendered_counter_slot = pre["slot"]
        else:
            counter = _next_counter(model)
            model.rendered_counter_slot = getattr(model, "last_counter_slot", None)
            pre = march_stage(model, rays_o, rays_d, counter, _budget(model), bool(perturb), False, float(dt_gamma),
                              int(max_steps))
        xyzs, dirs, deltas, rays, M = (pre[k] for k in ("xyzs", "dirs", "deltas", "rays", "M"))
        params = fnet.network_params(model)
        sigma, rgb, sv = fnet.nerf_forward(xyzs, dirs, fnet.network_cfg(model), True, params[0],
                                           fnet.encoder_offsets(model), *params[1:],
                                           valid_rows=pre["counter"] if SKIP_PADDING_ROWS else None)
        scale = float(model.density_scale)
        sigmas = sigma if scale == 1.0 else sigma * scale
        weights_sum = torch.empty(N, dtype=torch.float32, device=dev)
        image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        out_image = torch.empty(N, 3, dtype=torch.float32, device=dev)
        if isinstance(bg_color, torch.Tensor):
            bg_color = bg_color.detach().to(torch.float32).contiguous()
        if composite:
            _rb.composite_rays_train_forward_blend(sigmas, rgb, deltas, rays, M, N, weights_sum, None, image, bg_color,
                                                   out_image)
    ctx = dict(sv=sv, sigmas=sigmas, rgb=rgb, deltas=deltas, rays=rays, weights_sum=weights_sum, image=image,
               out_image=out_image, bg=bg_color, counter=pre["counter"], M=M, N=N, scale=scale, composited=composite)
    return out_image, ctx


def backward_raw(ctx, g_image=None, target=None, upstream=1.0, loss_out=None, raw=False, defer_table=0,
                 after_mlp=None):
    """Backward half.  Either g_image = d loss / d image [N,3], or target [N,3] for loss = mean((image - target)^2) *
    upstream (gradient and, with loss_out, value formed inside the composite backward).  -> the gradients of
    fused_network.network_params(model) (raw=True: (embedding gradient, flat dW), see fused_network.nerf_backward);
    the embedding gradient is None when it was added into an existing embeddings.grad."""
    N, M = ctx["N"], ctx["M"]
    with torch.no_grad():
        g_sigmas = torch.empty_like(ctx["sigmas"])
        g_rgbs = torch.empty_like(ctx["rgb"])
        if target is not None and not ctx["composited"]:
            _rb.composite_rays_train_fwd_bwd_mse(ctx["sigmas"], ctx["rgb"], ctx["deltas"], ctx["rays"], M, N,
                                                 ctx["weights_sum"], ctx["image"], ctx["bg"], ctx["out_image"],
                                                 target.contiguous().view(-1, 3), 2.0 * float(upstream) / (3 * N),
                                                 ctx["counter"], g_sigmas, g_rgbs, loss_out)
        elif target is not None:
            _rb.composite_rays_train_backward_mse(ctx["out_image"], target.contiguous().view(-1, 3),
                                                  2.0 * float(upstream) / (3 * N), ctx["bg"], ctx["counter"],
                                                  ctx["sigmas"], ctx["rgb"], ctx["deltas"], ctx["rays"],
                                                  ctx["weights_sum"], ctx["image"], M, N, g_sigmas, g_rgbs, loss_out)
        else:
            assert ctx["composited"], "render_train_raw(composite=False) needs backward_raw(target=...)"
            _rb.composite_rays_train_backward_mse(g_image.detach().to(torch.float32).contiguous().view(-1, 3), None,
                                                  1.0, ctx["bg"], ctx["counter"], ctx["sigmas"], ctx["rgb"],
                                                  ctx["deltas"], ctx["rays"], ctx["weights_sum"], ctx["image"], M, N,
                                                  g_sigmas, g_rgbs, None)
        return fnet.nerf_backward(ctx["sv"], g_sigmas, g_rgbs, sigma_scale=ctx["scale"], raw=raw, owner=True,
                                  defer_table=defer_table, after_mlp=after_mlp)


def train_step_mse(model, rays_o, rays_d, target, bg_color=1, perturb=True, dt_gamma=0, max_steps=1024, upstream=1.0,
                   after_forward=None, loss_out=None, raw=False, defer_table=False, after_mlp_backward=None):
    """Forward AND backward of a training render under loss = mean((image - target)^2) * upstream, without autograd:
    -> (image [N,3], gradients of fused_network.network_params(model) in that order; the first is None when the
    embedding gradient was added straight into the parameter's .grad).

    For loops whose loss is the reference's default (nerf/utils.py:628, MSE): the loss gradient and the blend's
    d/d(weights_sum) are formed inside the composite backward kernel, which also zero-fills what it does not write;
    depth is not computed.  What is skipped relative to render_train + autograd: the engine round trip, its
    AccumulateGrad nodes, ~18 elementwise / fill launches.  `after_forward()` is called once the forward is queued,
    `after_mlp_backward()` once both MLP backward kernels are (the hash table's backward and the optimizer follow);
    `loss_out` (a zeroed device scalar) receives the loss value from the backward kernel itself; raw=True returns the
    gradients as (embedding gradient, flat MLP dW accumulator) -- see fused_network.nerf_backward."""
    out_image, ctx = render_train_raw(model, rays_o, rays_d, bg_color, perturb, dt_gamma, max_steps,
                                      composite=not FUSED_COMPOSITE)
    if after_forward is not None:
        after_forward()                         # e.g. prefetch_march of the next batch on a side stream
    return out_image, backward_raw(ctx, target=target, upstream=upstream, loss_out=loss_out, raw=raw,
                                   defer_table=ctx["M"] if defer_table else 0, after_mlp=after_mlp_backward)


# ---------------------------------------------------------------------------------------------------------------------
# The closed-form RGB step as ONE library call (csrc/train_step.hip: enerf_train_step_mse).  The same entry points in the
# same order as train_step_mse above + FusedAdam.step_grid_table, issued by the library itself: what is left on this
# side is the bookkeeping (stage hand-over, buffers, optimizer step counts).
import ctypes as _ct

NATIVE_STEP = _os.environ.get("ENERF_NATIVE_STEP", "1") != "0"
_vp, _u32, _f32c = _ct.c_void_p, _ct.c_uint32, _ct.c_float


class _StepArgs(_ct.Structure):          # enerf_train_step_args (include/enerf_hip.h), field for field
    _fields_ = ([("struct_bytes", _u32), ("mlp_precision", _ct.c_int), ("stream", _vp), ("side_stream", _vp),
                 ("N", _u32), ("M", _u32)]
                + [(n, _vp) for n in ("xyzs", "dirs", "deltas", "rays", "counter", "target")]
                + [("bg_scalar", _f32c), ("grad_scale", _f32c), ("loss", _vp), ("embeddings", _vp), ("offsets", _vp),
                   ("level_scale_log2", _f32c), ("bound", _f32c), ("inv_two_bound", _f32c), ("base_resolution", _u32),
                   ("gridtype", _u32)]
                + [(n, _vp) for n in ("wseg_s", "wseg_c", "dwseg_s", "dwseg_c")]
                + [(n, _u32) for n in ("nh_s", "nh_c", "w0_cols_c", "out_c")]
                + [(n, _vp) for n in ("feats", "h32", "fb_s", "fb_c", "sigma", "rgb", "weights_sum", "image", "out_image",
                                      "g_sigmas", "g_rgbs", "dx32", "dfeat")]
                + [(n, _vp) for n in ("next_rays_o", "next_rays_d", "aabb", "bitfield")]
                + [("min_near", _f32c), ("dt_gamma", _f32c)]
                + [(n, _u32) for n in ("next_N", "next_M", "cascade", "grid_size", "max_steps", "perturb", "march_flags")]
                + [(n, _vp) for n in ("next_nears", "next_fars", "next_xyzs", "next_dirs", "next_deltas", "next_rays",
                                      "next_counter")]
                + [(n, _vp) for n in ("table", "table_grad", "table_m", "table_v")]
                + [(n, _f32c) for n in ("lr", "beta1", "beta2", "eps")]
                + [("table_step", _u32), ("n_small", _u32)]
                + [(n, _vp) for n in ("small_p", "small_g", "small_m", "small_v", "small_n", "small_lr", "small_step")]
                + [("flags", _u32), ("reserved", _u32)])


COLD_NATIVE = True            # the one-call step also serves the window before the first sample budget (see cold_capacity)


def cold_capacity(rows, N, max_steps, floor=0):
    """Rows reserved for a march of the cold window (no sample budget yet: nothing may be dropped), from an earlier
    render's row count: + 1/20, rounded up to 4 Ki rows, and never below what an earlier step of the window reserved
    (`floor`) -- the persistent buffers of the one-call step are keyed by these sizes, so the reservation has to settle
    after a step or two.  A march that turns out to need more gets its write pass repeated at the exact size
    (_repair_cold_stage), which costs one small launch."""
    c = rows + rows // 20
    c = (c + 4095) // 4096 * 4096
    return min(max(c, floor), N * max_steps)


def native_step_supported(model, rays_o, rays_d, opt, data_parallel=False):
    """The one-call step serves the closed-form RGB step with unit density scale and the fused optimizer: in the steady
    state (a sample budget exists) and -- one GPU -- in the cold window from its second step on, where the rows of an
    earlier render size the buffers (cold_capacity) and the count that comes back is checked before the stage is used."""
    ready = _budget(model) > 0 or (COLD_NATIVE and not data_parallel and int(getattr(model, "_cold_rows", 0)) > 0)
    return (NATIVE_STEP and ready and float(model.density_scale) == 1.0
            and hasattr(opt, "grid_table_args") and supported(model, rays_o, rays_d, 1, 0))


def _native_ctx(model, N, M, Nn, Mn, dev):
    """Everything of a native step that does not change from step to step, built once per (rays, sample budget) -- i.e.
    once per update_extra_state window: the step's scratch buffers, two sets of sample buffers for the march that runs
    ahead (the step reads one set while the side stream fills the other), the weight / gradient pointer arrays, and the
    argument struct with its constant fields filled in.  (20 torch.empty calls and ~60 struct fields per step otherwise:
    most of the host's share of a step once the launches themselves come from C.)"""
    params = fnet.network_params(model)
    emb, weights = params[0], params[1:]
    kind = fnet.kind_of(model)
    arch = fnet._ARCH[kind]
    prec = model.__dict__.get("mlp_precision")
    prec = arch["prec"] if prec is None else int(prec)
    out_c = weights[-1].shape[0] if kind == "linear" else 3
    key = (N, M, Nn, Mn, kind, prec, out_c, emb.data_ptr(), tuple(w.data_ptr() for w in weights), dev)
    ctx = model.__dict__.get("_native_ctx")
    if ctx is not None and ctx["key"] == key:
        return ctx
    import numpy as _np
    enc = model._modules["encoder"]
    bufs = model._buffers
    f32 = dict(dtype=torch.float32, device=dev)
    Mp = (M + 31) // 32 * 32
    t = dict(feats=torch.empty(16, Mp, 2, **f32), h32=torch.empty(M, 32, **f32),
             fb_s=torch.empty(arch["nh_s"], Mp, 64, **f32), fb_c=torch.empty(arch["nh_c"], Mp, 64, **f32),
             sigma=torch.empty(M, **f32), rgb=torch.empty(M, out_c, **f32), weights_sum=torch.empty(N, **f32),
             image=torch.empty(N, 3, **f32), out_image=torch.empty(N, 3, **f32), g_sigmas=torch.empty(M, **f32),
             g_rgbs=torch.empty(M, out_c, **f32), dx32=torch.empty(M, 32, **f32), dfeat=torch.empty(16, Mp, 2, **f32))
    seg_s, seg_c = fnet._weight_segments(kind, weights)
    dw, (dseg_s, dseg_c) = fnet._grad_segments(kind, dev, out_c)
    grads = fnet.unpack_weight_grads(dw, out_c, kind)
    stages = []
    if Nn:
        for _ in range(2):
            stages.append(dict(nears=torch.empty(Nn, **f32), fars=torch.empty(Nn, **f32),
                               rays=torch.empty(Nn, 3, dtype=torch.int32, device=dev), xyzs=torch.empty(Mn, 3, **f32),
                               dirs=torch.empty(Mn, 3, **f32), deltas=torch.empty(Mn, 2, **f32), M=Mn))
    a = _StepArgs()
    a.struct_bytes = _ct.sizeof(_StepArgs)
    a.mlp_precision = -1 if prec is None else prec
    a.N, a.M = N, M
    a.bg_scalar, a.grad_scale = 1.0, 2.0 / (3 * N)
    a.embeddings, a.offsets = emb.data_ptr(), enc._buffers["offsets"].data_ptr()
    a.level_scale_log2 = float(_np.log2(enc.per_level_scale))
    a.bound, a.inv_two_bound = float(model.bound), float(_np.float32(1.0) / _np.float32(2 * model.bound))
    a.base_resolution, a.gridtype = int(enc.base_resolution), int(enc.gridtype_id)
    a.wseg_s, a.wseg_c = _ct.cast(seg_s, _vp), _ct.cast(seg_c, _vp)
    a.dwseg_s, a.dwseg_c = _ct.cast(dseg_s, _vp), _ct.cast(dseg_c, _vp)
    a.nh_s, a.nh_c, a.w0_cols_c, a.out_c = arch["nh_s"], arch["nh_c"], arch["w0c"], out_c
    for name, buf in t.items():
        setattr(a, name, buf.data_ptr())
    a.aabb, a.bitfield = bufs["aabb_train"].data_ptr(), bufs["density_bitfield"].data_ptr()
    a.min_near = float(model.min_near)
    a.cascade, a.grid_size = int(model.cascade), int(model.grid_size)
    a.table = emb.data_ptr()
    ctx = dict(key=key, t=t, seg=(seg_s, seg_c), dseg=(dseg_s, dseg_c), dw=dw, grads=grads, stages=stages, flip=0, a=a,
               emb=emb, weights=weights, out_image=t["out_image"], kind=kind, out_c=out_c)
    model.__dict__["_native_ctx"] = ctx
    return ctx


def train_step_native(model, rays_o, rays_d, target, opt, next_rays=None, side_stream=None, loss_out=None, perturb=True,
                      dt_gamma=0, max_steps=1024, raw=False, defer_dp=False):
    """One closed-form RGB step (loss = mean((image - target)^2), white background) through enerf_train_step_mse:
    render of this batch (marched ahead of time when the previous step asked for it) + backward + the optimizer, and the
    march of `next_rays` = (rays_o, rays_d) on `side_stream` behind the MLP backward.  -> blended image [N,3] (a buffer
    that the next step of the same shape overwrites).  Gradients of the MLP weights are left in p.grad (views of one flat
    buffer, likewise reused), the table's dense gradient buffer comes back clean; the optimizer's step counts advance.
    raw=True (data parallel): the table's gradient is summed into the dense buffer `embeddings.grad` and NO optimizer
    runs -- -> (image, flat MLP dW buffer); the caller averages both over the ranks and steps the optimizer."""
    rays_o = rays_o.contiguous().view(-1, 3)
    rays_d = rays_d.contiguous().view(-1, 3)
    N, dev = rays_o.shape[0], rays_o.device
    with torch.no_grad():
        pre = _take_premarched(model, rays_o, rays_d, perturb, dt_gamma, max_steps, defer_cold_check=True)
        if pre is not None:
            model.rendered_counter_slot = pre["slot"]
        else:
            counter = _next_counter(model)
            model.rendered_counter_slot = getattr(model, "last_counter_slot", None)
            pre = march_stage(model, rays_o, rays_d, counter, _budget(model), bool(perturb), False, float(dt_gamma),
                              int(max_steps))
        nxt_ok = cold_next = False
        Nn = Mn = 0
        no = nd = None
        if next_rays is not None and side_stream is not None:
            no, nd = next_rays[0].contiguous().view(-1, 3), next_rays[1].contiguous().view(-1, 3)
            stash = getattr(model, "_premarched", None)
            if not isinstance(stash, dict):
                stash = model._premarched = {}
            if not any("pending" in p for p in stash.values()):
                nxt_ok = True
                Nn = no.shape[0]
                mc = _budget(model)
                cold_next = mc <= 0
                if cold_next:                    # cold window: rows reserved from the last render whose count is known
                    Mn = model._cold_cap = cold_capacity(int(model._cold_rows), Nn, int(max_steps),
                                                         int(getattr(model, "_cold_cap", 0)))
                else:
                    Mn = mc + (128 - mc % 128)
        nxt_counter = nxt_slot = None
        if nxt_ok:
            nxt_counter = _next_counter(model)
            nxt_slot = getattr(model, "last_counter_slot", None)

        def prepare(pre):
            M = pre["M"]
            ctx = _native_ctx(model, N, M, Nn, Mn, dev)
            a, emb, weights, grads = ctx["a"], ctx["emb"], ctx["weights"], ctx["grads"]
            if emb.grad is None:                    # the dense part of the table's gradient (levels too small to bin)
                emb.grad = torch.zeros_like(emb)
            a.stream = L.stream_handle()
            a.xyzs, a.dirs, a.deltas = pre["xyzs"].data_ptr(), pre["dirs"].data_ptr(), pre["deltas"].data_ptr()
            a.rays = pre["rays"].data_ptr()
            a.counter = pre["counter"].data_ptr() if SKIP_PADDING_ROWS else None
            a.target = target.contiguous().view(-1, 3).data_ptr()
            a.loss = None if loss_out is None else loss_out.data_ptr()
            # (bit 1: the sharded tail with an owner range set -- this rank's slice of the table stays as record lists)
            a.flags = (3 if defer_dp else 1) if raw else 0
            # the next batch's march: kernels on the side stream, into the buffer set the current batch is NOT using
            nxt = key = None
            a.next_rays_o = None
            if nxt_ok:
                st_bufs = ctx["stages"][ctx["flip"]]
                if any(pre[k] is st_bufs[k] for k in ("xyzs", "rays")):
                    st_bufs = ctx["stages"][ctx["flip"] ^ 1]
                    which = ctx["flip"] ^ 1
                else:
                    which = ctx["flip"]
                nxt = _Stage(st_bufs)
                nxt["counter"] = nxt_counter
                nxt["slot"] = nxt_slot
                nxt["_which"] = which
                a.side_stream = side_stream.cuda_stream
                a.next_rays_o, a.next_rays_d = no.data_ptr(), nd.data_ptr()
                a.dt_gamma = float(dt_gamma)
                a.next_N, a.next_M, a.max_steps = Nn, Mn, int(max_steps)
                a.perturb = 1 if perturb else 0
                # (bit 4: this march stays on the side stream -- the cold window reads its count back behind it there)
                a.march_flags = occupied_box_flag(model) | 3 | 8 | (16 if cold_next else 0)
                for name in ("nears", "fars", "xyzs", "dirs", "deltas", "rays", "counter"):
                    setattr(a, "next_" + name, nxt[name].data_ptr())
                key = (no.data_ptr(), nd.data_ptr(), Nn, bool(perturb), float(dt_gamma), int(max_steps))
            a.table_grad = emb.grad.data_ptr()
            if raw:
                a.n_small = 0
                ctx["plan"] = None                  # (a later optimizer-carrying step rebuilds its arrays)
            else:
                plan = ctx.get("plan")
                if plan is None or ctx.get("plan_opt") is not opt:
                    # the optimizer's host arrays (pointers, sizes, learning rates, step counts of the MLP weights): built once
                    plan = ctx["plan"] = opt.grid_table_plan(emb, list(weights), list(grads))
                    ctx["plan_opt"] = opt
                    small = plan.arrays
                    st = plan.state
                    a.table_m, a.table_v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    a.n_small = small[0]
                    for name, arr in zip(("small_p", "small_g", "small_m", "small_v", "small_n", "small_lr", "small_step"),
                                         small[1:]):
                        setattr(a, name, _ct.cast(arr, _vp) if arr is not None else None)
            return ctx, a, nxt, key, M

        ctx, a, nxt, key, M = prepare(pre)
        if "cold_check" in pre:
            # everything above was host work that did not need the count; only now is it waited for
            fixed = _check_cold_stage(model, pre, rays_o, rays_d, perturb, dt_gamma, max_steps)
            if fixed is not pre:
                pre = fixed
                ctx, a, nxt, key, M = prepare(pre)
        emb, weights, grads = ctx["emb"], ctx["weights"], ctx["grads"]
        which = 0
        if nxt is not None:
            which = nxt.pop("_which")
            ctx["flip"] = which ^ 1              # (the set the next call looks at first is the one this stage does not use)
        if not raw:
            a.lr, a.beta1, a.beta2, a.eps, a.table_step = ctx["plan"]()
        cold_host = None
        if nxt is not None and cold_next and MIRROR_COUNT:
            # the cold window's next march writes its count straight into pinned host memory (enerf_march_mirror_count)
            hosts = ctx.get("cold_hosts")
            if hosts is None:
                hosts = ctx["cold_hosts"] = [torch.empty(2, dtype=torch.int32, pin_memory=True) for _ in range(2)]
            cold_host = hosts[which]
            cold_host.fill_(-1)               # (what _check_cold_stage watches for: both words are >= 0 once the count has landed)
            L.check(L.lib().enerf_march_mirror_count(cold_host.data_ptr()), "march_mirror_count")
        carried_before = L.lib().enerf_debug_carry_count(-2) if nxt is not None else 0
        L.check(L.lib().enerf_train_step_mse(_ct.byref(a)), "train_step_mse")
        # (the next batch's march may have ridden in this call's own launches, on this stream: no event to wait for then)
        march_carried = nxt is not None and L.lib().enerf_debug_carry_count(-2) != carried_before
        # (the launch counters bench.py reads: the library issued one grid_encode_forward / backward over M points)
        from .backends import _gridencoder as _gbk
        _gbk.STATS["fwd_points"] += M
        _gbk.STATS["budget_rows"] += M
        _gbk.STATS["fwd_calls"] += 1
        _gbk.STATS["bwd_points"] += M
        _gbk.STATS["bwd_calls"] += 1
        _gbk.LIFETIME["fwd_points"] += M
        _gbk.LIFETIME["fwd_calls"] += 1
        if nxt is not None:
            if cold_next:
                # the count of the march just queued travels to pinned memory behind it (one slot per stage set)
                hosts = ctx.get("cold_hosts")
                if hosts is None:
                    hosts = ctx["cold_hosts"] = [torch.empty(2, dtype=torch.int32, pin_memory=True) for _ in range(2)]
                host = hosts[which]
                with torch.cuda.stream(side_stream):
                    if cold_host is None:
                        host.fill_(-1)            # (what _check_cold_stage watches for: both words are >= 0 once the copy has landed)
                        host.copy_(nxt["counter"], non_blocking=True)
                    done = torch.cuda.Event()
                    done.record(side_stream)
                nxt["cold_check"] = (done, host, Mn)
                nxt["ready"] = done
            elif march_carried:
                nxt["ready"] = None                  # (marched on this very stream)
            else:
                nxt["ready"] = torch.cuda.Event()
                nxt["ready"].record(side_stream)
            stash[key] = nxt
            model._last_march_event = None if march_carried else (nxt["ready"], side_stream)
        if not raw:
            views = ctx.get("grad_views")
            if views is None:
                views = ctx["grad_views"] = [g.view_as(p) for p, g in zip(weights, grads)]
            for p, g in zip(weights, views):
                if p.grad is not g:
                    p.grad = g
    return (ctx["out_image"], ctx["dw"]) if raw else ctx["out_image"]


# ---------------------------------------------------------------------------------------------------------------------
# The event-only step (two renders, event loss, one optimizer pass) as ONE library call: csrc/train_step.hip
# enerf_train_step_events -- events.train_step_events_manual + FusedAdam.step_grid_table, call for call.
class _StepRender(_ct.Structure):         # enerf_step_render
    _fields_ = ([("N", _u32), ("M", _u32)]
                + [(n, _vp) for n in ("xyzs", "dirs", "deltas", "rays", "counter")]
                + [(n, _vp) for n in ("feats", "h32", "fb_s", "fb_c", "sigma", "rgb", "weights_sum", "image", "out_image",
                                      "g_image", "g_sigmas", "g_rgbs", "dx32", "dfeat")]
                + [("next_rays_o", _vp), ("next_rays_d", _vp), ("next_N", _u32), ("next_M", _u32)]
                + [(n, _vp) for n in ("next_nears", "next_fars", "next_xyzs", "next_dirs", "next_deltas", "next_rays",
                                      "next_counter")])


class _EventStepArgs(_ct.Structure):      # enerf_event_step_args
    _fields_ = ([("struct_bytes", _u32), ("mlp_precision", _ct.c_int), ("stream", _vp), ("side_stream", _vp),
                 ("r", _StepRender * 2), ("bg_color", _vp), ("pols", _vp), ("use_luma", _u32), ("linlog", _u32),
                 ("C_thres", _f32c), ("log_thres", _f32c), ("upstream", _f32c), ("delta", _vp), ("loss", _vp),
                 ("embeddings", _vp), ("offsets", _vp), ("level_scale_log2", _f32c), ("bound", _f32c),
                 ("inv_two_bound", _f32c), ("base_resolution", _u32), ("gridtype", _u32)]
                + [(n, _vp) for n in ("wseg_s", "wseg_c", "dwseg_s", "dwseg_c")]
                + [(n, _u32) for n in ("nh_s", "nh_c", "w0_cols_c", "out_c")]
                + [("aabb", _vp), ("bitfield", _vp), ("min_near", _f32c), ("dt_gamma", _f32c)]
                + [(n, _u32) for n in ("cascade", "grid_size", "max_steps", "perturb", "march_flags", "reserved0")]
                + [(n, _vp) for n in ("table", "table_grad", "table_m", "table_v")]
                + [(n, _f32c) for n in ("lr", "beta1", "beta2", "eps")]
                + [("table_step", _u32), ("n_small", _u32)]
                + [(n, _vp) for n in ("small_p", "small_g", "small_m", "small_v", "small_n", "small_lr", "small_step")]
                + [("flags", _u32), ("reserved", _u32)]
                + [(n, _vp) for n in ("m_feats", "m_h32", "m_fb_s", "m_fb_c", "m_sigma", "m_rgb", "m_g_sigmas", "m_g_rgbs",
                                      "m_dx32", "m_dfeat")])


MERGE_EVENT_RENDERS = True    # the one-call event step runs both renders' samples as one batch of 2 M rows (flags bit 1)


def native_events_supported(model, data, loss_opt, opt):
    """What enerf_train_step_events serves: the steady state (a sample budget exists) of the event-only step with the
    fused loss kernel (C_thres != -1, fp32, one background colour: a batch of one image), unit density scale, the fused
    optimizer, default march parameters."""
    ro, rd = data["rays_evs_o1"], data["rays_evs_d1"]
    return (NATIVE_STEP and _budget(model) > 0 and float(model.density_scale) == 1.0 and hasattr(opt, "grid_table_plan")
            and loss_opt.event_only and loss_opt.C_thres != -1 and not loss_opt.render_kwargs
            and int(getattr(loss_opt, "out_dim_color", 3)) == 3
            and data["images"].shape[0] == 1 and data["pols"].dtype == torch.float32 and data["pols"].is_cuda
            and data["rays_evs_o1"].shape == data["rays_evs_o2"].shape
            and data["pols"].numel() * 3 == data["rays_evs_o1"].numel()
            and supported(model, ro.contiguous().view(-1, 3), rd.contiguous().view(-1, 3), 1, 0)
            and supported(model, data["rays_evs_o2"].contiguous().view(-1, 3),
                          data["rays_evs_d2"].contiguous().view(-1, 3), 1, 0))


def _native_events_ctx(model, N, Ms, Nn, Mn, luma, dev):
    params = fnet.network_params(model)
    emb, weights = params[0], params[1:]
    kind = fnet.kind_of(model)
    arch = fnet._ARCH[kind]
    prec = model.__dict__.get("mlp_precision")
    prec = arch["prec"] if prec is None else int(prec)
    out_c = weights[-1].shape[0] if kind == "linear" else 3
    key = (N, Ms, Nn, Mn, kind, prec, out_c, luma, emb.data_ptr(), tuple(w.data_ptr() for w in weights), dev)
    ctx = model.__dict__.get("_native_events_ctx")
    if ctx is not None and ctx["key"] == key:
        return ctx
    import numpy as _np
    enc = model._modules["encoder"]
    bufs = model._buffers
    f32 = dict(dtype=torch.float32, device=dev)
    seg_s, seg_c = fnet._weight_segments(kind, weights)
    dw, (dseg_s, dseg_c) = fnet._grad_segments(kind, dev, out_c)
    grads = fnet.unpack_weight_grads(dw, out_c, kind)
    a = _EventStepArgs()
    a.struct_bytes = _ct.sizeof(_EventStepArgs)
    a.mlp_precision = -1 if prec is None else prec
    ts, stages, pair_bufs = [], [], []
    tm = None
    if MERGE_EVENT_RENDERS and Ms[0] == Ms[1]:
        M2 = 2 * Ms[0]                                  # (2 M is a multiple of 256: no extra padding of the level-major rows)
        tm = dict(m_feats=torch.empty(16, M2, 2, **f32), m_h32=torch.empty(M2, 32, **f32),
                  m_fb_s=torch.empty(arch["nh_s"], M2, 64, **f32), m_fb_c=torch.empty(arch["nh_c"], M2, 64, **f32),
                  m_sigma=torch.empty(M2, **f32), m_rgb=torch.empty(M2, out_c, **f32), m_g_sigmas=torch.empty(M2, **f32),
                  m_g_rgbs=torch.empty(M2, out_c, **f32), m_dx32=torch.empty(M2, 32, **f32),
                  m_dfeat=torch.empty(16, M2, 2, **f32))
        for name, buf in tm.items():
            setattr(a, name, buf.data_ptr())
    for q, M in enumerate(Ms):
        Mp = (M + 31) // 32 * 32
        if tm is not None:
            # the per-render scratch (steps whose two stages were not marched together: right after an update) is the two
            # halves of the merged allocations, each half in the per-render shape: no second set of buffers
            def half(name, *shape):
                flat = tm["m_" + name].view(-1)
                n = flat.numel() // 2
                return flat[q * n:(q + 1) * n].view(*shape)
            t = dict(feats=half("feats", 16, Mp, 2), h32=half("h32", M, 32), fb_s=half("fb_s", arch["nh_s"], Mp, 64),
                     fb_c=half("fb_c", arch["nh_c"], Mp, 64), sigma=half("sigma", M), rgb=half("rgb", M, out_c),
                     g_sigmas=half("g_sigmas", M), g_rgbs=half("g_rgbs", M, out_c), dx32=half("dx32", M, 32),
                     dfeat=half("dfeat", 16, Mp, 2))
        else:
            t = dict(feats=torch.empty(16, Mp, 2, **f32), h32=torch.empty(M, 32, **f32),
                     fb_s=torch.empty(arch["nh_s"], Mp, 64, **f32), fb_c=torch.empty(arch["nh_c"], Mp, 64, **f32),
                     sigma=torch.empty(M, **f32), rgb=torch.empty(M, out_c, **f32), g_sigmas=torch.empty(M, **f32),
                     g_rgbs=torch.empty(M, out_c, **f32), dx32=torch.empty(M, 32, **f32), dfeat=torch.empty(16, Mp, 2, **f32))
        t.update(weights_sum=torch.empty(N, **f32), image=torch.empty(N, 3, **f32), out_image=torch.empty(N, 3, **f32),
                 g_image=torch.empty(N, 3, **f32))
        ts.append(t)
        a.r[q].N, a.r[q].M = N, M
        for name, buf in t.items():
            setattr(a.r[q], name, buf.data_ptr())
        sets = []
        if Nn:
            for f in range(2):
                # the two renders' sample buffers of one stage generation are the halves of ONE allocation: a step that
                # consumes both finds the second render's rows right behind the first's M (merged layout, see below)
                if q == 0:
                    pair_bufs.append(dict(xyzs=torch.empty(2 * Mn, 3, **f32), dirs=torch.empty(2 * Mn, 3, **f32),
                                          deltas=torch.empty(2 * Mn, 2, **f32)))
                pb = pair_bufs[f]
                sets.append(dict(nears=torch.empty(Nn, **f32), fars=torch.empty(Nn, **f32),
                                 rays=torch.empty(Nn, 3, dtype=torch.int32, device=dev), xyzs=pb["xyzs"][q * Mn:(q + 1) * Mn],
                                 dirs=pb["dirs"][q * Mn:(q + 1) * Mn], deltas=pb["deltas"][q * Mn:(q + 1) * Mn], M=Mn))
        stages.append(sets)
    delta = torch.empty(1, N, 1 if luma else 3, **f32)
    a.delta = delta.data_ptr()
    a.embeddings, a.offsets = emb.data_ptr(), enc._buffers["offsets"].data_ptr()
    a.level_scale_log2 = float(_np.log2(enc.per_level_scale))
    a.bound, a.inv_two_bound = float(model.bound), float(_np.float32(1.0) / _np.float32(2 * model.bound))
    a.base_resolution, a.gridtype = int(enc.base_resolution), int(enc.gridtype_id)
    a.wseg_s, a.wseg_c = _ct.cast(seg_s, _vp), _ct.cast(seg_c, _vp)
    a.dwseg_s, a.dwseg_c = _ct.cast(dseg_s, _vp), _ct.cast(dseg_c, _vp)
    a.nh_s, a.nh_c, a.w0_cols_c, a.out_c = arch["nh_s"], arch["nh_c"], arch["w0c"], out_c
    a.aabb, a.bitfield = bufs["aabb_train"].data_ptr(), bufs["density_bitfield"].data_ptr()
    a.min_near = float(model.min_near)
    a.cascade, a.grid_size = int(model.cascade), int(model.grid_size)
    a.table = emb.data_ptr()
    ctx = dict(key=key, t=ts, tm=tm, seg=(seg_s, seg_c), dseg=(dseg_s, dseg_c), dw=dw, grads=grads, stages=stages,
               flip=[0, 0], a=a, emb=emb, weights=weights, delta=delta, kind=kind, out_c=out_c)
    model.__dict__["_native_events_ctx"] = ctx
    return ctx


def train_step_events_native(model, data, loss_opt, opt, next_data=None, side_stream=None, bg_color=None, perturb=True,
                             dt_gamma=0, max_steps=1024):
    """One event-only training step through enerf_train_step_events: both renders (marched ahead of time when the
    previous step asked for it), the event loss, both backwards, the optimizer, and the two marches of `next_data` on
    `side_stream`.  -> (loss, delta).  Gradients of the MLP weights are left in p.grad, the table's dense gradient buffer
    comes back clean, the optimizer's step counts advance.  Values: those of events.train_step_events_manual +
    FusedAdam.step_grid_table (the same library calls, in the same order)."""
    pairs = ((data["rays_evs_o1"], data["rays_evs_d1"]), (data["rays_evs_o2"], data["rays_evs_d2"]))
    dev = pairs[0][0].device
    with torch.no_grad():
        bg = torch.rand((1, 1, 3), device=dev) if bg_color is None else bg_color.detach().to(torch.float32).contiguous()
        pres, slots = [], []
        for ro, rd in pairs:
            ro, rd = ro.contiguous().view(-1, 3), rd.contiguous().view(-1, 3)
            pre = _take_premarched(model, ro, rd, perturb, dt_gamma, max_steps)
            if pre is not None:
                model.rendered_counter_slot = pre["slot"]
            else:
                counter = _next_counter(model)
                model.rendered_counter_slot = getattr(model, "last_counter_slot", None)
                pre = march_stage(model, ro, rd, counter, _budget(model), bool(perturb), False, float(dt_gamma),
                                  int(max_steps))
            pres.append(pre)
            slots.append(model.rendered_counter_slot)
        N = pairs[0][0].numel() // 3
        nxt_pairs = None
        Nn = Mn = 0
        stash = getattr(model, "_premarched", None)
        if not isinstance(stash, dict):
            stash = model._premarched = {}
        if next_data is not None and side_stream is not None and not any("pending" in p for p in stash.values()):
            nxt_pairs = [(next_data["rays_evs_o1"].contiguous().view(-1, 3), next_data["rays_evs_d1"].contiguous().view(-1, 3)),
                         (next_data["rays_evs_o2"].contiguous().view(-1, 3), next_data["rays_evs_d2"].contiguous().view(-1, 3))]
            Nn = nxt_pairs[0][0].shape[0]
            mc = _budget(model)
            Mn = mc + (128 - mc % 128)
        ctx = _native_events_ctx(model, N, (pres[0]["M"], pres[1]["M"]), Nn, Mn, bool(loss_opt.use_luma), dev)
        a, emb, weights, grads = ctx["a"], ctx["emb"], ctx["weights"], ctx["grads"]
        if emb.grad is None:
            emb.grad = torch.zeros_like(emb)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        a.stream = L.stream_handle()
        a.bg_color = bg.data_ptr()
        a.pols = data["pols"].contiguous().data_ptr()
        a.use_luma, a.linlog = int(bool(loss_opt.use_luma)), int(bool(loss_opt.linlog))
        a.C_thres, a.log_thres, a.upstream = float(loss_opt.C_thres), float(loss_opt.log_thres), 1.0
        a.loss = loss.data_ptr()
        for q, pre in enumerate(pres):
            r = a.r[q]
            r.xyzs, r.dirs, r.deltas = pre["xyzs"].data_ptr(), pre["dirs"].data_ptr(), pre["deltas"].data_ptr()
            r.rays = pre["rays"].data_ptr()
            r.counter = pre["counter"].data_ptr() if SKIP_PADDING_ROWS else None
            r.next_rays_o = None
        # merged layout: the second render's samples lie right behind the first's M rows (stages of one generation do)
        M0 = pres[0]["M"]
        merged = (ctx["tm"] is not None and pres[1]["M"] == M0
                  and pres[1]["xyzs"].data_ptr() == pres[0]["xyzs"].data_ptr() + 12 * M0
                  and pres[1]["dirs"].data_ptr() == pres[0]["dirs"].data_ptr() + 12 * M0
                  and pres[1]["deltas"].data_ptr() == pres[0]["deltas"].data_ptr() + 8 * M0)
        a.flags = 2 if merged else 0
        staged = []
        if nxt_pairs is not None:
            a.side_stream = side_stream.cuda_stream
            a.dt_gamma, a.max_steps, a.perturb = float(dt_gamma), int(max_steps), 1 if perturb else 0
            a.march_flags = occupied_box_flag(model) | 3 | 8
            for q, (no, nd) in enumerate(nxt_pairs):
                st_bufs = ctx["stages"][q][ctx["flip"][q]]
                if any(pres[q][k] is st_bufs[k] for k in ("xyzs", "rays")):
                    st_bufs = ctx["stages"][q][ctx["flip"][q] ^ 1]
                else:
                    ctx["flip"][q] ^= 1
                nxt = _Stage(st_bufs)
                nxt["counter"] = _next_counter(model)
                nxt["slot"] = getattr(model, "last_counter_slot", None)
                r = a.r[q]
                r.next_rays_o, r.next_rays_d, r.next_N, r.next_M = no.data_ptr(), nd.data_ptr(), Nn, Mn
                for name in ("nears", "fars", "xyzs", "dirs", "deltas", "rays", "counter"):
                    setattr(r, "next_" + name, nxt[name].data_ptr())
                staged.append(((no.data_ptr(), nd.data_ptr(), Nn, bool(perturb), float(dt_gamma), int(max_steps)), nxt))
        a.table_grad = emb.grad.data_ptr()
        plan = ctx.get("plan")
        if plan is None or ctx.get("plan_opt") is not opt:
            plan = ctx["plan"] = opt.grid_table_plan(emb, list(weights), list(grads))
            ctx["plan_opt"] = opt
            small, st = plan.arrays, plan.state
            a.table_m, a.table_v = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
            a.n_small = small[0]
            for name, arr in zip(("small_p", "small_g", "small_m", "small_v", "small_n", "small_lr", "small_step"), small[1:]):
                setattr(a, name, _ct.cast(arr, _vp) if arr is not None else None)
        a.lr, a.beta1, a.beta2, a.eps, a.table_step = plan()
        carried_before = L.lib().enerf_debug_carry_count(-2) if staged else 0
        L.check(L.lib().enerf_train_step_events(_ct.byref(a)), "train_step_events")
        # (the next step's marches may have ridden in this call's own launches, on this stream: no event to wait for then)
        march_carried = bool(staged) and L.lib().enerf_debug_carry_count(-2) != carried_before
        from .backends import _gridencoder as _gbk
        # (the launch counters bench.py reads: merged, the library issued ONE grid_encode_forward / backward over 2 M points)
        for points in ([pres[0]["M"] + pres[1]["M"]] if merged else [pre["M"] for pre in pres]):
            _gbk.STATS["fwd_points"] += points
            _gbk.STATS["budget_rows"] += points
            _gbk.STATS["fwd_calls"] += 1
            _gbk.STATS["bwd_points"] += points
            _gbk.STATS["bwd_calls"] += 1
            _gbk.LIFETIME["fwd_points"] += points
            _gbk.LIFETIME["fwd_calls"] += 1
        if staged:
            ready = None
            if not march_carried:
                ready = torch.cuda.Event()
                ready.record(side_stream)
            for key, nxt in staged:
                nxt["ready"] = ready
                stash[key] = nxt
            model._last_march_event = None if march_carried else (ready, side_stream)
        views = ctx.get("grad_views")
        if views is None:
            views = ctx["grad_views"] = [g.view_as(p) for p, g in zip(weights, grads)]
        for p, g in zip(weights, views):
            if p.grad is not g:
                p.grad = g
        model.rendered_counter_slot = slots[-1]
    return loss, ctx["delta"]
