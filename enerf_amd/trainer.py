"""Minimal train / render harness around the hot path (the bench's "step"), re-stating what main_nerf.py +
Trainer.train_one_epoch do per iteration in the reference (main_nerf.py:211-214, nerf/utils.py:943-997):

    every 16 steps: model.update_extra_state()          (density-grid EMA + packbits + mean_count)
    render -> loss -> backward -> [all-reduce grads] -> Adam(betas=(0.9, 0.99), eps=1e-15) step

`occupancy="synthetic"` keeps marching against the analytic scene's bitfield: update_extra_state() still runs (its cost
is part of the step) but the synthetic grid is restored afterwards, so sample counts stay reproducible with
random-init weights.

`use_graphs=True` (fused fp32 HIP path only): once the sample budget is known (after the first 16 steps) the
render -> loss -> backward part of an RGB step is captured into a HIP graph per (ray count, sample budget) and replayed;
update_extra_state, the gradient all-reduce and the optimizer stay outside.  The sample budget is rounded up to a
multiple of 8192 samples so that the 16-step windows share graphs (a larger budget never drops a ray the reference's
budget would keep).  A 4096-ray step is otherwise bounded by the host's launch rate, not by the kernels.
"""
import torch

from . import scene
from .optim import FusedAdam
from .parallel import GradAverager


_UNSET = object()


class TrainHarness:
    def __init__(self, model, lr=1e-2, occupancy="synthetic", world=1, update_interval=16, use_graphs=False,
                 optimizer=None, fp16=False, amp=None, prime_pool=None):
        self.model = model
        adam = optimizer or (FusedAdam if next(model.parameters()).is_cuda else torch.optim.Adam)
        self.opt = adam(model.get_params(lr), betas=(0.9, 0.99), eps=1e-15)
        self.occupancy = occupancy
        self.update_interval = update_interval
        self.global_step = 0
        self.avg = GradAverager(list(model.parameters())) if world > 1 else None
        self._syn = None
        if model.cuda_ray and occupancy == "synthetic":
            self._syn = scene.install_occupancy(model)
        self.early_budget = True      # update_extra_state: the sample budget is read before the update is queued (no stall)
        self.prefetch = True          # data parallel: march the next batch underneath the gradient all-reduce
        self.perturb = True           # training renders jitter their rays (nerf/utils.py:605 `perturb=True`); tests of
        #                               shard-vs-whole-batch equality switch it off: the jitter is seeded by the ray's
        #                               index in ITS batch (raymarching.cu:349-350)
        self.manual_mse = True        # RGB step: closed-form MSE gradient into the fused render node, no autograd engine
        self.fuse_table_adam = True   # one GPU: the table gradient's tile sums feed Adam straight from LDS
        self.overlap_update = True    # update steps: the render's count pass is queued before the update's read-back
        self.native_step = True       # steady-state RGB steps on one GPU as ONE library call (enerf_train_step_mse)
        self._native_route_sig, self._native_route_dp, self._pending_sig = None, False, None
        self._params = [p for g in self.opt.param_groups for p in g["params"]]
        enc = getattr(model, "encoder", None)
        table = getattr(enc, "embeddings", None)
        self._small_params = [p for p in self._params if p is not table]
        self._opt_step = getattr(self.opt, "step_now", self.opt.step)
        self._side = None             # HIP stream of the next batch's march (created on first use)
        self.comm_chunks = 4          # data parallel: pieces of the hash-table gradient all-reduce (0: one bucket + Adam)
        self.comm_mode = "allreduce"  # or "sharded": reduce-scatter -> Adam on this rank's slice -> all-gather
        self.comm_dtype = None        # torch.bfloat16: halve the table gradient's bytes on the wire (changes rounding)
        # sharded tail: this rank's slice of the table keeps its own record lists for the optimizer pass (the one-GPU
        # flush, csrc/gridencoder.hip OwnerRange) and only the rest of the gradient is made dense for the reduce-scatter
        self.fused_sharded = True
        # where the next batch's march is released on the side stream: once the "forward" is queued (runs beside the
        # MLP backward), once the MLP backward is ("mlp_backward": runs beside the hash table's backward and the
        # optimizer -- the MFMA kernels, one register-filling wavefront per SIMD, then have the CUs to themselves:
        # steady step 0.470 -> 0.434 ms on one MI355X), or, data parallel, once the "collectives" are
        self.prefetch_at = "mlp_backward"
        self._raw_grads = None
        self._cleared_grad = None     # the embeddings' gradient buffer as the last Adam pass left it (all zeros)
        dev0 = next(model.parameters()).device
        self._prime_pool(dev0, prime_pool)
        self._loss_ring = torch.zeros(64, device=dev0)
        self._loss_cursor = 0
        # Mixed precision, two separate switches:
        #   fp16=True: the shipped configs' `fp16 = True` (nerf/utils.py:350,964-975: autocast(float16) + GradScaler).  Two
        #     routes to the same regime, chosen here:
        #       * the closed-form step on fp16 operands (`amp_f16`; one GPU, a model the fused path serves, FusedAdam): the
        #         nn.Linear nets on v_mfma_f32_32x32x16_f16 with every layer's activations and activation gradients rounded
        #         to fp16 (enerf_mlp32_precision(3)), fp32 accumulation, fp32 marching / compositing / trunc_exp as under
        #         the reference's autocast, and the GradScaler's protocol on the device (csrc/optim.hip enerf_amp_*: the
        #         loss gradient is multiplied by the scale, a step whose weight gradients are not finite is not applied
        #         and halves the scale, 2000 clean steps double it) on the scaler's own tensors -- so its state_dict() is
        #         what a checkpoint stores and restores.  What is NOT reproduced: the half copy of the hash table
        #         (gridencoder/grid.py:38-39: the gather reads the fp32 table, features are fp32 until the first layer
        #         rounds them).
        #       * fp16="autocast" (and whatever the closed-form step does not serve): torch.autocast + GradScaler around
        #         the op-by-op route, half hash table + half table gradient (gridencoder/grid.py:38-39,72), half SH.
        #   amp="bf16": NOT the reference's regime, named apart for that reason -- the closed-form step with the networks
        #     on bf16 operands (mlp32 precision 2), fp32 hash table, fp32 master weights.  bf16 has fp32's exponent range,
        #     so there is no loss scaling: the GradScaler is kept disabled.  The model's own `mlp_precision` is only
        #     overridden for the duration of this harness's steps.
        if fp16 == "bf16":                      # (the spelling of earlier rounds)
            fp16, amp = False, "bf16"
        if amp not in (None, "bf16"):
            raise ValueError(f"amp={amp!r}: None or 'bf16'")
        self.fp16 = bool(fp16)                  # the autocast route (all of it, or the steps the closed form cannot take)
        self.amp_bf16 = self.amp_f16 = False
        cuda_fused = False
        if (fp16 or amp) and next(model.parameters()).is_cuda:
            from . import fused_network
            cuda_fused = fused_network.kind_of(model) is not None
        if fp16 is True and cuda_fused and world == 1 and hasattr(self.opt, "step_grid_table") \
                and getattr(model, "cuda_ray", False):
            self.fp16, self.amp_f16 = False, True
        if amp == "bf16" and not fp16:
            if cuda_fused:
                self.amp_bf16 = True            # (scoped to this harness's own steps: _amp_scope)
            else:
                raise ValueError("amp='bf16' needs a CUDA model the fused path serves (network.py / network_ff.py nets)")
        self.scaler = (torch.amp.GradScaler("cuda", enabled=self.fp16 or self.amp_f16)
                       if (self.fp16 or self.amp_bf16 or self.amp_f16) else None)
        self._amp_words = None
        if self.amp_f16:
            self.scaler._lazy_init_scale_growth_tracker(dev0)
            self._amp_words = torch.zeros(2, dtype=torch.int32, device=dev0)       # [found_inf, skipped steps]
        # what Trainer keeps beside the model and lands in its checkpoints (nerf/utils.py:381-389,1300-1304)
        self.epoch = 1
        self.stats = {"loss": [], "valid_loss": [], "results": [], "checkpoints": [], "best_result": None}
        self.lr_scheduler = None      # set_lr_scheduler(): stepped after every optimizer step (main_nerf.py:212,214)
        self.use_graphs = bool(use_graphs)
        self._graphs = {}
        self._graph_generation = 0
        if self.use_graphs:
            model.sample_budget_quantum = 8192

    @staticmethod
    def _prime_pool(dev, want):
        """Once per process: leave one large free block in torch's caching pool.  The first 16 steps size their sample
        buffers from each render's own count (a different size every step), and every size the pool cannot serve is a
        hipMalloc in the middle of a step -- milliseconds on a fresh process, in a loop whose step is 0.4 ms.
        `want`: None = on unless ENERF_PRIME_POOL=0, False = off, True = on, an int = that many bytes.  Sized from what
        is actually free (at most a quarter of it, 2 GiB at most, nothing under 1 GiB free): several ranks on one device
        or a nearly full GPU must not lose a harness to it, and a failed priming is a no-op."""
        import os
        if dev.type != "cuda" or getattr(TrainHarness, "_pool_primed", False) or want is False:
            return
        if want is None and os.environ.get("ENERF_PRIME_POOL", "1") == "0":
            return
        try:
            free, _total = torch.cuda.mem_get_info(dev)
            size = int(want) if (want is not True and want is not None) else min(2 << 30, free // 4)
            if free < (1 << 30) or size <= 0 or size > free // 2:
                return
            torch.empty(size, dtype=torch.uint8, device=dev)
            TrainHarness._pool_primed = True
        except Exception:                       # noqa: BLE001 -- out of memory or no device API: train without it
            pass

    def _amp_scope(self):
        """amp='bf16' / the fp16 closed form: the fused kernels read `model.mlp_precision` at launch -- set for this
        harness's step, restored after it (the model keeps its own arithmetic for inference and for other harnesses)."""
        m = self.model
        prev = m.__dict__.get("mlp_precision", _UNSET)
        m.mlp_precision = 3 if self.amp_f16 else 2
        return prev

    def amp_skipped_steps(self):
        """fp16 closed form: optimizer steps the loss scaling has skipped so far (their gradients were not finite) -- the
        optimizer's own `step` counts them, its bias corrections do not.  One 4-byte read-back."""
        return int(self._amp_words[1]) if self._amp_words is not None else 0

    def _amp_closed_form_ok(self):
        """Decided per step (graph replay, an attached averager and `fuse_table_adam` can all change after __init__): what
        the device-side GradScaler protocol cannot serve takes the autocast route with the same scaler instead of raising."""
        return not self.use_graphs and self.avg is None and bool(self.fuse_table_adam)

    def _amp_step(self, fn, *args, **kw):
        """One step under the device-side GradScaler protocol (include/enerf_hip.h: enerf_amp_begin / enerf_amp_end)."""
        from . import _lib as L
        lib = L.lib()
        sc = self.scaler
        if self.use_graphs or self.avg is not None or not self.fuse_table_adam:
            raise RuntimeError("the fp16 closed-form step needs one GPU, the fused table optimizer and no graph replay "
                               "(use fp16='autocast')")
        L.check(lib.enerf_amp_begin(sc._scale.data_ptr(), sc._growth_tracker.data_ptr(), self._amp_words.data_ptr(),
                                    self._amp_words.data_ptr() + 4), "amp_begin")
        try:
            out = fn(*args, **kw)
        except BaseException:
            lib.enerf_amp_cancel()
            raise
        L.check(lib.enerf_amp_end(float(sc.get_growth_factor()), float(sc.get_backoff_factor()),
                                  int(sc.get_growth_interval()), L.stream_handle()), "amp_end")
        return out

    def _amp_restore(self, prev):
        if prev is _UNSET:
            self.model.__dict__.pop("mlp_precision", None)
        else:
            self.model.mlp_precision = prev

    def set_lr_scheduler(self, factory):
        """`factory(optimizer) -> scheduler`, as the reference's Trainer takes it (main_nerf.py:212: LambdaLR with
        0.1 ** min(iter / iters, 1)); stepped once after every optimizer step (`scheduler_update_every_step=True`).
        Every optimizer route of the harness reads `param_groups[...]['lr']` at launch time, the fused table pass
        (FusedAdam.step_grid_table) included, so the schedule reaches all of them."""
        self.lr_scheduler = factory(self.opt)
        self.opt._opt_called = True       # the harness calls step_now / step_grid_table, not the wrapped step()
        return self.lr_scheduler

    def save_checkpoint(self, path, full=False):
        """The reference's checkpoint dict (nerf/utils.py:1295-1351) -> `path`."""
        from .checkpoint import save_checkpoint
        return save_checkpoint(self, path, full=full)

    def load_checkpoint(self, checkpoint, model_only=False):
        """Resume from a checkpoint in the reference's format (nerf/utils.py:1353-1415), whoever wrote it."""
        from .checkpoint import load_checkpoint
        self._graphs.clear()
        return load_checkpoint(self, checkpoint, model_only=model_only)

    def maybe_update_extra_state(self, coming_render=None):
        """`coming_render` = (rays_o, rays_d) of the render this step starts with, when it will take the fused path with
        default march settings: its near/far + count pass are queued before the update's read-back is waited for
        (fused_render.premarch_count), so the device works while the host waits."""
        m = self.model
        if m.cuda_ray and self.global_step % self.update_interval == 0:
            begin = getattr(m, "update_extra_state_begin", None)
            early = None
            if begin is None or coming_render is None or not self.overlap_update or self.use_graphs:
                m.update_extra_state()
                handle = None
            else:
                if self.early_budget:
                    # the window's step counters, read behind the last march on the side stream: the budget is known
                    # before the update is even queued, and the host never waits behind the sweep
                    from . import fused_render as _fr
                    early = _fr.early_mean_count(m)
                handle = begin()
            if self._syn is not None:
                m.density_grid.copy_(self._syn[0])
                m.density_bitfield.copy_(self._syn[1])
            if handle is not None:
                from . import fused_render
                m._premarched = None                    # (anything marched against the old bitfield is void)
                fused_render.premarch_count(m, *coming_render, perturb=self.perturb)
                if early is not None:
                    m.update_extra_state_end(handle, early)
                else:
                    m.update_extra_state_end(handle)
            self._agree_on_budget()

    def _agree_on_budget(self):
        """Data parallel: every rank must take the same route through the step (the routes differ in the collectives
        they issue), and the routes are chosen from static properties of the model plus -- for graph replay -- the
        sample budget.  The budget comes from each rank's own step counters ("all-reduce-MAX it every 16
        steps so M is uniform"): one 8-byte MAX all-reduce per update_extra_state window makes it the same everywhere
        (a larger budget never drops a ray the local budget would keep)."""
        if self.avg is None:
            return
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        m = self.model
        dev = next(m.parameters()).device
        t = torch.tensor([int(m.mean_count)], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        m.mean_count = int(t.item())

    # ------------------------------------------------------------------ HIP-graph replay of render + loss + backward
    def _graph_key(self, tag, rays_o):
        m = self.model
        q = m.sample_budget_quantum
        return (tag, tuple(rays_o.shape), (int(m.mean_count) + q - 1) // q * q)

    def _capture(self, inputs, loss_fn, fwd_bwd_fn=None):
        """Capture loss_fn(static inputs) + backward -- or fwd_bwd_fn(static inputs), which leaves the gradients in
        p.grad itself and returns the loss; returns the replay state."""
        from . import _lib
        m = self.model
        _lib.prof.enable(False)                       # hipEvent timing hooks cannot live inside a captured graph
        if getattr(m, "graph_counter", None) is None:
            m.graph_counter = torch.zeros(2, dtype=torch.int32, device=inputs[0].device)
        st = {"in": [t.clone() for t in inputs]}
        params = [p for p in m.parameters() if p.requires_grad]

        def fwd_bwd():
            if fwd_bwd_fn is not None:
                return fwd_bwd_fn(*st["in"])
            loss = loss_fn(*st["in"])
            loss.backward()
            return loss

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                        # warm-up on the capture stream: workspaces reach their size
                self.opt.zero_grad(set_to_none=True)
                fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        gen = _lib.lib().enerf_workspace_generation()
        if gen != self._graph_generation:             # the warm-up grew a scratch buffer: older captures point at
            self._graphs.clear()                      # freed memory (see _graph_for)
            self._graph_generation = gen
        self.opt.zero_grad(set_to_none=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st["loss"] = fwd_bwd().detach()
        st["graph"] = g
        st["grads"] = [(p, p.grad) for p in params]
        assert _lib.lib().enerf_workspace_generation() == gen, "a scratch buffer grew inside a graph capture"
        return st

    def _graph_for(self, key):
        """The captured state for `key`, or None.  The library's scratch buffers are grow-only: a launch that needed
        more (a larger budget or ray count) re-allocated one since the captures were made, and every captured graph
        holds the old pointers -- all of them are dropped and re-captured on demand."""
        if self._graphs:
            from . import _lib
            if _lib.lib().enerf_workspace_generation() != self._graph_generation:
                self._graphs.clear()
        return self._graphs.get(key)

    def _replay(self, st, inputs, renders):
        m = self.model
        for dst, src in zip(st["in"], inputs):
            dst.copy_(src)
        # the captured count pass clips rays against the library's occupied-cell box, which only host code refreshes
        # (fused_render.occupied_box_flag): update_extra_state may have rewritten the bitfield since the capture
        from . import fused_render
        fused_render.occupied_box_flag(m)
        st["graph"].replay()
        for p, g in st["grads"]:
            p.grad = g
        # every render of the step used the same fixed-address counter; the ring gets the last render's counts
        for _ in range(renders):
            m.step_counter[m.local_step % 16].copy_(m.graph_counter)
            m.local_step += 1
        if self.avg is not None:
            self.avg()
        self._opt_step()
        return st["loss"].detach()

    def _graphable(self, rays_o, rays_d):
        from . import fused_render
        return (self.use_graphs and not self.fp16 and self.model.mean_count > 0
                and fused_render.supported(self.model, rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3),
                                           1, 0))

    def _reduce_grads(self, next_rays=None):
        """Average gradients across ranks.  With `next_rays` = (rays_o, rays_d) of the following step, the part of that
        step's render that does not read the parameters (near_far + march_rays_train, ~20 % of a step) is issued right
        after the collectives and runs underneath them; skipped whenever the next step starts with
        update_extra_state (new bitfield / sample budget) or the sample budget is not known yet."""
        if self.avg is None:
            return
        self.avg.start()
        m = self.model
        if (next_rays is not None and self.prefetch and m.cuda_ray
                and self.global_step % self.update_interval != 0):
            from . import fused_render
            ro, rd = next_rays
            if fused_render.supported(m, ro.contiguous().view(-1, 3), rd.contiguous().view(-1, 3), 1, 0):
                fused_render.prefetch_march(m, ro, rd, perturb=self.perturb)
        self.avg.finish()

    def _manual_ok(self, rays_o, rays_d, target, render_kw):
        from . import fused_render
        m = self.model
        return (self.manual_mse and m.cuda_ray and target.dtype == torch.float32
                and not set(render_kw) - {"dt_gamma", "max_steps"}
                and fused_render.supported(m, rays_o.contiguous().view(-1, 3), rays_d.contiguous().view(-1, 3), 1,
                                           render_kw.get("dt_gamma", 0)))

    def _side_prefetch(self, *next_rays):
        """-> a callable that marches the next step's ray sets (one (rays_o, rays_d) pair per render) on the side
        stream (see fused_render.prefetch_march), or None when the next step starts with update_extra_state (new
        bitfield / budget) or nothing is known about it."""
        m = self.model
        if (not next_rays or any(r is None for r in next_rays) or not self.prefetch
                or getattr(m, "graph_counter", None) is not None or self.global_step % self.update_interval == 0):
            return None
        from . import fused_render
        for ro, rd in next_rays:
            if not fused_render.supported(m, ro.contiguous().view(-1, 3), rd.contiguous().view(-1, 3), 1, 0):
                return None
        if self._side is None:
            self._side = torch.cuda.Stream()

        def issue(background=True, signalled=False):
            # signalled: called right after the MLP backward's reduce launch, which carries a completion signal
            # (fused_network.nerf_backward) -- the side stream waits for that instead of an event record here.
            # (the second march of an event step is ordered behind the first on the side stream: no wait of its own)
            for k, (ro, rd) in enumerate(next_rays):
                fused_render.prefetch_march(m, ro, rd, perturb=self.perturb, stream=self._side, background=background,
                                            after_signal=(True if k == 0 else "ordered") if signalled else False)
        return issue

    def _manual_fwd_bwd(self, rays_o, rays_d, target, dt_gamma=0, max_steps=1024, after_forward=None, raw=False,
                        defer_table=False, after_mlp_backward=None):
        """Render + MSE + backward with the loss gradient in closed form (fused_render.train_step_mse): same kernels
        for the render and its backward, no autograd graph, no loss-backward / blend / depth / fill launches.
        Leaves the gradients in p.grad, returns the loss."""
        from . import fused_network, fused_render
        m = self.model
        # nothing accumulates across steps (zero_grad(set_to_none=True)) -- except that the hash table's gradient
        # buffer is kept when the previous step's Adam pass cleared it (step_now(zero_grads=True)): the grid backward
        # then adds straight into it, with no 52 MB allocation and fill
        emb = m._modules["encoder"]._parameters["embeddings"]
        keep = emb.grad if (not self.use_graphs and self._cleared_grad is not None
                            and emb.grad is self._cleared_grad) else None
        self._cleared_grad = None
        for p in self._params:
            p.grad = None
        if keep is not None:
            emb.grad = keep
        loss = None
        if not self.use_graphs:                     # (a captured graph would always accumulate into the same slot)
            # loss values land in a ring of device scalars, cleared once per lap: no loss kernels, no per-step fill
            # (laps are counted on the ring's own cursor: steps of other kinds in between do not use slots)
            loss = self._loss_slot()
        if defer_table and emb.grad is None:        # the dense part of the gradient (levels too small to bin) needs a home
            emb.grad = torch.zeros_like(emb)
        image, grads = fused_render.train_step_mse(m, rays_o, rays_d, target, 1, self.perturb, dt_gamma, max_steps,
                                                   after_forward=after_forward, loss_out=loss, raw=raw,
                                                   defer_table=defer_table, after_mlp_backward=after_mlp_backward)
        if raw:
            self._raw_grads = grads                 # (embedding gradient, flat dW): _finish_distributed takes over
        else:
            for p, g in zip(fused_network.network_params(m), grads):
                if g is not None:
                    p.grad = g.view_as(p)
        if loss is not None:
            return loss             # a view into the ring: overwritten one lap (64 steps) later -- clone to keep it
        with torch.no_grad():
            return torch.nn.functional.mse_loss(image, target.view(-1, 3))

    def _finish_distributed(self, issue_prefetch=None):
        """Data-parallel tail of the closed-form step: the hash-table gradient is all-reduced in `comm_chunks` pieces
        and Adam runs on each piece as it lands (the optimizer pass over the table hides under the remaining
        collectives); the MLP gradients travel as the backward's one flat dW buffer."""
        import torch.distributed as dist
        from . import fused_network
        m = self.model
        g_emb, dw = self._raw_grads
        self._raw_grads = None
        emb = m.encoder.embeddings
        if g_emb is None:
            g_emb = emb.grad                                  # the kept buffer: the grid backward added into it
        else:
            emb.grad = g_emb
        nccl = dist.get_backend() == "nccl"                 # RCCL averages in the collective; gloo only sums
        op = dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM
        inv = 1.0 / dist.get_world_size()
        flat = g_emb.view(-1)
        n = flat.numel()
        step = -(-n // self.comm_chunks)
        step += (-step) % 4                                  # FusedAdam ranges start on multiples of 4 elements
        bounds = [(lo, min(lo + step, n)) for lo in range(0, n, step)]
        if self.comm_dtype is None:
            wire = [flat[lo:hi] for lo, hi in bounds]
        else:                                                # opt-in: the table gradient crosses xGMI in 16 bits
            wire = [flat[lo:hi].to(self.comm_dtype) for lo, hi in bounds]
        works = [dist.all_reduce(t, op=op, async_op=True) for t in wire]
        w_dw = dist.all_reduce(dw, op=op, async_op=True)
        if issue_prefetch is not None:
            issue_prefetch(background=False)                  # marches while the gradients are on the wire
        for (lo, hi), t, w in zip(bounds, wire, works):
            w.wait()
            if self.comm_dtype is not None:
                flat[lo:hi].copy_(t)
            if not nccl:
                flat[lo:hi].mul_(inv)
            self.opt.step_now(only=[emb], ranges={emb: (lo, hi)}, zero_grads=True)
        self._cleared_grad = emb.grad                         # every piece cleared by its Adam pass: kept for the next step
        w_dw.wait()
        if not nccl:
            dw.mul_(inv)
        small = fused_network.network_params(m)[1:]
        for p, g in zip(small, fused_network.unpack_weight_grads(dw, getattr(m, "out_dim_color", 3),
                                                                 fused_network.kind_of(m))):
            p.grad = g.view_as(p)
        self.opt.step_now(only=small)

    def _owner_range(self):
        """(lo, hi, world) of this rank's slice of the flat table for the fused sharded tail, or None when that tail does
        not apply (all-reduce tail, ragged shards, 16-bit wire format, an optimizer without the record-list pass)."""
        import torch.distributed as dist
        if not (self.fused_sharded and self.comm_mode == "sharded" and self.comm_dtype is None and self.avg is not None
                and hasattr(self.opt, "step_grid_table") and not self.use_graphs
                and dist.is_available() and dist.is_initialized()):
            return None
        emb = getattr(getattr(self.model, "encoder", None), "embeddings", None)
        if emb is None or getattr(self.model.encoder, "level_dim", 0) != 2:
            return None
        world, rank = dist.get_world_size(), dist.get_rank()
        # measurement aid (tools/dp_tail_overhead.py): a one-rank world that OWNS only 1 / N of the table, i.e. pays an
        # N-rank world's dense route for the other (N - 1) / N (the collectives degenerate; nothing is averaged)
        pretend = int(getattr(self, "pretend_world", 0) or 0)
        if pretend > 1 and world == 1:
            world = pretend
        n = emb.numel()
        if n % world or (n // world) % 4:
            return None
        shard = n // world
        return rank * shard, (rank + 1) * shard, world

    def _finish_sharded_fused(self, own, issue_prefetch=None):
        """The sharded tail with the one-GPU flush kept for this rank's own slice: the backward left the slice's tiles as
        record lists and made only the rest of the gradient dense (enerf_grid_owner_range); the dense buffer is
        reduce-scattered in place (SUM: this rank's slice receives the OTHER ranks' share), the optimizer pass sums its own
        lists in LDS on top of it, divides by the number of ranks, updates the slice and clears the buffer, and the slices
        are all-gathered in place.  Same update as _finish_sharded up to the order of the fp32 sums."""
        import torch.distributed as dist
        from . import _lib as L
        from . import fused_network
        m = self.model
        lo, hi, world = own
        g_emb, dw = self._raw_grads
        self._raw_grads = None
        emb = m.encoder.embeddings
        if g_emb is None:
            g_emb = emb.grad
        else:
            emb.grad = g_emb
        nccl = dist.get_backend() == "nccl"
        flat = g_emb.view(-1)
        try:
            w_dw = dist.all_reduce(dw, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, async_op=True)
            # SUM, not AVG: the optimizer pass applies 1 / ranks to dense share + own lists together (and a one-rank world's
            # in-place SUM is free where RCCL's AVG runs a scaling kernel over the 52 MB)
            real = dist.get_world_size() == world             # (False: tools/dp_tail_overhead.py's pretend world)
            if nccl:                                          # in place: slice r of the buffer <- sum of everybody's
                work = dist.reduce_scatter_tensor(flat[lo:hi], flat if real else flat[lo:hi], op=dist.ReduceOp.SUM,
                                                  async_op=True)
            else:                                             # gloo has no reduce-scatter
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
            if issue_prefetch is not None:
                issue_prefetch(background=False)
            work.wait()
            w_dw.wait()
            if not nccl:
                dw.mul_(1.0 / world)
            small = fused_network.network_params(m)[1:]
            for q, g in zip(small, fused_network.unpack_weight_grads(dw, getattr(m, "out_dim_color", 3),
                                                                     fused_network.kind_of(m))):
                q.grad = g.view_as(q)
            enc = m.encoder
            self.opt.step_grid_table(emb, enc.offsets, enc.level_dim, extra=small)      # (the owner range is still set)
        finally:
            L.lib().enerf_grid_owner_range(0, 0, 1.0)
        self._cleared_grad = emb.grad                         # cleared everywhere by the optimizer pass
        p = emb.data.view(-1)
        if nccl:
            dist.all_gather_into_tensor(p if real else p[lo:hi], p[lo:hi])          # in place: slice r of p <- rank r
        else:
            pieces = [torch.empty(hi - lo, dtype=p.dtype, device=p.device) for _ in range(world)]
            dist.all_gather(pieces, p[lo:hi].contiguous())
            for r, piece in enumerate(pieces):
                if r * (hi - lo) != lo:
                    p[r * (hi - lo):(r + 1) * (hi - lo)].copy_(piece)

    @staticmethod
    def _discard_pending_records():
        """A step that left the table gradient as record lists died before its optimizer pass: empty the lists so that
        the next backward is not refused."""
        from . import _lib
        _lib.lib().enerf_grid_records_discard(_lib.stream_handle())

    def _finish_sharded(self, issue_prefetch=None):
        """The other data-parallel tail (SURVEY.md 8e "scaling risk (b)"): the table gradient is reduce-scattered, every
        rank runs Adam on its own 1/N of the table only (the 28 B/element optimizer pass shrinks N-fold) and the updated
        slices are all-gathered into every replica's table.  Same bytes on the wire as the ring all-reduce
        (2 (N-1)/N x 52 MB), but the gather half moves parameters, which the next step needs only at its first grid
        encode.  Replicas stay bit-identical: every element is updated by exactly one rank and copied to the others.
        The MLP gradients (37 KB) keep their all-reduce; their Adam runs everywhere."""
        import torch.distributed as dist
        from . import fused_network
        m = self.model
        g_emb, dw = self._raw_grads
        self._raw_grads = None
        emb = m.encoder.embeddings
        if g_emb is None:
            g_emb = emb.grad
        else:
            emb.grad = g_emb
        world, rank = dist.get_world_size(), dist.get_rank()
        nccl = dist.get_backend() == "nccl"
        flat = g_emb.view(-1)
        n = flat.numel()
        shard = -(-n // world)
        shard += (-shard) % 4                                 # FusedAdam ranges start on multiples of 4 elements
        lo, hi = min(rank * shard, n), min((rank + 1) * shard, n)
        even = shard * world == n
        w_dw = dist.all_reduce(dw, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, async_op=True)
        if nccl and even:
            mine = torch.empty(shard, dtype=flat.dtype, device=flat.device)
            work = dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.AVG, async_op=True)
        else:                                                 # gloo has no reduce-scatter; ragged tables: all-reduce
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG if nccl else dist.ReduceOp.SUM, async_op=True)
            mine = None
        if issue_prefetch is not None:
            issue_prefetch(background=False)
        work.wait()
        if mine is not None:
            flat[lo:hi].copy_(mine)
        elif not nccl:
            flat[lo:hi].mul_(1.0 / world)
        if hi > lo:
            self.opt.step_now(only=[emb], ranges={emb: (lo, hi)}, zero_grads=True, advance=True)
        # this rank's local contributions to the other slices are spent: clear them for the next step
        flat[:lo].zero_()
        flat[hi:].zero_()
        self._cleared_grad = emb.grad
        p = emb.data.view(-1)
        if even and nccl:
            gather = dist.all_gather_into_tensor(p, p[lo:hi], async_op=True)       # in place: slice r of p <- rank r
        else:                                                 # equal-sized (padded) pieces for any backend / ragged table
            send = torch.zeros(shard, dtype=p.dtype, device=p.device)
            send[:hi - lo] = p[lo:hi]
            pieces = [torch.empty(shard, dtype=p.dtype, device=p.device) for _ in range(world)]
            gather = dist.all_gather(pieces, send, async_op=True)
        w_dw.wait()
        if not nccl:
            dw.mul_(1.0 / world)
        small = fused_network.network_params(m)[1:]
        for q, g in zip(small, fused_network.unpack_weight_grads(dw, getattr(m, "out_dim_color", 3),
                                                                 fused_network.kind_of(m))):
            q.grad = g.view_as(q)
        self.opt.step_now(only=small)
        gather.wait()
        if not (even and nccl):
            for r, piece in enumerate(pieces):
                a, b = min(r * shard, n), min((r + 1) * shard, n)
                if r != rank and b > a:
                    p[a:b].copy_(piece[:b - a])

    def gather_sharded_optimizer_state(self):
        """After steps taken with the sharded tail every rank holds current Adam moments only for its own slice of the
        table.  Before anything that needs them whole -- switching back to the all-reduce tail, saving a checkpoint --
        the slices are all-gathered (2 x 52 MB, once)."""
        import torch.distributed as dist
        emb = getattr(getattr(self.model, "encoder", None), "embeddings", None)
        opt = getattr(self, "opt", None)
        st = opt.state.get(emb) if emb is not None and opt is not None else None
        if not st or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        world, rank = dist.get_world_size(), dist.get_rank()
        n = emb.numel()
        shard = -(-n // world)
        shard += (-shard) % 4
        lo, hi = min(rank * shard, n), min((rank + 1) * shard, n)
        for key in ("exp_avg", "exp_avg_sq"):
            flat = st[key].view(-1)
            send = torch.zeros(shard, dtype=flat.dtype, device=flat.device)
            send[:hi - lo] = flat[lo:hi]
            pieces = [torch.empty(shard, dtype=flat.dtype, device=flat.device) for _ in range(world)]
            dist.all_gather(pieces, send)
            for r, piece in enumerate(pieces):
                a, b = min(r * shard, n), min((r + 1) * shard, n)
                if r != rank and b > a:
                    flat[a:b].copy_(piece[:b - a])

    def tune_comm(self, step_fn, candidates=(1, 2, 4, 8), window=None):
        """Data parallel: pick `comm_chunks` by measurement.  How the table-gradient all-reduce is best cut depends on
        the link topology and the number of ranks (per-collective latency against Adam / collective overlap), so each
        candidate runs `window` steps of `step_fn(i)` -- one update_extra_state period, so every window holds the same
        work -- timed between device synchronisations; the slowest rank's time decides (MAX all-reduce, hence the same
        choice on every rank).  -> {chunks: ms_per_step}, {} when there is nothing to tune."""
        import time
        import torch.distributed as dist
        if self.avg is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return {}
        window = int(window or self.update_interval)
        dev = next(self.model.parameters()).device
        sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
        i = 0
        timings = {}
        self.comm_mode = "allreduce"
        for n, c in enumerate((candidates[0],) + tuple(candidates)):       # the first window only warms up
            self.comm_chunks = int(c)
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(window):
                step_fn(i)
                i += 1
            sync()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            if n:
                timings[int(c)] = float(dt.item()) / window * 1e3
        self.comm_chunks = min(timings, key=timings.get)
        # the other tail: reduce-scatter -> Adam on this rank's slice -> all-gather (two windows: the first warms up)
        self.comm_mode = "sharded"
        sharded_ms = None
        for n in range(2):
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(window):
                step_fn(i)
                i += 1
            sync()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            sharded_ms = float(dt.item()) / window * 1e3
        self.gather_sharded_optimizer_state()                 # whichever tail runs next starts from whole moments
        self.comm_mode = "sharded" if sharded_ms < timings[self.comm_chunks] else "allreduce"
        # with the cut settled: where the next batch's march is issued (beside the backward, or beside the collectives)
        placements = {}
        for at in ("forward", "mlp_backward", "collectives"):
            self.prefetch_at = at
            sync()
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(window):
                step_fn(i)
                i += 1
            sync()
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            placements[at] = float(dt.item()) / window * 1e3
        self.prefetch_at = min(placements, key=placements.get)
        self.tuned = {"chunks_ms_per_step": dict(timings), "sharded_ms_per_step": sharded_ms,
                      "mode": self.comm_mode, "prefetch_at_ms_per_step": placements}
        return timings

    def probe_comm_dtype(self, step_fn, dtype=torch.bfloat16, window=None, first_step=0):
        """Data parallel: ms per step over one window with the table gradient on the wire in `dtype` (the opt-in
        `comm_dtype`), for reporting next to the fp32 figure; the setting itself is restored.  None on one rank."""
        import time
        import torch.distributed as dist
        if self.avg is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        window = int(window or self.update_interval)
        dev = next(self.model.parameters()).device
        keep, self.comm_dtype = self.comm_dtype, dtype
        try:
            torch.cuda.synchronize(dev)
            dist.barrier()
            t0 = time.perf_counter()
            for i in range(first_step, first_step + window):
                step_fn(i)
            torch.cuda.synchronize(dev)
            dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        finally:
            self.comm_dtype = keep
        return float(dt.item()) / window * 1e3

    def _loss_slot(self):
        """Loss values land in a ring of device scalars, cleared once per lap: no loss kernels, no per-step fill (laps are
        counted on the ring's own cursor: steps of other kinds in between do not use slots)."""
        slot = self._loss_cursor % self._loss_ring.numel()
        self._loss_cursor += 1
        if slot == 0:
            self._loss_ring.zero_()
        return self._loss_ring[slot]

    def _route_signature(self, rays_o, rays_d, target, next_rays):
        from . import fused_network, fused_render, raymarching
        m = self.model

        def of(t):
            return (t.shape, t.dtype, t.device, t.is_contiguous(), t.requires_grad)
        nxt = None if next_rays is None or any(r is None for r in next_rays) else (of(next_rays[0]), of(next_rays[1]))
        return (of(rays_o), of(rays_d), of(target), nxt, m.training, m.cuda_ray, m.bg_radius, float(m.density_scale),
                torch.is_grad_enabled(), torch.is_autocast_enabled(), fused_render.ENABLED, fused_render.NATIVE_STEP,
                fused_network.ENABLED, raymarching._DEVICE, self.use_graphs, self.fp16, self.manual_mse, self.prefetch,
                self.prefetch_at, self.avg is None, self.fuse_table_adam, self.comm_chunks, self.comm_mode,
                self.comm_dtype, getattr(m, "graph_counter", None) is None, getattr(m, "disable_view_direction", None),
                id(self.opt))

    def _step_rgb_native(self, rays_o, rays_d, target, next_rays, data_parallel=False, checked=False):
        """The steady-state step as one library call (fused_render.train_step_native -> enerf_train_step_mse): the same
        launches in the same order as _step_rgb_manual's one-GPU route, issued from C.  data_parallel: the call stops
        after the table's backward (dense gradient, no optimizer) and one of the data-parallel tails takes over -- the
        native one (two more library calls) when the library has its communicator."""
        from . import fused_render
        m = self.model
        emb = m._modules["encoder"]._parameters["embeddings"]
        if not (self._cleared_grad is not None and emb.grad is self._cleared_grad):
            emb.grad = None                     # only a buffer the last flush left clean may be added into
        self._cleared_grad = None
        if not checked:                         # (decided by the full checks: later steps of the same signature skip them)
            self._native_route_sig, self._native_route_dp = self._pending_sig, data_parallel
        nxt = None
        if (next_rays is not None and all(r is not None for r in next_rays) and self.prefetch
                and self.global_step % self.update_interval != 0
                and (checked or fused_render.supported(m, next_rays[0].contiguous().view(-1, 3),
                                                       next_rays[1].contiguous().view(-1, 3), 1, 0))):
            if self._side is None:
                self._side = torch.cuda.Stream()
            nxt = next_rays
        if (nxt is None and self.early_budget and self.prefetch and self.global_step % self.update_interval == 0
                and getattr(m, "_last_march_event", None) is None and not data_parallel):
            # the window's last step (the next call starts with update_extra_state): where the marches ride on this
            # stream, the ring of step counters is copied out here, a whole step before the update wants the budget
            fused_render.stage_ring_copy(m)
        loss = self._loss_slot()
        own = self._owner_range() if data_parallel else None
        if own is not None:
            from . import _lib as L
            L.check(L.lib().enerf_grid_owner_range(own[0], own[1], 1.0 / own[2]), "grid_owner_range")
        try:
            out = fused_render.train_step_native(m, rays_o, rays_d, target, self.opt, next_rays=nxt,
                                                 side_stream=self._side, loss_out=loss, perturb=self.perturb,
                                                 raw=data_parallel, defer_dp=own is not None)
        except BaseException:
            if own is not None:
                L.lib().enerf_grid_owner_range(0, 0, 1.0)
            self._discard_pending_records()
            raise
        if data_parallel:
            self._raw_grads = (None, out[1])            # (the table's gradient sits in embeddings.grad)
            if own is not None:
                self._finish_sharded_fused(own, None)
                return loss
            tail = self._finish_sharded if self.comm_mode == "sharded" else self._finish_distributed
            tail(None)                                  # the next batch's march is already queued (behind the MLP backward)
            return loss
        self._cleared_grad = emb.grad
        return loss

    def _step_rgb_manual(self, rays_o, rays_d, target, next_rays, **render_kw):
        chunked = (self.avg is not None and isinstance(self.avg, GradAverager) and hasattr(self.opt, "step_now")
                   and self.comm_chunks > 0)
        if (self.native_step and not render_kw and not self.use_graphs and self.prefetch_at == "mlp_backward"
                and getattr(self.model, "graph_counter", None) is None
                and ((self.avg is None and self.fuse_table_adam) or chunked)):
            from . import fused_render
            if fused_render.native_step_supported(self.model, rays_o.contiguous().view(-1, 3),
                                                  rays_d.contiguous().view(-1, 3), self.opt, data_parallel=chunked):
                return self._step_rgb_native(rays_o, rays_d, target, next_rays, data_parallel=chunked)
        side = self._side_prefetch(next_rays) if not render_kw else None
        late = chunked and side is not None and self.prefetch_at == "collectives"
        # one GPU: nobody but Adam reads the table's gradient, so the backward leaves it as record lists and the
        # optimizer's pass over the table sums them tile by tile in LDS (FusedAdam.step_grid_table)
        fuse_table = (self.fuse_table_adam and self.avg is None and hasattr(self.opt, "step_grid_table")
                      and not self.use_graphs)
        own = self._owner_range() if chunked else None
        if own is not None:
            from . import _lib as L
            L.check(L.lib().enerf_grid_owner_range(own[0], own[1], 1.0 / own[2]), "grid_owner_range")
        try:
            tail_side = side if (not late and self.prefetch_at == "mlp_backward") else None
            loss = self._manual_fwd_bwd(rays_o, rays_d, target,
                                        after_forward=None if (late or tail_side is not None) else side, raw=chunked,
                                        defer_table=fuse_table or own is not None, after_mlp_backward=tail_side,
                                        **render_kw)
        except BaseException:
            if own is not None:
                L.lib().enerf_grid_owner_range(0, 0, 1.0)
            self._discard_pending_records()
            raise
        if chunked:
            if own is not None:
                self._finish_sharded_fused(own, side if late else None)
                return loss
            tail = self._finish_sharded if self.comm_mode == "sharded" else self._finish_distributed
            tail(side if late else None)
            return loss
        self._reduce_grads(None if side is not None else next_rays)
        if fuse_table:
            enc = self.model._modules["encoder"]
            emb = enc._parameters["embeddings"]
            self.opt.step_grid_table(emb, enc._buffers["offsets"], enc.level_dim, extra=self._small_params)
            self._cleared_grad = emb.grad
        elif self.avg is None and hasattr(self.opt, "step_now") and not self.use_graphs:
            self.opt.step_now(zero_grads=True)          # Adam clears what it has read: the next step needs no fill
            self._cleared_grad = self.model.encoder.embeddings.grad
        else:
            self._opt_step()
        return loss

    def step_rgb(self, rays_o, rays_d, target, next_rays=None, **render_kw):
        """One RGB training step (nerf/utils.py:575-640 train_step + the optimizer part of train_one_epoch)."""
        if self.amp_bf16 or self.amp_f16:
            prev = self._amp_scope()
            try:
                if self.amp_f16 and self._amp_closed_form_ok() and self._manual_ok(rays_o, rays_d, target, render_kw):
                    loss = self._amp_step(self._step_rgb, rays_o, rays_d, target, next_rays, **render_kw)
                elif self.amp_f16:              # a step the closed form does not serve: the autocast route, same scaler
                    self.fp16 = True
                    try:
                        loss = self._step_rgb(rays_o, rays_d, target, next_rays, **render_kw)
                    finally:
                        self.fp16 = False
                else:
                    loss = self._step_rgb(rays_o, rays_d, target, next_rays, **render_kw)
            finally:
                self._amp_restore(prev)
        else:
            loss = self._step_rgb(rays_o, rays_d, target, next_rays, **render_kw)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return loss

    def _step_rgb(self, rays_o, rays_d, target, next_rays=None, **render_kw):
        if not self.model.training:                 # Module.train() walks every submodule: 40 us of a 900 us step
            self.model.train()
        coming = None
        if (self.global_step % self.update_interval == 0 and self.overlap_update and not render_kw and not self.fp16
                and self._manual_ok(rays_o, rays_d, target, render_kw)):
            coming = (rays_o, rays_d)
        self.maybe_update_extra_state(coming)
        self.global_step += 1
        if self._graphable(rays_o, rays_d):
            m = self.model
            key = self._graph_key("rgb", rays_o)
            if self._graph_for(key) is None:
                manual = self._manual_ok(rays_o, rays_d, target, render_kw)
                self._graphs[key] = self._capture(
                    (rays_o, rays_d, target),
                    lambda ro, rd, tg: torch.nn.functional.mse_loss(
                        m.render(ro, rd, staged=False, bg_color=None, perturb=self.perturb, **render_kw)["image"], tg),
                    (lambda ro, rd, tg: self._manual_fwd_bwd(ro, rd, tg, **render_kw)) if manual else None)
            return self._replay(self._graphs[key], (rays_o, rays_d, target), 1)
        if self.fp16:
            self.opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                out = self.model.render(rays_o, rays_d, staged=False, bg_color=None, perturb=self.perturb, **render_kw)
                loss = torch.nn.functional.mse_loss(out["image"], target)
            self.scaler.scale(loss).backward()
            self.scaler.unscale_(self.opt)
            self._reduce_grads(next_rays)
            self.scaler.step(self.opt)
            self.scaler.update()
            return loss.detach()
        # everything the route decision below reads, as one tuple: a step whose tuple equals that of the last step that
        # went the one-call route skips the checks (five `supported` walks, ~60 us of a 0.3 ms host budget)
        sig = None
        if self.native_step and not render_kw and (self.model.mean_count > 0 or getattr(self.model, "_cold_rows", 0) > 0):
            sig = self._route_signature(rays_o, rays_d, target, next_rays)
            if sig == self._native_route_sig:
                return self._step_rgb_native(rays_o, rays_d, target, next_rays, data_parallel=self._native_route_dp,
                                             checked=True)
        self._pending_sig = sig
        if self._manual_ok(rays_o, rays_d, target, render_kw):
            return self._step_rgb_manual(rays_o, rays_d, target, next_rays, **render_kw)
        self.opt.zero_grad(set_to_none=True)
        out = self.model.render(rays_o, rays_d, staged=False, bg_color=None, perturb=self.perturb, **render_kw)
        loss = torch.nn.functional.mse_loss(out["image"], target)
        loss.backward()
        self._reduce_grads(next_rays)
        self.opt.step()
        return loss.detach()

    def step_events(self, data, opt, next_data=None):
        """One event training step: two renders sharing one backward (nerf/utils.py:482-573)."""
        if self.amp_bf16 or self.amp_f16:
            prev = self._amp_scope()
            try:
                if self.amp_f16 and self._amp_closed_form_ok() and opt.event_only and self._events_manual_ok(data, opt):
                    loss = self._amp_step(self._step_events, data, opt, next_data)
                elif self.amp_f16:
                    self.fp16 = True
                    try:
                        loss = self._step_events(data, opt, next_data)
                    finally:
                        self.fp16 = False
                else:
                    loss = self._step_events(data, opt, next_data)
            finally:
                self._amp_restore(prev)
        else:
            loss = self._step_events(data, opt, next_data)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        return loss

    def _step_events(self, data, opt, next_data=None):
        from .events import train_step_events
        if not self.model.training:
            self.model.train()
        coming = None
        if (self.global_step % self.update_interval == 0 and self.overlap_update and not self.fp16 and opt.event_only
                and not opt.render_kwargs and self._events_manual_ok(data, opt)):
            # (the first render's count pass only: the marcher's chunk log holds ONE outstanding count pass, the second
            # render is marched whole once the first's write pass has run)
            coming = (data["rays_evs_o1"], data["rays_evs_d1"])
        self.maybe_update_extra_state(coming)
        self.global_step += 1
        from .events import wants_no_event_term
        no_ev = wants_no_event_term(opt)                 # two more renders per step: neither graphs nor the one-call step
        if opt.event_only and not no_ev and self._graphable(data["rays_evs_o1"], data["rays_evs_d1"]):
            names = ("images", "rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2", "pols")
            key = self._graph_key("events", data["rays_evs_o1"])
            inputs = tuple(data[n] for n in names)
            if self._graph_for(key) is None:
                self._graphs[key] = self._capture(
                    inputs, lambda *ts: train_step_events(self.model, dict(zip(names, ts)), opt)[0])
            return self._replay(self._graphs[key], inputs, 2)
        if self.fp16:
            # the shipped configs' fp16 = True around the event step (nerf/utils.py:964-975): autocast + GradScaler
            self.opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss, _ = train_step_events(self.model, data, opt)
            self.scaler.scale(loss).backward()
            self.scaler.unscale_(self.opt)
            if self.avg is not None:
                self.avg()
            self.scaler.step(self.opt)
            self.scaler.update()
            return loss.detach()
        if self._events_manual_ok(data, opt):
            from .events import train_step_events_manual
            if (self.native_step and self.fuse_table_adam and self.avg is None and not self.use_graphs and not no_ev
                    and self.prefetch_at == "mlp_backward" and hasattr(self.opt, "grid_table_plan")
                    and getattr(self.model, "graph_counter", None) is None):
                from . import fused_render
                if fused_render.native_events_supported(self.model, data, opt, self.opt):
                    return self._step_events_native(data, opt, next_data)
            side = None
            if next_data is not None and not opt.render_kwargs:
                side = self._side_prefetch((next_data["rays_evs_o1"], next_data["rays_evs_d1"]),
                                           (next_data["rays_evs_o2"], next_data["rays_evs_d2"]))
            fuse_table = (self.fuse_table_adam and self.avg is None and hasattr(self.opt, "step_grid_table")
                          and not self.use_graphs)
            emb = self.model.encoder.embeddings
            if emb.grad is not self._cleared_grad:          # only a buffer the last flush left clean may be added into
                emb.grad = None
            self._cleared_grad = None
            try:
                tail = self.prefetch_at == "mlp_backward"
                loss, _ = train_step_events_manual(self.model, data, opt, after_forward=None if tail else side,
                                                   after_mlp_backward=side if tail else None, defer_table=fuse_table)
            except BaseException:
                self._discard_pending_records()
                raise
            self._reduce_grads()
            if fuse_table:
                enc = self.model.encoder
                self.opt.step_grid_table(enc.embeddings, enc.offsets, enc.level_dim, extra=self._small_params)
                self._cleared_grad = enc.embeddings.grad
            else:
                self._opt_step()
            return loss
        self.opt.zero_grad(set_to_none=True)
        loss, _ = train_step_events(self.model, data, opt)
        loss.backward()
        if self.avg is not None:
            self.avg()
        self.opt.step()
        return loss.detach()

    def _step_events_native(self, data, opt, next_data):
        """The steady-state event-only step as one library call (fused_render.train_step_events_native ->
        enerf_train_step_events): the launches of the manual route below, in its order, issued from C."""
        from . import fused_render
        m = self.model
        emb = m._modules["encoder"]._parameters["embeddings"]
        if not (self._cleared_grad is not None and emb.grad is self._cleared_grad):
            emb.grad = None                     # only a buffer the last flush left clean may be added into
        self._cleared_grad = None
        nxt = None
        if (next_data is not None and self.prefetch and self.global_step % self.update_interval != 0
                and all(fused_render.supported(m, next_data[o].contiguous().view(-1, 3),
                                               next_data[d].contiguous().view(-1, 3), 1, 0)
                        for o, d in (("rays_evs_o1", "rays_evs_d1"), ("rays_evs_o2", "rays_evs_d2")))):
            if self._side is None:
                self._side = torch.cuda.Stream()
            nxt = next_data
        if (nxt is None and self.early_budget and self.prefetch and self.global_step % self.update_interval == 0
                and getattr(m, "_last_march_event", None) is None):
            fused_render.stage_ring_copy(m)     # (the window's last step, marches on this stream: see _step_rgb_native)
        try:
            loss, _ = fused_render.train_step_events_native(m, data, opt, self.opt, next_data=nxt, side_stream=self._side)
        except BaseException:
            self._discard_pending_records()
            raise
        self._cleared_grad = emb.grad
        return loss

    def _events_manual_ok(self, data, opt):
        from . import fused_render
        m = self.model
        ro, rd = data["rays_evs_o1"], data["rays_evs_d1"]
        return (self.manual_mse and opt.event_only and m.cuda_ray
                and not set(opt.render_kwargs) - {"dt_gamma", "max_steps", "out_dim_color"}
                and fused_render.supported(m, ro.contiguous().view(-1, 3), rd.contiguous().view(-1, 3), 1,
                                           opt.render_kwargs.get("dt_gamma", 0)))
