"""One autograd node for `NeRFNetwork.forward` on the MI355X fp32 path (nerf/network.py:104-132 of the reference):

    x -> (x + bound) / (2 bound) -> hash grid -> sigma MLP -> [trunc_exp | geo_feat]
    d -> SH -------------------------------------+-> colour MLP -> sigmoid

The kernels are the library's own entry points (grid_encode_forward/backward, sh_encode_forward, enerf_mlp32_*); what
this node removes is the glue between them -- the normalisation kernels, the permute of the encoding, trunc_exp and
sigmoid as separate elementwise passes (forward and backward), the [SH | geo_feat] concatenation, the slicing /
re-packing of gradients -- about two dozen small launches and as many torch dispatches per call, which is what bounds
a 4096-ray step once the big kernels are fast.  Data flow:

  * the grid kernel normalises its input itself (affine = (bound, 1/(2 bound)), rounded like torch's two kernels) and
    writes the level-major [16, Bp, 2] tensor the sigma MLP reads;
  * the sigma MLP writes its 16 outputs into columns 0..15 of a [B, 32] buffer and exp(column 0) into `sigma`;
    the SH kernel writes its 16 outputs into columns 16..31 of the same buffer, which is the colour MLP's input
    (its first-layer weight columns are permuted to match: column 0 -- the raw density -- gets zero weight);
  * the colour MLP applies the sigmoid; its backward takes d(rgb) and the saved rgb, the sigma MLP's backward takes
    columns 0..15 of the colour MLP's input gradient in place (row stride 32) with column 0 replaced by
    d(sigma) * exp(clamp(h0, -15, 15)), and hands its input gradient to the grid backward in level-major order.

Same parameters (nn.Linear weights, `encoder.embeddings`), same values to fp32 round-off (the colour net's first layer
sums its 31 products in a different order).  Anything it does not cover (CPU tensors, autocast, other widths, inputs
that need gradients, disable_view_direction) takes the unfused route in network.py.
"""
import ctypes

import numpy as np
import torch
from torch.autograd import Function

from . import _lib as L
from . import gridencoder as _ge
from .backends import _gridencoder as _gb
from .fused_mlp import pad32

ENABLED = True




def _architecture_kind(net):
    """"linear": nerf/network.py (nn.Linear nets, fp32: sigma 32-64-16, colour 31-64-64-out); "ff": nerf/network_ff.py
    (FFMLP nets with flat fp32 master weights computed in bf16: sigma 32-64-64-16, colour 32-64-64-64-3); None: neither."""
    from .gridencoder import GridEncoder
    from .shencoder import SHEncoder
    enc, encd = getattr(net, "encoder", None), getattr(net, "encoder_dir", None)
    if not (isinstance(enc, GridEncoder) and enc.num_levels == 16 and enc.level_dim == 2 and enc.input_dim == 3
            and enc.embeddings.dtype == torch.float32):
        return None
    if not (isinstance(encd, SHEncoder) and encd.degree == 4):
        return None
    s, c = getattr(net, "sigma_net", None), getattr(net, "color_net", None)
    if isinstance(s, torch.nn.ModuleList) and isinstance(c, torch.nn.ModuleList):
        if not all(m.weight.is_contiguous() and m.weight.dtype == torch.float32 and m.bias is None
                   for m in list(s) + list(c)):
            return None
        ok = (len(s) == 2 and tuple(s[0].weight.shape) == (64, 32) and tuple(s[1].weight.shape) == (16, 64)
              and len(c) == 3 and tuple(c[0].weight.shape) == (64, 31) and tuple(c[1].weight.shape) == (64, 64)
              and c[2].weight.shape[1] == 64 and c[2].weight.shape[0] <= 32)
        return "linear" if ok else None
    from .ffmlp import FFMLP
    if isinstance(s, FFMLP) and isinstance(c, FFMLP):
        relu, none = 0, 6
        ok = (s.input_dim == 32 and s.hidden_dim == 64 and s.num_layers == 2 and s.output_dim == 16
              and c.input_dim == 32 and c.hidden_dim == 64 and c.num_layers == 3 and c.output_dim == 3
              and int(s.activation) == relu and int(c.activation) == relu and int(s.output_activation) == none
              and int(c.output_activation) == none and s.compute_dtype == torch.bfloat16
              and c.compute_dtype == torch.bfloat16 and s.weights.dtype == torch.float32
              and c.weights.dtype == torch.float32 and s.weights.is_contiguous() and c.weights.is_contiguous())
        return "ff" if ok else None
    return None


def kind_of(net):
    """Architecture served by this module ("linear" / "ff") or None; cached ON the net (with its embeddings parameter as
    the check key).  (It used to live in a module-level dict keyed by id(net): a model built right after another was
    deleted can get the same id -- and the same id for its embeddings -- and then inherited the dead model's kind: a
    network.py model was taken for a network_ff one, once in a few runs of bench.py's legs.)"""
    enc = net._modules.get("encoder")
    emb = enc._parameters.get("embeddings") if enc is not None else None
    key = (id(emb), emb.dtype) if emb is not None else (0, None)
    hit = net.__dict__.get("_fused_kind")
    if hit is None or hit[0] != key:
        hit = net.__dict__["_fused_kind"] = (key, _architecture_kind(net))
    return hit[1]


def supported(net, x, d):
    if not (ENABLED and x.is_cuda and x.dtype == torch.float32 and d.dtype == torch.float32 and x.dim() == 2
            and d.dim() == 2 and not torch.is_autocast_enabled() and not getattr(net, "disable_view_direction", True)):
        return False
    if x.requires_grad or d.requires_grad:
        return False
    # (registered sub-modules / parameters are read from the module's own dicts: nn.Module.__getattr__ is a slow path,
    # and this runs several times per 0.5 ms training step)
    return kind_of(net) is not None and _ge._supports_layout()


# per architecture: hidden layers of the two nets, floats per first-layer row of the colour net in memory, arithmetic
# mode of the mlp32 kernels (enerf_mlp32_precision; None = whatever the process runs, i.e. the fp32 / split-bf16 default)
_ARCH = {"linear": dict(nh_s=1, nh_c=2, w0c=31, prec=None), "ff": dict(nh_s=2, nh_c=3, w0c=32, prec=2)}
_FF_SIGMA, _FF_COLOR = 64 * (32 + 64 + 16), 64 * (32 + 2 * 64 + 16)      # FFMLP blobs: [W0 | Wh ... | Wout padded to 16 rows]


def _kind_of_weights(weights):
    return "ff" if len(weights) == 2 else "linear"


def _weight_segments(kind, weights):
    """-> (sigma net, colour net) pointer arrays {first layer, hidden 0, hidden 1, output layer} of enerf_mlp32_*_p."""
    if kind == "linear":
        ws0, ws1, wc0, wc1, wc2 = weights
        return _segments(ws0, None, None, ws1), _segments(wc0, wc1, None, wc2)
    ws, wc = weights                                   # flat FFMLP blobs (ffmlp/ffmlp.py:115-121)
    ps, pc = ws.data_ptr(), wc.data_ptr()
    return ((ctypes.c_void_p * 4)(ps, ps + 4 * 2048, None, ps + 4 * (2048 + 4096)),
            (ctypes.c_void_p * 4)(pc, pc + 4 * 2048, pc + 4 * (2048 + 4096), pc + 4 * (2048 + 8192)))


class _precision:
    """with _precision(mode): the mlp32 kernels' arithmetic for the calls inside (None: leave it alone)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = L.lib().enerf_mlp32_precision(self.mode) if self.mode is not None else None

    def __exit__(self, *exc):
        if self.mode is not None:
            L.lib().enerf_mlp32_precision(self.prev)
        return False


def nerf_forward(x, d, cfg, train, embeddings, offsets, *weights, out=None, valid_rows=None):
    """The kernel sequence itself (no autograd): returns sigma [B], rgb [B,out], and -- when `train` -- the tensors
    nerf_backward needs.  `weights`: the five nn.Linear weights of nerf/network.py (ws0, ws1, wc0, wc1, wc2) or the two
    flat FFMLP blobs of nerf/network_ff.py (sigma net, colour net).  `out` = (sigma [B], rgb [B,out]) contiguous fp32
    tensors to write into (e.g. slices of a frame's sample buffers).  `valid_rows`: device int32 tensor whose first
    element is the number of real rows (the march's counter): rows beyond it are the sample budget's padding, which the
    MLP kernels then skip (enerf_mlp32_valid_rows) -- their sigma / rgb stay unwritten and must not be read."""
    bound, per_level_scale, base_resolution, gridtype = cfg[:4]
    kind = _kind_of_weights(weights)
    arch = _ARCH[kind]
    if len(cfg) > 4 and cfg[4] is not None:          # the model asks for an arithmetic of its own (NeRFNetwork.mlp_precision)
        arch = dict(arch, prec=int(cfg[4]))
    x = x.contiguous()
    d = d.contiguous()
    B = x.shape[0]
    Bp = pad32(B)
    dev = x.device
    lib = L.lib()
    out_c = weights[-1].shape[0] if kind == "linear" else 3
    if out is None:
        sigma = torch.empty(B, dtype=torch.float32, device=dev)
        rgb = torch.empty(B, out_c, dtype=torch.float32, device=dev)
    else:
        sigma, rgb = out
        assert sigma.shape == (B,) and rgb.shape == (B, out_c) and sigma.is_contiguous() and rgb.is_contiguous() \
            and sigma.dtype == rgb.dtype == torch.float32
    if B == 0:
        return sigma, rgb, None
    S = float(np.log2(per_level_scale))
    affine = (float(bound), float(np.float32(1.0) / np.float32(2 * bound)))   # torch: (x + b) * (1.0f / (2b))
    emb = embeddings.contiguous()
    feats = torch.empty(16, Bp, 2, dtype=torch.float32, device=dev)
    # dy_dx is not computed (calc_grad_inputs = False): any tensor serves as the placeholder
    _gb.grid_encode_forward(x, emb, offsets, feats, B, 3, 2, 16, S, base_resolution, False, feats, gridtype,
                            layout=2, affine=affine)
    stream = L.stream_handle()
    # the MLP kernels stage the weights straight from the parameters (the colour net's first-layer column order
    # [SH | geo_feat (| pad)] -> [0 | geo_feat | SH] is applied on the way into LDS): nothing is packed per step
    seg_s, seg_c = _weight_segments(kind, weights)
    nh_s, nh_c = arch["nh_s"], arch["nh_c"]
    h32 = fb_s = fb_c = None
    fused = False
    if valid_rows is not None:
        lib.enerf_mlp32_valid_rows(valid_rows.data_ptr())
    try:
        with _precision(arch["prec"]):
            # nerf/network.py's nets in the split-bf16 default: both as ONE launch (csrc/nerf_mlp.hip) -- the sigma net's
            # outputs and the SH basis reach the colour net in registers, no [B, 32] hand-over tensor
            fused = kind == "linear" and out_c <= 16 and bool(lib.enerf_nerf_mlp_available())
            if fused:
                L.check(lib.enerf_nerf_mlp_forward(feats.data_ptr(), d.data_ptr(), seg_s, seg_c, arch["w0c"], B, out_c,
                                                   sigma.data_ptr(), rgb.data_ptr(), 0, stream), "nerf_mlp_forward")
            else:
                h32 = torch.empty(B, 32, dtype=torch.float32, device=dev)
                fb_s = torch.empty(nh_s, Bp, 64, dtype=torch.float32, device=dev) if train else None
                # the sigma kernel also fills the SH columns 16..31 of h32 from the directions (no separate encoder launch)
                L.check(lib.enerf_mlp32_forward_p(feats.data_ptr(), seg_s, 32, 0, B, 32, 16, nh_s, 0, 6,
                                                  fb_s.data_ptr() if train else None, h32.data_ptr(), 1, 32,
                                                  sigma.data_ptr(), d.data_ptr(), stream), "mlp32_forward_p(sigma)")
                # (colour net input columns: [raw density (zero weight) | geo_feat 15 | SH 16])
                fb_c = torch.empty(nh_c, Bp, 64, dtype=torch.float32, device=dev) if train else None
                L.check(lib.enerf_mlp32_forward_p(h32.data_ptr(), seg_c, arch["w0c"], 1, B, 32, out_c, nh_c, 0, 3,
                                                  fb_c.data_ptr() if train else None, rgb.data_ptr(), 0, 0, None, None,
                                                  stream), "mlp32_forward_p(color)")
    finally:                                 # the row count is per call: never left behind for another model's launch
        if valid_rows is not None:
            lib.enerf_mlp32_valid_rows(None)
    saved = None
    if train:
        saved = dict(x=x, d=d, emb=emb, offsets=offsets, feats=feats, h32=h32, fb_s=fb_s, fb_c=fb_c, seg_s=seg_s,
                     seg_c=seg_c, weights=weights, rgb=rgb, B=B, S=S, H=base_resolution, gridtype=gridtype, kind=kind,
                     arch=arch, affine=affine, out_c=out_c, param=embeddings, valid_rows=valid_rows, fused=fused)
    return sigma, rgb, saved


def _segments(w0, h0, h1, wout):
    """ctypes {first layer, hidden 0, hidden 1, output layer} pointer array of enerf_mlp32_*_p."""
    return (ctypes.c_void_p * 4)(*[None if t is None else t.data_ptr() for t in (w0, h0, h1, wout)])


# flat gradient buffer of the five MLP weights, in parameter order: ws0 | ws1 | wc0 [64,31] | wc1 | wc2 [out_c,64]
_DW_OFFSETS = (0, 2048, 3072, 5056, 9152)


def unpack_weight_grads(dw, out_c, kind="linear"):
    """The MLP weight gradients (views of the backward's flat buffer, which is laid out in parameter order): five
    matrices for the nn.Linear nets, the two flat blobs for the FFMLP nets."""
    if kind == "ff":
        return (dw[:_FF_SIGMA], dw[_FF_SIGMA:])
    o = _DW_OFFSETS
    return (dw[o[0]:o[1]].view(64, 32), dw[o[1]:o[2]].view(16, 64), dw[o[2]:o[3]].view(64, 31),
            dw[o[3]:o[4]].view(64, 64), dw[o[4]:].view(out_c, 64))


def _grad_segments(kind, dev, out_c):
    """-> the flat dW buffer and the two pointer arrays the backward's reduce pass writes through."""
    if kind == "ff":
        # the blobs' unused parts (colour net: output rows 3..15, the pad column of the first layer) get no gradient
        dw = torch.zeros(_FF_SIGMA + _FF_COLOR, dtype=torch.float32, device=dev)
        ps = dw.data_ptr()
        pc = ps + 4 * _FF_SIGMA
        return dw, ((ctypes.c_void_p * 4)(ps, ps + 4 * 2048, None, ps + 4 * (2048 + 4096)),
                    (ctypes.c_void_p * 4)(pc, pc + 4 * 2048, pc + 4 * (2048 + 4096), pc + 4 * (2048 + 8192)))
    dw = torch.empty(_DW_OFFSETS[4] + 64 * out_c, dtype=torch.float32, device=dev)
    o = _DW_OFFSETS
    p0 = dw.data_ptr()
    return dw, ((ctypes.c_void_p * 4)(p0 + 4 * o[0], None, None, p0 + 4 * o[1]),
                (ctypes.c_void_p * 4)(p0 + 4 * o[2], p0 + 4 * o[3], None, p0 + 4 * o[4]))


def nerf_backward(sv, g_sigma, g_rgb, sigma_scale=1.0, raw=False, owner=False, defer_table=0, after_mlp=None):
    """Gradients of (embeddings, *weights) given d(sigma) [B] and d(rgb) [B,out] (contiguous fp32).
    `sigma_scale` multiplies d(sigma) on the fly (the renderer's density_scale).  The embedding gradient is None when
    it was added straight into the parameter's .grad (`owner`: the caller drives this backward itself, outside autograd,
    and owns that buffer; under autograd the shortcut is opt-in, gridencoder.ACCUMULATE_INTO_PARAM_GRAD).  raw=True: -> (embedding gradient, flat dW accumulator) -- what a
    data-parallel caller all-reduces (two buffers) before unpack_weight_grads."""
    B, out_c, kind = sv["B"], sv["out_c"], sv["kind"]
    arch = sv["arch"]
    nh_s, nh_c = arch["nh_s"], arch["nh_c"]
    Bp = pad32(B)
    dev = sv["x"].device
    lib = L.lib()
    stream = L.stream_handle()
    fused = bool(sv.get("fused"))
    if sigma_scale != 1.0 and not fused:
        g_sigma = g_sigma * sigma_scale
    # weight gradients: written (not accumulated) by the backward's reduce pass, straight in parameter order
    dw, (dseg_s, dseg_c) = _grad_segments(kind, dev, out_c)
    dfeat = torch.empty(16, Bp, 2, dtype=torch.float32, device=dev)
    if sv.get("valid_rows") is not None:     # the forward skipped the budget's padding rows: so must the backward
        lib.enerf_mlp32_valid_rows(sv["valid_rows"].data_ptr())
    if after_mlp is not None:                # its side stream waits for the reduce launch's own completion signal
        lib.enerf_mlp32_signal_next_reduce(1)
    try:
        with _precision(arch["prec"]):
            if fused:
                # (the forward of this very step built the operand fragments from these weights: flags = 1)
                L.check(lib.enerf_nerf_mlp_backward(g_rgb.data_ptr(), g_sigma.data_ptr(), float(sigma_scale),
                                                    sv["feats"].data_ptr(), sv["d"].data_ptr(), sv["rgb"].data_ptr(),
                                                    sv["seg_s"], sv["seg_c"], dseg_s, dseg_c, arch["w0c"], 1, B, out_c,
                                                    dfeat.data_ptr(), 1, stream), "nerf_mlp_backward")
            else:
                # (scratch of the unfused dgrad / wgrad kernels only: the fused backward never touches it)
                bb_c = torch.empty(nh_c if kind == "linear" else 0, Bp, 64, dtype=torch.float32, device=dev)
                dx32 = torch.empty(B, 32, dtype=torch.float32, device=dev)
                lib.enerf_mlp32_defer_reduce(1)      # the colour net's dW partial sums are reduced by the sigma net's launch
                L.check(lib.enerf_mlp32_backward_p(g_rgb.data_ptr(), sv["h32"].data_ptr(), sv["seg_c"], dseg_c, arch["w0c"],
                                                   1, 1, sv["fb_c"].data_ptr(), B, 32, out_c, nh_c, 0, bb_c.data_ptr(),
                                                   dx32.data_ptr(), 0, 0, sv["rgb"].data_ptr(), out_c, None, None, 0,
                                                   stream), "mlp32_backward_p(color)")
                bb_s = torch.empty(nh_s if kind == "linear" else 0, Bp, 64, dtype=torch.float32, device=dev)
                L.check(lib.enerf_mlp32_backward_p(dx32.data_ptr(), sv["feats"].data_ptr(), sv["seg_s"], dseg_s, 32, 0, 1,
                                                   sv["fb_s"].data_ptr(), B, 32, 16, nh_s, 0, bb_s.data_ptr(),
                                                   dfeat.data_ptr(), 1, 32, None, 0, g_sigma.data_ptr(),
                                                   sv["h32"].data_ptr(), 32, stream), "mlp32_backward_p(sigma)")
    finally:
        lib.enerf_mlp32_defer_reduce(0)
        lib.enerf_mlp32_signal_next_reduce(0)
        if sv.get("valid_rows") is not None:
            lib.enerf_mlp32_valid_rows(None)
    if after_mlp is not None:
        # both MLP backward kernels and their reduce launch are queued (the latter carries the signal armed above: see
        # enerf_stream_wait_mlp32_signal); the table's backward follows
        after_mlp(signalled=True)
    param, emb = sv["param"], sv["emb"]
    target = _ge.param_grad_target(param, torch.float32, owner=owner)
    direct = target is not None
    g_emb = target if direct else torch.zeros_like(emb)
    # defer_table = total samples of the step's renders: the table gradient's record lists are left for the optimizer's
    # fused flush (FusedAdam.step_grid_table) instead of being summed into g_emb
    _gb.grid_encode_backward(dfeat, sv["x"], emb, sv["offsets"], g_emb, B, 3, 2, 16, sv["S"], sv["H"], False, dfeat,
                             dfeat, sv["gridtype"], layout=2, affine=sv["affine"], defer=defer_table > 0,
                             reserve=defer_table)
    if raw:
        return (None if direct else g_emb, dw)
    return (None if direct else g_emb,) + unpack_weight_grads(dw, out_c, kind)


class _FusedNeRF(Function):
    @staticmethod
    def forward(ctx, x, d, cfg, train, embeddings, offsets, *weights):
        sigma, rgb, saved = nerf_forward(x, d, cfg, train, embeddings, offsets, *weights)
        ctx.sv = saved
        return sigma, rgb

    @staticmethod
    def backward(ctx, g_sigma, g_rgb):
        sv = ctx.sv
        dev = sv["x"].device
        B, out_c = sv["B"], sv["out_c"]
        g_sigma = torch.zeros(B, dtype=torch.float32, device=dev) if g_sigma is None else g_sigma.float().contiguous()
        g_rgb = torch.zeros(B, out_c, dtype=torch.float32, device=dev) if g_rgb is None else g_rgb.float().contiguous()
        g = nerf_backward(sv, g_sigma, g_rgb)
        return (None, None, None, None, g[0], None) + g[1:]


def network_params(net):
    """(embeddings, *MLP weights) in the order nerf_forward / nerf_backward use."""
    mods = net._modules
    s, c = mods["sigma_net"], mods["color_net"]
    emb = mods["encoder"]._parameters["embeddings"]
    if "weights" in s._parameters:                     # FFMLP: one flat parameter per net
        return (emb, s._parameters["weights"], c._parameters["weights"])
    s, c = s._modules, c._modules
    return (emb, s["0"]._parameters["weight"], s["1"]._parameters["weight"],
            c["0"]._parameters["weight"], c["1"]._parameters["weight"], c["2"]._parameters["weight"])


def network_cfg(net):
    """(bound, per-level scale, base resolution, grid type, arithmetic of the MLP kernels).  The last is
    `net.mlp_precision` (enerf_mlp32_precision's codes; None / absent = the architecture's own: the process-wide fp32 /
    split-bf16 default for the nn.Linear nets, bf16 operands for the FFMLP nets)."""
    enc = net._modules["encoder"]
    return (net.bound, enc.per_level_scale, enc.base_resolution, enc.gridtype_id, net.__dict__.get("mlp_precision"))


def encoder_offsets(net):
    return net._modules["encoder"]._buffers["offsets"]


def forward(net, x, d):
    """sigma [N], rgb [N, out_dim_color] for x [N,3] in [-bound, bound], d [N,3]."""
    params = network_params(net)
    train = torch.is_grad_enabled() and any(p.requires_grad for p in params)
    return _FusedNeRF.apply(x, d, network_cfg(net), train, params[0], encoder_offsets(net), *params[1:])


@torch.no_grad()
def forward_into(net, x, d, sigma_out, rgb_out):
    """Inference forward writing straight into caller-owned buffers (no autograd, no copies)."""
    params = network_params(net)
    nerf_forward(x, d, network_cfg(net), False, params[0], encoder_offsets(net), *params[1:], out=(sigma_out, rgb_out))


def density_sigma_sweep(net, n_cascades, grid_size, seed):
    """density_sigma at the query points of a full density-grid sweep (one jittered point per cell of every cascade,
    x fastest: csrc/sweep_points.h), which the grid kernel generates itself -- no position array is written or read.
    -> sigma [n_cascades * grid_size^3], in the sweep's order."""
    enc = net.encoder
    B = int(n_cascades) * int(grid_size) ** 3
    dev = enc.embeddings.device
    sigma = _density_scratch(net, "sigma", (B,), torch.float32, dev)
    Bp = pad32(B)
    S = float(np.log2(enc.per_level_scale))
    feats = _density_scratch(net, "feats", (16, Bp, 2), torch.float32, dev)
    _gb.grid_encode_forward_sweep(enc.embeddings.detach().contiguous(), enc.offsets, feats, n_cascades, grid_size,
                                  net.bound, seed, 2, 16, S, enc.base_resolution, enc.gridtype_id, 2,
                                  (float(net.bound), float(np.float32(1.0) / np.float32(2 * net.bound))))
    _sigma_only(net, feats, sigma, B)
    return sigma


def _density_scratch(net, name, shape, dtype, dev):
    """Buffers of update_extra_state that live from one update to the next (features of up to C x H^3 query points:
    0.4 - 0.8 GB, positions, indices).  Taken from torch's caching pool every 16 steps they were sometimes served and
    sometimes a hipMalloc / hipFree round in the middle of the step (the bench's partial-update leg: 0.36 or 0.45
    ms/step from run to run); 288 GB of HBM can afford to keep them."""
    pool = net.__dict__.setdefault("_density_scratch", {})
    n = 1
    for d in shape:
        n *= int(d)
    # the pool is not the caching allocator: nothing orders a new user's stream behind the previous user's.  Its buffers
    # belong to ONE stream at a time; a call from another stream (a side-stream update, a capture) waits for everything the
    # previous owner has queued before it reuses them.
    cur = torch.cuda.current_stream(dev)
    owner = pool.get("_stream")
    if owner is not None and owner != cur:
        cur.wait_stream(owner)
    pool["_stream"] = cur
    buf = pool.get(name)
    if buf is None or buf.numel() < n or buf.dtype != dtype or buf.device != dev:
        buf = torch.empty(max(n, 1), dtype=dtype, device=dev)
        pool[name] = buf
    return buf[:n].view(*shape)


def _sigma_only(net, feats, sigma, B):
    """sigma = exp(output 0) of the sigma net on level-major features (no other output written)."""
    kind = kind_of(net)
    arch = _ARCH[kind]
    if net.__dict__.get("mlp_precision") is not None:
        arch = dict(arch, prec=int(net.mlp_precision))
    seg = _weight_segments(kind, network_params(net)[1:])[0]
    with _precision(arch["prec"]):
        L.check(L.lib().enerf_mlp32_forward_p(feats.data_ptr(), seg, 32, 0, B, 32, 16, arch["nh_s"], 0, 6, None, None, 1, 0,
                                              sigma.data_ptr(), None, L.stream_handle()), "mlp32_forward_p(sigma only)")


def density_sigma(net, x):
    """sigma [N] only (no geo_feat, no autograd): what update_extra_state needs from density() for its 2 M cell
    samples per cascade -- the sigma MLP writes exp(column 0) and nothing else."""
    enc = net.encoder
    x = x.contiguous()
    B = x.shape[0]
    dev = x.device
    sigma = torch.empty(B, dtype=torch.float32, device=dev)      # (the caller's to keep; the features are scratch)
    if B == 0:
        return sigma
    Bp = pad32(B)
    S = float(np.log2(enc.per_level_scale))
    affine = (float(net.bound), float(np.float32(1.0) / np.float32(2 * net.bound)))
    feats = _density_scratch(net, "feats", (16, Bp, 2), torch.float32, dev)
    _gb.grid_encode_forward(x, enc.embeddings.detach().contiguous(), enc.offsets, feats, B, 3, 2, 16, S,
                            enc.base_resolution, False, feats, enc.gridtype_id, layout=2, affine=affine)
    _sigma_only(net, feats, sigma, B)
    return sigma
