"""Fully-fused MLP (MFMA, bf16 / fp16 storage, fp32 accumulate) through the C ABI vs the CPU oracle that rounds storage
at the same points, and vs an fp32 nn.Linear stack at 16-bit tolerance (the reference accumulates in fp16; parity for
this module is defined against the fp32 statement)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(i, k, B, seed):
    rng = np.random.default_rng(seed)
    nW = 64 * (i + 64 * (k - 1) + 16)
    W = (rng.uniform(-1, 1, nW) * np.sqrt(3 / 64) * 1.5).astype(np.float32)
    x = rng.uniform(-1, 1, (B, i)).astype(np.float32)
    g = rng.uniform(-1, 1, (B, 16)).astype(np.float32)
    return W, x, g


def _ulp_tol(ref, dtype, n_ulp=2.0):
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    return n_ulp * eps * np.maximum(np.abs(ref), 1e-3 if dtype == torch.bfloat16 else 1e-4)


@pytest.fixture
def buffered_ffmlp():
    """The reference's data flow for enerf_ffmlp_forward / _backward (forward_buffer / backward_buffer written and read);
    the default since round 4 recomputes the hidden activations in the backward and leaves both buffers alone."""
    from enerf_amd import _lib
    prev = _lib.lib().enerf_ffmlp_recompute(0)
    yield
    _lib.lib().enerf_ffmlp_recompute(prev)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,B", [(2, 256), (3, 640), (2, 100 * 128), (3, 128)])
def test_ffmlp_recomputing_entry_points_vs_oracle(dtype, k, B):
    """enerf_ffmlp_forward / _backward on the recomputing data flow (the default for the two nets of nerf/network_ff.py:
    input_dim 32, two / three hidden layers): outputs, input gradients and weight gradients against the oracle's rounded
    FFMLP, `forward_buffer` / `backward_buffer` untouched, and the same results as the buffered kernels'."""
    from enerf_amd import _lib
    from enerf_amd.backends import _ffmlp as ff
    assert _lib.lib().enerf_ffmlp_recompute(-1) == 1
    i = 32
    rnd = 1 if dtype == torch.bfloat16 else 2
    W, x, g = _mk(i, k, B, 300 + k)
    Wd = torch.from_numpy(W).to(DEV).to(dtype)
    xd = torch.from_numpy(x).to(DEV).to(dtype)
    gd = torch.from_numpy(g).to(DEV).to(dtype)
    Wr, xr, gr = Wd.float().cpu().numpy(), xd.float().cpu().numpy(), gd.float().cpu().numpy()
    out_ref, fb_ref = O.ffmlp_forward(xr, Wr, i, 16, 64, k, 0, 6, rnd=rnd)
    gi_ref, gw_ref, _ = O.ffmlp_backward(gr, xr, Wr, fb_ref, i, 16, 64, k, 0, True, rnd=rnd)

    def run():
        fb = torch.full((k, B, 64), 7.0, device=DEV, dtype=dtype)
        bb = torch.full((k, B, 64), 7.0, device=DEV, dtype=dtype)
        out = torch.empty(B, 16, device=DEV, dtype=dtype)
        gi = torch.zeros(B, i, device=DEV, dtype=dtype)
        gw = torch.zeros_like(Wd)
        ff.ffmlp_forward(xd, Wd, B, i, 16, 64, k, 0, 6, fb, out)
        ff.ffmlp_backward(gd, xd, Wd, fb, B, i, 16, 64, k, 0, 6, True, bb, gi, gw)
        return out, gi, gw, fb, bb
    out, gi, gw, fb, bb = run()
    assert float((fb - 7.0).abs().max()) == 0.0 and float((bb - 7.0).abs().max()) == 0.0       # neither written
    o = out.float().cpu().numpy()
    assert (np.abs(o - out_ref) <= _ulp_tol(out_ref, dtype, 4)).mean() > 0.995 and np.abs(o - out_ref).max() < 0.06
    gi_ = gi.float().cpu().numpy()
    assert (np.abs(gi_ - gi_ref) <= _ulp_tol(gi_ref, dtype, 4)).mean() > 0.985
    gw_ = gw.float().cpu().numpy()
    scale = np.abs(gw_ref).max()
    assert np.abs(gw_ - gw_ref).max() < 0.02 * scale
    np.testing.assert_allclose(gw_, gw_ref, rtol=0.05, atol=0.01 * scale)
    # the buffered kernels on the same tensors: same outputs (both round every layer to 16 bits at the same points)
    prev = _lib.lib().enerf_ffmlp_recompute(0)
    try:
        out_b, gi_b, gw_b, fb_b, _ = run()
    finally:
        _lib.lib().enerf_ffmlp_recompute(prev)
    assert float((fb_b - 7.0).abs().max()) > 0.0
    assert (np.abs(o - out_b.float().cpu().numpy()) <= _ulp_tol(out_ref, dtype, 2)).mean() > 0.999
    assert np.abs(gw_ - gw_b.float().cpu().numpy()).max() < 0.02 * scale
    # without grad_inputs
    gw2 = torch.zeros_like(Wd)
    ff.ffmlp_backward(gd, xd, Wd, fb, B, i, 16, 64, k, 0, 6, False, bb, torch.zeros(1, device=DEV, dtype=dtype), gw2)
    assert torch.equal(gw2, gw)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("i,k,B", [(32, 2, 256), (32, 3, 640), (16, 2, 128), (64, 4, 384)])
def test_ffmlp_forward_inference_backward_vs_oracle(dtype, i, k, B, buffered_ffmlp):
    from enerf_amd.backends import _ffmlp as ff
    rnd = 1 if dtype == torch.bfloat16 else 2
    W, x, g = _mk(i, k, B, 100 + i + k)
    Wd = torch.from_numpy(W).to(DEV).to(dtype)
    xd = torch.from_numpy(x).to(DEV).to(dtype)
    gd = torch.from_numpy(g).to(DEV).to(dtype)
    # oracle sees exactly the 16-bit values
    Wr, xr, gr = Wd.float().cpu().numpy(), xd.float().cpu().numpy(), gd.float().cpu().numpy()
    out_ref, fb_ref = O.ffmlp_forward(xr, Wr, i, 16, 64, k, 0, 6, rnd=rnd)

    fb = torch.empty(k, B, 64, device=DEV, dtype=dtype)
    out = torch.empty(B, 16, device=DEV, dtype=dtype)
    ff.ffmlp_forward(xd, Wd, B, i, 16, 64, k, 0, 6, fb, out)
    o = out.float().cpu().numpy()
    f = fb.float().cpu().numpy()
    assert (np.abs(f - fb_ref) <= _ulp_tol(fb_ref, dtype)).mean() > 0.999
    assert np.abs(f - fb_ref).max() < 0.05
    assert (np.abs(o - out_ref) <= _ulp_tol(out_ref, dtype, 4)).mean() > 0.995
    assert np.abs(o - out_ref).max() < 0.06

    out_inf = torch.empty(B, 16, device=DEV, dtype=dtype)
    ff.ffmlp_inference(xd, Wd, B, i, 16, 64, k, 0, 6, torch.empty(B, 64, device=DEV, dtype=dtype), out_inf)
    assert torch.equal(out_inf, out)

    # backward on the GPU's own forward buffer (so both sides mask identically)
    gi_ref, gw_ref, bb_ref = O.ffmlp_backward(gr, xr, Wr, f, i, 16, 64, k, 0, True, rnd=rnd)
    bb = torch.zeros(k, B, 64, device=DEV, dtype=dtype)
    gi = torch.zeros(B, i, device=DEV, dtype=dtype)
    gw = torch.zeros_like(Wd)
    ff.ffmlp_backward(gd, xd, Wd, fb, B, i, 16, 64, k, 0, 6, True, bb, gi, gw)
    b_ = bb.float().cpu().numpy()
    assert (np.abs(b_ - bb_ref) <= _ulp_tol(bb_ref, dtype, 4)).mean() > 0.995
    gi_ = gi.float().cpu().numpy()
    assert (np.abs(gi_ - gi_ref) <= _ulp_tol(gi_ref, dtype, 4)).mean() > 0.99
    # weight gradient: computed from the *rounded* backward buffer on the GPU vs the oracle's own chain
    gw_ = gw.float().cpu().numpy()
    scale = np.abs(gw_ref).max()
    assert np.abs(gw_ - gw_ref).max() < 0.02 * scale
    np.testing.assert_allclose(gw_, gw_ref, rtol=0.05, atol=0.01 * scale)
    # without grad_inputs
    gw2 = torch.zeros_like(Wd); bb2 = torch.zeros_like(bb)
    ff.ffmlp_backward(gd, xd, Wd, fb, B, i, 16, 64, k, 0, 6, False, bb2, torch.zeros(1, device=DEV, dtype=dtype), gw2)
    assert torch.equal(bb2, bb) and torch.equal(gw2, gw)


@pytest.mark.parametrize("act", [1, 3, 4, 5, 6])
def test_ffmlp_other_activations(act, buffered_ffmlp):
    from enerf_amd.backends import _ffmlp as ff
    i, k, B, dtype = 32, 2, 128, torch.bfloat16
    W, x, g = _mk(i, k, B, 7)
    W *= 0.3
    Wd = torch.from_numpy(W).to(DEV).to(dtype); xd = torch.from_numpy(x).to(DEV).to(dtype)
    gd = torch.from_numpy(g).to(DEV).to(dtype)
    out_ref, fb_ref = O.ffmlp_forward(xd.float().cpu().numpy(), Wd.float().cpu().numpy(), i, 16, 64, k, act, 6, rnd=1)
    fb = torch.empty(k, B, 64, device=DEV, dtype=dtype); out = torch.empty(B, 16, device=DEV, dtype=dtype)
    ff.ffmlp_forward(xd, Wd, B, i, 16, 64, k, act, 6, fb, out)
    np.testing.assert_allclose(out.float().cpu().numpy(), out_ref, rtol=0.03, atol=0.02)
    gi_ref, gw_ref, bb_ref = O.ffmlp_backward(gd.float().cpu().numpy(), xd.float().cpu().numpy(),
                                              Wd.float().cpu().numpy(), fb.float().cpu().numpy(), i, 16, 64, k, act,
                                              True, rnd=1)
    bb = torch.zeros(k, B, 64, device=DEV, dtype=dtype); gi = torch.zeros(B, i, device=DEV, dtype=dtype)
    gw = torch.zeros_like(Wd)
    ff.ffmlp_backward(gd, xd, Wd, fb, B, i, 16, 64, k, act, 6, True, bb, gi, gw)
    np.testing.assert_allclose(gi.float().cpu().numpy(), gi_ref, rtol=0.05, atol=0.02)
    np.testing.assert_allclose(gw.float().cpu().numpy(), gw_ref, rtol=0.05, atol=0.02 * np.abs(gw_ref).max())


def test_ffmlp_module_vs_fp32_linear_stack():
    """FFMLP module (bf16) forward/backward vs the same weights as an fp32 nn.Linear/ReLU stack: bf16 tolerance."""
    from enerf_amd.ffmlp import FFMLP
    for (i, o, k, B) in ((32, 16, 2, 1000), (32, 3, 3, 4096)):
        net = FFMLP(i, o, 64, k).to(DEV).train()
        x = (torch.rand(B, i, device=DEV) * 2 - 1).requires_grad_(True)
        y = net(x)
        assert y.shape == (B, o)
        g = torch.randn(B, o, device=DEV)
        (y.float() * g).sum().backward()
        W = net.weights.detach().float()
        xr = x.detach().clone().requires_grad_(True)
        Wr = W.clone().requires_grad_(True)
        h = torch.relu(xr @ Wr[: 64 * i].view(64, i).t())
        off = 64 * i
        for _ in range(k - 1):
            h = torch.relu(h @ Wr[off: off + 4096].view(64, 64).t()); off += 4096
        yr = (h @ Wr[off:].view(16, 64).t())[:, :o]
        (yr * g).sum().backward()
        assert (y.float() - yr).abs().max() < 0.05 * yr.abs().max()
        # a bf16-rounded pre-activation can land on the other side of a ReLU kink than its fp32 value: compare the
        # input gradient statistically (such flips change single entries by O(|w|)), the summed weight gradient tightly
        ex = (x.grad - xr.grad).abs()
        sx = xr.grad.abs().max()
        # (CPU oracle, bf16-rounded vs fp32, same shapes: mean error ~0.3-1% of max, ~1-3% of entries off by > 5%)
        assert ex.mean() < 0.02 * sx and (ex > 0.05 * sx).float().mean() < 0.05
        assert (net.weights.grad - Wr.grad).abs().max() < 0.08 * Wr.grad.abs().max()   # CPU oracle bf16-vs-fp32: ~6%
        net.eval()
        with torch.no_grad():
            yi = net(x.detach())
        assert (yi.float() - yr).abs().max() < 0.05 * yr.abs().max()


def test_network_ff_render_runs_and_matches_fp32_stack():
    """nerf/network_ff-shaped model on the HIP path: sigma / rgb close to the fp32 restatement of the same weights."""
    from enerf_amd.network_ff import NeRFNetwork
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to(DEV).eval()
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = (torch.rand(3000, 3, device=DEV) * 4 - 2)
    d = torch.nn.functional.normalize(torch.randn(3000, 3, device=DEV), dim=-1)
    with torch.no_grad():
        sigma, rgb = m(x, d)
        enc = m.encoder(x, bound=2)
        W = m.sigma_net.weights.float()
        h = torch.relu(enc @ W[:2048].view(64, 32).t())
        h = torch.relu(h @ W[2048:2048 + 4096].view(64, 64).t())
        o = h @ W[2048 + 4096:].view(16, 64).t()
        sig_ref = torch.exp(o[:, 0])
    rel = ((sigma - sig_ref).abs() / sig_ref.clamp(min=1e-3))
    assert rel.median() < 0.02 and rel.max() < 0.3
    assert torch.isfinite(rgb).all() and rgb.shape == (3000, 3) and rgb.min() >= 0 and rgb.max() <= 1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("n", [1, 4000, 70001])
def test_fused_ff_network_inference_matches_op_by_op_route(monkeypatch, dtype, n):
    """nerf/network_ff.py's forward as grid encode + ONE MFMA kernel (csrc/ffnerf.hip) against the op-by-op route
    (encoder -> FFMLP -> exp ; SH, cat -> FFMLP -> sigmoid): same 16-bit roundings, so the results agree to an ulp of
    the compute type almost everywhere (a hidden activation that lands on a rounding boundary may flip one)."""
    from enerf_amd import fused_network_ff
    from enerf_amd.ffmlp import FFMLP
    from enerf_amd.network_ff import NeRFNetwork
    monkeypatch.setattr(FFMLP, "compute_dtype", dtype)
    torch.manual_seed(0)
    net = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to("cuda").eval()
    net.encoder.embeddings.data.uniform_(-1.0, 1.0)
    g = torch.Generator(device="cuda").manual_seed(n)
    net.color_net.weights.data.copy_((torch.rand(net.color_net.weights.shape, generator=g, device="cuda") - 0.5) * 0.6)
    x = (torch.rand(n, 3, generator=g, device="cuda") * 2 - 1) * 2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, device="cuda"), dim=-1)
    calls = []
    orig = fused_network_ff.forward
    monkeypatch.setattr(fused_network_ff, "forward", lambda *a: (calls.append(1), orig(*a))[1])
    with torch.no_grad():
        s1, c1 = net(x, d)
        monkeypatch.setattr(fused_network_ff, "ENABLED", False)
        s0, c0 = net(x, d)
    assert len(calls) == 1 and s1.dtype == torch.float32 and c1.shape == (n, 3)
    s0, c0 = s0.float(), c0.float()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    rel = ((s1 - s0).abs() / s0.abs().clamp(min=1e-6))
    assert float((rel > 2.5 * ulp).float().mean()) < 0.02 and float(rel.max()) < 0.1, float(rel.max())
    err = (c1 - c0).abs()
    assert float((err > 2.5 * ulp).float().mean()) < 0.02 and float(err.max()) < 0.05, float(err.max())
    if n > 1:
        assert float(s1.std()) > 0 and float(c1.std()) > 0.01      # not a degenerate comparison
    # a network that may need gradients keeps the autograd route
    monkeypatch.setattr(fused_network_ff, "ENABLED", True)
    net.train()
    net(x[:64], d[:64])
    assert len(calls) == 1


@pytest.mark.parametrize("n", [1, 4000, 70001])
def test_fused_ff_network_training_node_matches_op_by_op_route(monkeypatch, n):
    """nerf/network_ff.py's TRAINING forward + backward as one autograd node (enerf_amd/fused_network.py, kind "ff": grid
    encode -> csrc/mlp32s.hip with bf16 operands -> grid backward) against the op-by-op route through the FFMLP entry
    points (encoder -> ffmlp_forward -> exp ; SH, cat -> ffmlp_forward -> sigmoid, and autograd back through
    ffmlp_backward).  Forward: the same 16-bit roundings at the same points, so the results agree to an ulp of bf16
    almost everywhere.  Backward: the node keeps weight gradients in fp32 where the FFMLP route rounds them (and its
    intermediate input gradients) to bf16, so the bars are bf16-sized, relative to each tensor's largest entry."""
    from enerf_amd import fused_network as fn
    from enerf_amd.network_ff import NeRFNetwork
    torch.manual_seed(0)
    net = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to("cuda").train()
    assert fn.kind_of(net) == "ff"
    net.encoder.embeddings.data.uniform_(-1.0, 1.0)
    g = torch.Generator(device="cuda").manual_seed(n)
    net.color_net.weights.data.copy_((torch.rand(net.color_net.weights.shape, generator=g, device="cuda") - 0.5) * 0.6)
    x = (torch.rand(n, 3, generator=g, device="cuda") * 2 - 1) * 2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g, device="cuda"), dim=-1)
    gs = torch.randn(n, generator=g, device="cuda") * 0.1
    gc = torch.randn(n, 3, generator=g, device="cuda")

    def run(enabled):
        monkeypatch.setattr(fn, "ENABLED", enabled)
        net.zero_grad()
        s, c = net(x, d)
        ((s.float() * gs).sum() + (c.float() * gc).sum()).backward()
        return s.detach().float(), c.detach().float(), {k: p.grad.clone() for k, p in net.named_parameters()}

    calls = []
    orig = fn.forward
    monkeypatch.setattr(fn, "forward", lambda *a: (calls.append(1), orig(*a))[1])
    s1, c1, g1 = run(True)
    assert len(calls) == 1 and s1.dtype == torch.float32 and c1.shape == (n, 3)
    s0, c0, g0 = run(False)
    assert len(calls) == 1
    ulp = 2.0 ** -8
    rel = ((s1 - s0).abs() / s0.abs().clamp(min=1e-6))
    assert float((rel > 2.5 * ulp).float().mean()) < 0.02 and float(rel.max()) < 0.1, float(rel.max())
    err = (c1 - c0).abs()
    assert float((err > 2.5 * ulp).float().mean()) < 0.02 and float(err.max()) < 0.05, float(err.max())
    for k in g0:
        a, b = g1[k].float(), g0[k].float()
        top = float(b.abs().max())
        assert top > 0, k
        # a handful of ReLU decisions differ between the two routes (bf16 boundaries), each worth one sample's terms
        assert float((a - b).abs().max()) <= (0.04 if n > 1 else 0.02) * top, (k, float((a - b).abs().max()) / top)
        assert float((a - b).abs().mean()) <= 2e-2 * float(b.abs().mean()) + 1e-12, k          # (bf16: 2^-8 per rounding)
    # the pad column of the colour net's first layer and its unused output rows never receive a gradient
    wc = g1["color_net.weights"]
    assert float(wc[:2048].view(64, 32)[:, 31].abs().max()) == 0.0
    assert float(wc[2048 + 8192 + 3 * 64:].abs().max()) == 0.0


@pytest.mark.parametrize("fp16", [False, True])
def test_network_ff_training_step_takes_the_closed_form_path(fp16):
    """(fp16=True: the shipped configs' regime on network_ff -- fp16 operands for both FFMLP nets, the three-hidden-layer
    backward included, under the device-side GradScaler.)
    TrainHarness drives network_ff through the same closed-form step as the nn.Linear nets (march -> grid -> fused MLPs
    -> composite + MSE gradient -> fused MLP backward -> table records -> one optimizer launch): the loss falls, the
    table gradient is never materialised, and the FFMLP master weights are what Adam updates."""
    from enerf_amd import fused_render
    from enerf_amd.network_ff import NeRFNetwork
    from enerf_amd.trainer import TrainHarness
    from test_gpu_training import _batches
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to(DEV)
    h = TrainHarness(model, lr=1e-2, occupancy="synthetic", fp16=fp16)
    assert h.amp_f16 == fp16
    data = _batches(8, 2048, 2)
    assert h._manual_ok(data[0][0], data[0][1], data[0][2], {})
    w0 = model.sigma_net.weights.detach().clone()
    calls = []
    orig, orig_native = fused_render.train_step_mse, fused_render.train_step_native
    fused_render.train_step_mse = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    fused_render.train_step_native = lambda *a, **k: (calls.append(1), orig_native(*a, **k))[1]      # (the same, from C)
    try:
        losses = [float(h.step_rgb(*data[i % 8])) for i in range(120)]
    finally:
        fused_render.train_step_mse, fused_render.train_step_native = orig, orig_native
    assert len(calls) == 120
    assert np.isfinite(losses).all() and np.mean(losses[-8:]) < 0.5 * np.mean(losses[:8]), (losses[:4], losses[-4:])
    assert float((model.sigma_net.weights.detach() - w0).abs().max()) > 1e-3
    assert model.mean_count > 0
    if fp16:
        assert float(h.scaler.get_scale()) >= 256.0 and h.amp_skipped_steps() <= 10


@pytest.mark.parametrize("route", ["network_ff fused inference", "network_ff op by op", "network.py, one launch per net"])
def test_mfma_kernels_are_bit_stable_from_run_to_run(monkeypatch, route):
    """Soak for every MFMA forward kernel that shares its SIMDs with other wavefronts (several workgroups per CU): the same
    400 000 rows 40 times, every output bit-identical to the first run's.  Round 5 found two ways for such kernels to be
    right on most launches only -- an MFMA result allocated over its own operands (csrc/mfma_guard.h, which these kernels
    carried since round 3) and whatever made csrc/nerf_mlp.hip's forward unstable at more than one workgroup per CU; a test
    that compares with a reference at 1e-4 on one launch sees neither."""
    from enerf_amd import _lib, fused_network, fused_network_ff
    n = 400000
    torch.manual_seed(3)
    if route.startswith("network_ff"):
        from enerf_amd.network_ff import NeRFNetwork
        monkeypatch.setattr(fused_network_ff, "ENABLED", "fused" in route)
    else:
        from enerf_amd.network import NeRFNetwork
        prev = _lib.lib().enerf_debug_nerf_mlp_fused(0)
    try:
        net = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True).to(DEV).eval()
        net.encoder.embeddings.data.uniform_(-1.0, 1.0)
        x = (torch.rand(n, 3, device=DEV) * 2 - 1) * 2
        d = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV), dim=-1)
        with torch.no_grad():
            s0, c0 = net(x, d)
            assert float(c0.float().std()) > 1e-3
            for it in range(40):
                s1, c1 = net(x, d)
                assert torch.equal(s1, s0) and torch.equal(c1, c0), (route, it, int((s1 != s0).sum()), int((c1 != c0).any(dim=1).sum()))
    finally:
        if not route.startswith("network_ff"):
            _lib.lib().enerf_debug_nerf_mlp_fused(prev)
