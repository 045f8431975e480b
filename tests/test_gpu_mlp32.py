"""Fused fp32 MLP vs the plain nn.Linear / ReLU stack it replaces, in both of its arithmetic modes
(enerf_mlp32_precision): "fp32" = v_mfma_f32_32x32x2_f32, every dot product an fp32 fmaf chain -- the bars below are
fp32 round-off (2e-6 of the output scale on the forward); "split-bf16" (the default of the product) = operands as bf16
hi + lo, three bf16 MFMA products per fp32 product -- the same bars times 8, i.e. 1.6e-5 on the forward, all inside the
path's 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, params=["split-bf16", "fp32"])
def precision(request):
    """Runs every test of this file in both arithmetic modes; the value is the factor on the fp32 round-off bars."""
    from enerf_amd import _lib
    mode = 1 if request.param == "split-bf16" else 0
    prev = _lib.lib().enerf_mlp32_precision(mode)
    assert _lib.lib().enerf_mlp32_precision(-1) == mode
    yield 8.0 if mode == 1 else 1.0
    _lib.lib().enerf_mlp32_precision(prev)


def _ref(x, ws):
    h = x
    for k, w in enumerate(ws):
        h = h @ w.t()
        if k != len(ws) - 1:
            h = torch.relu(h)
    return h


@pytest.mark.parametrize("dims,B", [((32, 64, 16), 4096), ((31, 64, 64, 3), 5000), ((32, 64, 64, 64, 32), 96),
                                    ((20, 64, 1), 33)])
def test_fused_mlp_forward_backward(dims, B, precision):
    k = precision
    from enerf_amd.fused_mlp import fused_mlp, supported
    torch.manual_seed(len(dims) * 7 + B)
    ws = [(torch.rand(dims[k + 1], dims[k], device=DEV) * 2 - 1) * (3.0 / dims[k]) ** 0.5 for k in range(len(dims) - 1)]
    x = torch.rand(B, dims[0], device=DEV) * 2 - 1
    g = torch.randn(B, dims[-1], device=DEV)
    xa = x.clone().requires_grad_(True)
    wa = [w.clone().requires_grad_(True) for w in ws]
    assert supported(xa, wa)
    y = fused_mlp(xa, wa)
    (y * g).sum().backward()
    xb = x.clone().requires_grad_(True)
    wb = [w.clone().requires_grad_(True) for w in ws]
    # fp64 reference of the same stack, keeping every layer's input and output gradient (the terms of dW = dY^T X)
    acts, pres = [xb.double()], []
    for li, w in enumerate(wb):
        pre = acts[-1] @ w.double().t()
        pre.retain_grad()
        pres.append(pre)
        acts.append(torch.relu(pre) if li != len(wb) - 1 else pre)
    yr = acts[-1]
    (yr * g.double()).sum().backward()
    sc = float(yr.detach().abs().max())
    assert float((y.double() - yr).abs().max()) < k * 2e-6 * max(sc, 1.0)
    # a hidden unit whose fp64 pre-activation lies within the forward error of zero may sit on the other side of the
    # ReLU on the device: that sample's input gradient and one term of each weight gradient legitimately differ
    with torch.no_grad():
        h, kink = xb.double(), torch.zeros(B, dtype=torch.bool, device=DEV)
        for w in wb[:-1]:
            pre = h @ w.double().t()
            kink |= ((pre.abs() < k * 2e-6 * max(float(pre.abs().max()), 1.0)) & (pre != 0)).any(dim=1)
            h = torch.relu(pre)
    n_kink = int(kink.sum())
    assert n_kink <= k * (2 + B // 500)
    ok = ~kink
    assert float((xa.grad.double() - xb.grad)[ok].abs().max()) < k * 1e-5 * float(xb.grad.abs().max())
    # Weight gradients, bounded per entry by a forward error analysis of the reference's own computation: dW_l = dY_l^T X_l
    # where X_l comes through l layers and dY_l back through the L - l layers behind it, each a matrix product whose
    # rounding error is at most u x (the product of the operands' MAGNITUDES).  So the magnitudes are propagated: X^_0 = |x|,
    # X^_l = mask_l (X^_{l-1} |W_{l-1}|^T), dY^_L = |g|, dY^_{l-1} = mask (dY^_l |W_l|), A^_l = dY^_l^T X^_l, and every entry
    # must satisfy  |dW - dW_fp64| <= 1e-4 |dW_fp64| + (L + 1) u A^  with u = 2^-16 (one split-bf16 product: hi*hi + hi*lo +
    # lo*hi) or 8 eps32 (fp32 fmaf chains of up to 64 terms + the per-workgroup partial sums).  T = the largest single
    # term is allowed once per knife-edge sample (a ReLU that may legitimately sit on the other side); the second pass below
    # takes those samples out of the batch and allows nothing.
    u = 2.0 ** -16 if k > 1 else 8 * 2.0 ** -23
    nl = len(ws)

    def magnitudes(xin, gin, wlist, pre_list):
        xa = [xin.abs()]
        for li in range(nl - 1):
            xa.append((xa[-1] @ wlist[li].abs().t()) * (pre_list[li] > 0))
        da = [None] * nl
        da[nl - 1] = gin.abs()
        for li in range(nl - 1, 0, -1):
            da[li - 1] = (da[li] @ wlist[li].abs()) * (pre_list[li - 1] > 0)
        return [da[li].t() @ xa[li] for li in range(nl)]

    with torch.no_grad():
        near = torch.zeros(B, dtype=torch.bool, device=DEV)
        for pre in pres[:-1]:
            near |= ((pre.abs() < 4 * k * 2e-6 * max(float(pre.abs().max()), 1.0)) & (pre != 0)).any(dim=1)
        A_hat = magnitudes(xb.double(), g.double(), [w.double() for w in wb], [p_.detach() for p_ in pres])
    n_near = int(near.sum())
    worst = 0.0
    for li, (a, b) in enumerate(zip(wa, wb)):
        X, dY = acts[li].detach(), pres[li].grad
        G = dY.t() @ X
        # (the hooks saw the real terms: autograd's own gradient, rounded to the fp32 leaf, is this sum)
        assert float((G - b.grad.double()).abs().max()) <= 4 * 2.0 ** -23 * max(float(G.abs().max()), 1.0)
        T = torch.zeros_like(G)
        for c0 in range(0, B, 2048):
            T = torch.maximum(T, (dY[c0:c0 + 2048, :, None].abs() * X[c0:c0 + 2048, None, :].abs()).amax(0))
        err = (a.grad.double() - G).abs()
        bar = 1e-4 * G.abs() + (nl + 1) * u * A_hat[li] + n_near * T + 1e-12
        worst = max(worst, float((err / bar).max()))
        assert bool((err <= bar).all()), (li, float((err / bar).max()), float(err.max() / G.abs().max()))
    print(f"dW per-entry error / bar (dims {dims}, B {B}, {'split-bf16' if k > 1 else 'fp32'}): worst {worst:.3f}, kinks {n_kink} (near: {n_near})")
    keep = ~near
    if n_near and int(keep.sum()) >= 32:
        xs, gs = x[keep].contiguous(), g[keep].contiguous()
        wa2 = [w.clone().requires_grad_(True) for w in ws]
        (fused_mlp(xs, wa2) * gs).sum().backward()
        acts2, pres2 = [xs.double()], []
        wd = [w.double().requires_grad_(True) for w in ws]
        for li, w in enumerate(wd):
            pre = acts2[-1] @ w.t()
            pre.retain_grad()
            pres2.append(pre)
            acts2.append(torch.relu(pre) if li != len(wd) - 1 else pre)
        (acts2[-1] * gs.double()).sum().backward()
        with torch.no_grad():
            A2 = magnitudes(xs.double(), gs.double(), [w.detach() for w in wd], [p_.detach() for p_ in pres2])
        worst2 = 0.0
        for li, (a, b) in enumerate(zip(wa2, wd)):
            err = (a.grad.double() - b.grad).abs()
            bar = 1e-4 * b.grad.abs() + (nl + 1) * u * A2[li] + 1e-12
            worst2 = max(worst2, float((err / bar).max()))
            assert bool((err <= bar).all()), ("no-kink batch", li, float((err / bar).max()))
        print(f"   without the {n_near} knife-edge samples: worst error / bar {worst2:.3f}")
    # inference path (no grad) gives the same values
    with torch.no_grad():
        y2 = fused_mlp(x, ws)
    assert torch.equal(y2, y.detach())


@pytest.mark.parametrize("B", [1, 31, 4097, 70000])
def test_level_major_encoder_into_fused_mlp(B, precision):
    """Grid encoder (out_layout 2, [16,Bp,2]) -> fused MLP (x_layout 1) -> and back: same values / gradients as the
    row-major route, bit for bit on the encoding and to fp32 round-off through the MLP."""
    from enerf_amd.fused_mlp import fused_mlp, pad32
    from enerf_amd.gridencoder import GridEncoder
    k = precision
    torch.manual_seed(B)
    enc = GridEncoder(desired_resolution=4096).to(DEV)
    enc.embeddings.data.uniform_(-1, 1)
    ws = [(torch.rand(o, i, device=DEV) * 2 - 1) * (3.0 / i) ** 0.5 for o, i in ((64, 32), (16, 64))]
    x = (torch.rand(B, 3, device=DEV) * 2 - 1)
    x[0] = 1.5                                   # one out-of-range point: zero features, zero gradient
    g = torch.randn(B, 16, device=DEV)

    feats_rows = enc(x, bound=1)
    lm, n = enc.forward_level_major(x, bound=1)
    assert n == B and tuple(lm.shape) == (16, pad32(B), 2)
    assert torch.equal(lm[:, :B].permute(1, 0, 2).reshape(B, 32), feats_rows)
    assert float(lm[:, B:].abs().sum()) == 0.0   # pad rows are written as zeros

    def run(level_major):
        enc.zero_grad()
        wa = [w.clone().requires_grad_(True) for w in ws]
        xa = x.clone().requires_grad_(True)
        if level_major:
            f, nn_ = enc.forward_level_major(xa, bound=1)
            y = fused_mlp(f, wa, x_layout=1, batch=nn_)
        else:
            y = fused_mlp(enc(xa, bound=1), wa)
        (y * g).sum().backward()
        return y.detach(), enc.embeddings.grad.clone(), xa.grad.clone(), [w.grad.clone() for w in wa]

    y1, ge1, gx1, gw1 = run(True)
    y0, ge0, gx0, gw0 = run(False)
    sc = max(float(y0.abs().max()), 1.0)
    assert float((y1 - y0).abs().max()) < k * 2e-6 * sc
    # fp64 statement of the MLP on the exact encoder output
    pre = feats_rows.double() @ ws[0].double().t()
    h = torch.relu(pre) @ ws[1].double().t()
    assert float((y1.double() - h).abs().max()) < k * 2e-6 * sc
    # The two layouts sum the first layer in different orders, so a pre-activation within round-off of zero may
    # take the other side of the ReLU kink; such a sample legitimately changes its own gradients (8 rows per level
    # of the embedding gradient, its input gradient, one term of each weight gradient).
    kink = ((pre.abs() < k * 2e-6) & (pre != 0)).any(dim=1)      # exact zeros (out-of-range point) are not kinks
    n_kink = int(kink.sum())
    assert n_kink <= (3 + B // 1000) * k
    bad_rows = ((ge1 - ge0).abs().max(dim=1).values > k * 2e-5 * float(ge0.abs().max()) + 1e-12).sum()
    assert int(bad_rows) <= 128 * n_kink
    ok = ~kink
    assert float((gx1[ok] - gx0[ok]).abs().max()) <= k * 2e-5 * float(gx0.abs().max()) + 1e-12
    for a, b in zip(gw1, gw0):
        assert float((a - b).abs().max()) <= (k * 2e-5 + 1e-3 * n_kink) * float(b.abs().max()) + 1e-12


@pytest.mark.parametrize("N", [1, 4096, 70001])
def test_fused_network_node_matches_unfused_route(monkeypatch, N, precision):
    """enerf_amd.fused_network (one autograd node for grid -> sigma MLP -> trunc_exp / SH -> colour MLP -> sigmoid)
    against the op-by-op route of network.py, which itself is pinned against the nn.Linear loop below."""
    from enerf_amd import fused_network as fn
    from enerf_amd.network import NeRFNetwork
    torch.manual_seed(1)
    m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(N, 3, device=DEV) * 6 - 3
    x[0] = torch.tensor([3.0, -3.0, 0.5])                      # on the boundary
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    gs, gc = torch.randn(N, device=DEV), torch.randn(N, 3, device=DEV)

    def run(enabled):
        monkeypatch.setattr(fn, "ENABLED", enabled)
        m.zero_grad()
        s, c = m(x, d)
        ((s * gs).sum() + (c * gc).sum()).backward()
        return s.detach(), c.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}

    calls = []
    orig = fn.forward
    monkeypatch.setattr(fn, "forward", lambda *a: (calls.append(1), orig(*a))[1])
    s1, c1, g1 = run(True)
    assert len(calls) == 1
    s0, c0, g0 = run(False)
    assert len(calls) == 1
    assert float(((s1 - s0).abs() / s0.abs().clamp(min=1e-6)).max()) < precision * 2e-5
    assert float((c1 - c0).abs().max()) < precision * 2e-6
    for n in g0:
        tol = (5e-4 if N > 10000 else 5e-5) * (2 if precision > 1 else 1)          # a ReLU kink flipped by the summation order moves one sample's terms
        assert float((g1[n] - g0[n]).abs().max()) <= tol * float(g0[n].abs().max()) + 1e-9, n
    # accumulation straight into an existing .grad gives the same embedding gradient
    m.zero_grad()
    m.encoder.embeddings.grad = torch.zeros_like(m.encoder.embeddings)
    monkeypatch.setattr(fn, "ENABLED", True)
    s, c = m(x, d)
    ((s * gs).sum() + (c * gc).sum()).backward()
    ref = g1["encoder.embeddings"]
    assert float((m.encoder.embeddings.grad - ref).abs().max()) <= precision * 2e-5 * float(ref.abs().max()) + 1e-9
    with torch.no_grad():
        s2, c2 = m(x, d)
    assert torch.equal(s2, s1) and torch.equal(c2, c1)


def test_network_uses_fused_path_and_matches_linear_loop(monkeypatch, precision):
    from enerf_amd import fused_mlp as fm
    from enerf_amd import fused_network as fn
    from enerf_amd.network import NeRFNetwork
    monkeypatch.setattr(fn, "ENABLED", False)
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(4096, 3, device=DEV) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(4096, 3, device=DEV), dim=-1)
    calls = []
    orig = fm.fused_mlp
    monkeypatch.setattr(fm, "fused_mlp", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    s1, c1 = m(x, d)
    (s1.sum() + c1.sum()).backward()
    assert len(calls) == 2
    g1 = {n: p.grad.clone() for n, p in m.named_parameters()}
    m.zero_grad()
    monkeypatch.setattr(fm, "supported", lambda *a: False)
    monkeypatch.setattr(fm, "supported_level_major", lambda *a: False)
    s2, c2 = m(x, d)
    (s2.sum() + c2.sum()).backward()
    assert float(((s1 - s2).abs() / s2.abs().clamp(min=1e-6)).max()) < 1e-4
    assert float((c1 - c2).abs().max()) < 1e-5
    for n, p in m.named_parameters():
        # split-bf16: ~10x the forward error of the fp32 chains, so a few more hidden units of the 4096 x 192 sit on the
        # other side of their ReLU than in the nn.Linear loop; each moves one row of a weight gradient by one sample's term
        err, top = (p.grad - g1[n]).abs(), float(p.grad.abs().max())
        if precision == 1:
            assert float(err.max()) <= 2e-4 * top + 1e-7, n
        else:
            assert float(err.max()) <= 4e-3 * top + 1e-7 and float((err > 2e-4 * top).float().mean()) <= 0.05, n


def test_fused_adam_matches_torch_adam():
    from enerf_amd.optim import FusedAdam
    torch.manual_seed(3)
    shapes = [(100003, 2), (64, 32), (7,)]
    pa = [torch.randn(s, device=DEV).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    oa = FusedAdam([{"params": pa[:1], "lr": 1e-2}, {"params": pa[1:], "lr": 3e-3}], betas=(0.9, 0.99), eps=1e-15)
    ob = torch.optim.Adam([{"params": pb[:1], "lr": 1e-2}, {"params": pb[1:], "lr": 3e-3}], betas=(0.9, 0.99),
                          eps=1e-15)
    sa = torch.optim.lr_scheduler.LambdaLR(oa, lambda it: 0.1 ** (it / 10))
    sb = torch.optim.lr_scheduler.LambdaLR(ob, lambda it: 0.1 ** (it / 10))
    for it in range(6):
        for a, b in zip(pa, pb):
            g = torch.randn_like(a) * (10.0 ** (-it))
            if it == 2:
                g[::2] = 0            # untouched rows keep decaying through m / v
            a.grad = g.clone()
            b.grad = g.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
        for a, b in zip(pa, pb):
            assert float((a - b).abs().max()) < 2e-6 * float(b.abs().max())


def test_density_sigma_only_path_matches_density(precision):
    """fused_network.density_sigma (update_extra_state's evaluator: sigma MLP writes exp(column 0) only) vs
    NeRFNetwork.density on the same points."""
    from enerf_amd import fused_network as fn
    from enerf_amd.network import NeRFNetwork
    torch.manual_seed(2)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(50001, 3, device=DEV) * 4 - 2
    with torch.no_grad():
        ref = m.density(x)["sigma"]
        got = fn.density_sigma(m, x)
    assert got.shape == ref.shape
    # (both sides run the same kernels' arithmetic mode; the sigma-only variant sums its output row on the VALU in fp32)
    assert float(((got - ref).abs() / ref.abs().clamp(min=1e-6)).max()) < (2e-5 if precision == 1 else 1e-4)


def test_grid_table_adam_from_records_equals_backward_then_adam():
    """FusedAdam.step_grid_table (deferred grid backward: the tiles' record lists are summed in LDS by the optimizer's
    own pass over the table, csrc/gridencoder.hip k_grid_tile_adam) against grid_encode_backward into a dense gradient
    followed by the fused Adam kernel: same moments, same parameters, the dense gradient buffer comes back clean; two
    backward calls joining one flush (an event step's two renders) included; below the binning threshold the same
    entry point is a plain Adam over the dense gradient."""
    from enerf_amd.backends import _gridencoder as ge
    from enerf_amd.gridencoder import GridEncoder
    from enerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    # (9000, 30000) / (30000, 9000): one call of the session below the binning threshold (16384) -- the session is
    # decided once, from the step's total, and every call that joins it is binned
    for sizes in ((70000,), (40000, 52000), (3000,), (9000, 30000), (30000, 9000)):
        encs = [GridEncoder(desired_resolution=2048 * 2).to("cuda") for _ in range(2)]
        encs[1].embeddings.data.copy_(encs[0].embeddings.data)
        opts = [FusedAdam([{"params": [e.embeddings], "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15) for e in encs]
        S = float(np.log2(encs[0].per_level_scale))
        dummy = torch.empty(1, device="cuda")
        for step in range(3):
            xs = [torch.rand(n, 3, device="cuda") for n in sizes]
            gs = [torch.randn(n, 32, device="cuda") * (0.1 + step) for n in sizes]
            for k, (enc, opt) in enumerate(zip(encs, opts)):
                p = enc.embeddings
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                for x, g in zip(xs, gs):
                    ge.grid_encode_backward(g, x, p.data, enc.offsets, p.grad, x.shape[0], 3, 2, 16, S, 16, False, dummy,
                                            dummy, 0, layout=1, defer=(k == 1), reserve=sum(sizes) if k == 1 else 0)
                if k == 1:
                    opt.step_grid_table(p, enc.offsets, 2)
                    assert not bool(p.grad.any())                      # what was written densely has been consumed
                else:
                    opt.step_now(zero_grads=True)
            a, b = (o.state[e.embeddings] for o, e in zip(opts, encs))
            for key in ("exp_avg", "exp_avg_sq"):
                scale = float(a[key].abs().max())
                assert float((a[key] - b[key]).abs().max()) <= 2e-6 * scale, (sizes, step, key)
            pa, pb = encs[0].embeddings.data, encs[1].embeddings.data
            # Adam's first steps move every touched row by ~lr whatever the gradient's size: rows whose gradient is
            # round-off-sized noise may flip; everything else must agree to fp32
            assert float(((pa - pb).abs() > 1e-6).float().mean()) < 2e-4, (sizes, step)
            assert a["step"] == b["step"] == step + 1


def test_pending_record_lists_refuse_foreign_calls_and_can_be_discarded():
    """While a deferred flush is pending only calls that join it are accepted (a plain backward would silently lose
    its binned levels to the session); enerf_grid_records_discard empties the lists, after which a plain backward gives
    the same gradient as if nothing had happened."""
    from enerf_amd import _lib
    from enerf_amd.backends import _gridencoder as ge
    from enerf_amd.gridencoder import GridEncoder
    enc = GridEncoder(desired_resolution=2048 * 2).to("cuda")
    S = float(np.log2(enc.per_level_scale))
    dummy = torch.empty(1, device="cuda")
    x = torch.rand(30000, 3, device="cuda")
    g = torch.randn(30000, 32, device="cuda")

    def bwd(out, **kw):
        ge.grid_encode_backward(g, x, enc.embeddings.data, enc.offsets, out, 30000, 3, 2, 16, S, 16, False, dummy, dummy, 0,
                                layout=1, **kw)
    ref = torch.zeros_like(enc.embeddings.data)
    bwd(ref)
    scratch = torch.zeros_like(ref)
    bwd(scratch, defer=True)
    with pytest.raises(RuntimeError, match="deferred flush is pending"):
        bwd(torch.zeros_like(ref))
    _lib.check(_lib.lib().enerf_grid_records_discard(_lib.stream_handle()), "discard")
    again = torch.zeros_like(ref)
    bwd(again)
    assert float((again - ref).abs().max()) <= 1e-6 * float(ref.abs().max())


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("nh,out,xl,B", [(1, 16, 1, 4097), (1, 16, 0, 5000), (2, 3, 0, 5000), (2, 16, 1, 96),
                                         (3, 16, 0, 4100), (2, 3, 0, 31)])
def test_backward_recomputing_the_activations_is_bit_identical_to_loading_them(mode, nh, out, xl, B):
    """enerf_mlp32_recompute(1) (default): the split / bf16 backward computes the hidden activations again from X and
    the training forward does not store them.  Same instruction sequence as the forward -> same values, same ReLU
    masks: dX and dW must be BIT-identical to the stored-activation route, for every shape the kernel serves; and the
    forward buffer really is left alone."""
    from enerf_amd import _lib as L
    from enerf_amd.fused_mlp import pad32
    if nh == 3 and mode != 2:
        pytest.skip("three hidden layers: bf16 operands only")
    lib, s = L.lib(), L.stream_handle()
    prev_mode = lib.enerf_mlp32_precision(mode)
    prev_rc = lib.enerf_mlp32_recompute(-1)
    try:
        torch.manual_seed(11)
        Bp = pad32(B)
        dims = [32] + [64] * nh + [out]
        ws = [(torch.rand(dims[k + 1], dims[k], device=DEV) * 2 - 1) * (3.0 / dims[k]) ** 0.5 for k in range(len(dims) - 1)]
        W = torch.cat([w.reshape(-1) for w in ws]).contiguous()
        xr = torch.rand(B, 32, device=DEV) * 2 - 1
        if xl:
            X = torch.zeros(16, Bp, 2, device=DEV)
            X[:, :B] = xr.view(B, 16, 2).permute(1, 0, 2)
        else:
            X = xr.contiguous()
        dY = torch.randn(B, out, device=DEV)
        res = {}
        for rc in (0, 1):
            lib.enerf_mlp32_recompute(rc)
            fb = torch.full((nh * Bp * 64,), 7.0, device=DEV)
            bb = torch.empty(nh, Bp, 64, device=DEV)
            Y = torch.empty(B, out, device=DEV)
            dX, dW = torch.zeros_like(X), torch.zeros_like(W)
            L.check(lib.enerf_mlp32_forward(X.data_ptr(), W.data_ptr(), B, 32, out, nh, 0, 6, fb.data_ptr(), Y.data_ptr(),
                                            xl, 0, None, s), "fwd")
            L.check(lib.enerf_mlp32_backward(dY.data_ptr(), X.data_ptr(), W.data_ptr(), fb.data_ptr(), B, 32, out, nh, 0,
                                             bb.data_ptr(), dX.data_ptr(), dW.data_ptr(), xl, 0, None, 0, None, None, 0, s),
                    "bwd")
            torch.cuda.synchronize()
            res[rc] = (Y, dX, dW, fb)
        # (three hidden layers recompute under either setting since round 5: the activation-loading instance of that shape
        #  was retired, csrc/mfma_guard.h -- its forward buffer is never written)
        assert bool((res[1][3] == 7.0).all()) and (nh == 3) == bool((res[0][3][:B * 64] == 7.0).all())
        for a, b, what in zip(res[0][:3], res[1][:3], ("Y", "dX", "dW")):
            assert torch.equal(a, b), what
        assert float(res[1][2].abs().max()) > 0 and float(res[1][1].abs().max()) > 0
    finally:
        lib.enerf_mlp32_recompute(prev_rc)
        lib.enerf_mlp32_precision(prev_mode)


@pytest.mark.parametrize("N,out_c", [(1, 3), (33, 1), (4097, 3), (70001, 3), (2048, 7)])
def test_both_nets_in_one_launch_match_one_launch_per_net(N, out_c, precision):
    """csrc/nerf_mlp.hip (sigma + colour net of nerf/network.py:104-132 as one launch per direction, split-bf16) against
    the one-launch-per-net kernels it replaces and against an fp64 nn.Linear loop: the same arithmetic up to the summation
    order inside the colour net's first layer, so the bars are the split-bf16 ones of this file."""
    from enerf_amd import _lib, fused_network as fn
    from enerf_amd.network import NeRFNetwork
    lib = _lib.lib()
    if precision == 1.0:
        assert lib.enerf_nerf_mlp_available() == 0          # the exact-fp32 arithmetic keeps the one-net kernels
        return
    assert lib.enerf_nerf_mlp_available() == 1
    torch.manual_seed(5)
    m = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=out_c).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(N, 3, device=DEV) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    gs, gc = torch.randn(N, device=DEV), torch.randn(N, out_c, device=DEV)
    params = fn.network_params(m)
    cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)

    def run(fused, scale=1.0, valid=None):
        prev = lib.enerf_debug_nerf_mlp_fused(1 if fused else 0)
        try:
            sigma, rgb, sv = fn.nerf_forward(x, d, cfg, True, params[0], offs, *params[1:], valid_rows=valid)
            assert bool(sv["fused"]) == fused
            g = fn.nerf_backward(sv, gs, gc, sigma_scale=scale)
        finally:
            lib.enerf_debug_nerf_mlp_fused(prev)
        return sigma, rgb, g

    # gradients: a hidden unit whose pre-activation sits within a round-off of zero falls on either side of its ReLU under
    # the other summation order of the colour net's first layer, which moves that one sample's terms (|g| |a| of ONE of the
    # N samples, against sums of N of them): a handful of entries may move by that much, everything else is round-off
    def grads_agree(ga, gb):
        for a, b in zip(ga, gb):
            assert a.shape == b.shape
            err, top = (a - b).abs(), float(b.abs().max())
            assert float(err.max()) <= 4e-3 * top + 1e-9, (float(err.max()), top)
            # (one flipped unit moves a whole row / column of a 64-wide matrix: a few of them are 10 % of its entries)
            assert float((err > 1e-4 * top).float().mean()) <= 0.2, float((err > 1e-4 * top).float().mean())
            assert float((err > 1e-3 * top).float().mean()) <= 0.01, float((err > 1e-3 * top).float().mean())
    s1, c1, g1 = run(True)
    s0, c0, g0 = run(False)
    assert float(((s1 - s0).abs() / s0.abs().clamp(min=1e-6)).max()) < 2e-5
    assert float((c1 - c0).abs().max()) < 2e-6
    grads_agree(g1, g0)
    # the fp64 loop
    with torch.no_grad():
        ws = [p.double() for p in params[1:]]
        feat = m.encoder(x, bound=m.bound).double()
        h = torch.relu(feat @ ws[0].t()) @ ws[1].t()
        sh = m.encoder_dir(d).double()
        hc = torch.relu(torch.relu(torch.cat([sh, h[:, 1:]], dim=1) @ ws[2].t()) @ ws[3].t()) @ ws[4].t()
        assert float(((s1.double() - torch.exp(h[:, 0])).abs() / torch.exp(h[:, 0]).clamp(min=1e-6)).max()) < 1e-4
        assert float((c1.double() - torch.sigmoid(hc)).abs().max()) < 2e-5
    # density_scale on the fly, and the sample budget's padding rows skipped by both directions
    if N >= 2048:
        _, _, g2 = run(True, scale=0.25)
        _, _, g3 = run(False, scale=0.25)
        grads_agree(g2, g3)
        real = N - 700
        cnt = torch.tensor([real, 0], dtype=torch.int32, device=DEV)
        gs[real:] = 0
        gc[real:] = 0
        s4, c4, g4 = run(True, valid=cnt)
        s5, c5, g5 = run(False, valid=cnt)
        assert float(((s4 - s5).abs() / s5.abs().clamp(min=1e-6))[:real].max()) < 2e-5
        assert float((c4 - c5).abs()[:real].max()) < 2e-6
        grads_agree(g4, g5)


def test_both_nets_in_one_launch_are_bit_stable_from_run_to_run(precision):
    """csrc/nerf_mlp.hip, soak: the same batch through forward and backward 60 times -- every output bit-identical to the
    first run's.  (Round 5: with two or three workgroups of the forward kernel on a CU a few per cent of the later workgroups'
    tiles came out with wrong colours for samples 16..31, different rows every launch; the kernel now takes a CU for itself.
    This test is what would notice a relapse: 4160 tiles, i.e. more workgroups than CUs.)"""
    from enerf_amd import _lib, fused_network as fn
    from enerf_amd.network import NeRFNetwork
    lib = _lib.lib()
    if not lib.enerf_nerf_mlp_available():
        pytest.skip("split-bf16 only")
    N = 133000
    torch.manual_seed(2)
    m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    x = torch.rand(N, 3, device=DEV) * 6 - 3
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    gs, gc = torch.randn(N, device=DEV), torch.randn(N, 3, device=DEV)
    params = fn.network_params(m)
    cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)

    def step():
        s, c, sv = fn.nerf_forward(x, d, cfg, True, params[0], offs, *params[1:])
        assert sv["fused"]
        g_emb, dw = fn.nerf_backward(sv, gs, gc, raw=True)
        return s, c, dw

    first = step()
    for it in range(60):
        again = step()
        for a, b, what in zip(again, first, ("sigma", "rgb", "dW")):
            assert torch.equal(a, b), (it, what, int((a != b).sum()))


def test_backward_is_bit_stable_beside_a_marching_stream():
    """k_nerf_bwd is the one 16-bit MFMA kernel compiled without the operand barrier (csrc/nerf_mlp_bwd.hip: one wavefront of it
    per SIMD, nothing of ITS kind shares the matrix pipe).  In the training step it does not run alone, though: the next
    batch's march sits on a second stream and its wavefronts -- plain VALU code, 66 registers -- may be co-resident wherever
    registers and LDS allow.  The same batch through forward and backward 40 times while a second stream marches rays
    back to back: every output bit-identical to the run made on an idle device."""
    from enerf_amd import _lib, fused_network as fn, scene
    from enerf_amd.backends import _raymarching as rb
    from enerf_amd.network import NeRFNetwork
    lib = _lib.lib()
    if not lib.enerf_nerf_mlp_available():
        pytest.skip("split-bf16 only")
    N = 133000
    torch.manual_seed(3)
    m = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(DEV)
    m.encoder.embeddings.data.uniform_(-1, 1)
    scene.install_occupancy(m)
    x = torch.rand(N, 3, device=DEV) * 6 - 3
    d = torch.nn.functional.normalize(torch.randn(N, 3, device=DEV), dim=-1)
    gs, gc = torch.randn(N, device=DEV), torch.randn(N, 3, device=DEV)
    params = fn.network_params(m)
    cfg, offs = fn.network_cfg(m), fn.encoder_offsets(m)

    def step():
        s, c, sv = fn.nerf_forward(x, d, cfg, True, params[0], offs, *params[1:])
        assert sv["fused"]
        g_emb, dw = fn.nerf_backward(sv, gs, gc, raw=True)
        return s, c, dw

    first = step()
    torch.cuda.synchronize()
    # the neighbour: 4096-ray training marches (count + scan + write), queued back to back on a stream of their own
    (ro, rd), _ = scene.training_batch(0, 4096, DEV)
    ro, rd = ro.view(-1, 3).contiguous(), rd.view(-1, 3).contiguous()
    n = ro.shape[0]
    aabb = m.aabb_train if hasattr(m, "aabb_train") else torch.tensor([-3.0] * 3 + [3.0] * 3, device=DEV)
    nears, fars = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    rb.near_far_from_aabb(ro, rd, aabb, n, 0.2, nears, fars)
    M = 160000
    xyzs, dirs, deltas = (torch.empty(M, 3, device=DEV), torch.empty(M, 3, device=DEV), torch.empty(M, 2, device=DEV))
    rays = torch.empty(n, 3, dtype=torch.int32, device=DEV)
    counter = torch.zeros(2, dtype=torch.int32, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def neighbour(k):
        with torch.cuda.stream(side):
            for _ in range(k):
                counter.zero_()
                rb.march_rays_train(ro, rd, m.density_bitfield, m.bound, 0.0, 1024, n, m.cascade, m.grid_size, M, nears,
                                    fars, xyzs, dirs, deltas, rays, counter, True)
    for it in range(40):
        neighbour(3)                                   # (~0.15 ms of marching queued beside each ~0.07 ms forward + backward)
        again = step()
        for a, b, what in zip(again, first, ("sigma", "rgb", "dW")):
            assert torch.equal(a, b), (it, what, int((a != b).sum()))
    torch.cuda.synchronize()
    assert int(counter[0]) > 0                         # (the neighbour really marched)
