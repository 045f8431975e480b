"""The reference's OWN grid encoder on the same GPU (oracle/_ref/_ref_gridencoder: `gridencoder/src/gridencoder.cu` built
for gfx950 by oracle/build_ref.py -- PyTorch's translator, two call names respelled in its at::Half-only code, hipcc;
the recipe's docstring says exactly what that touches) against the C oracle and against the product's kernels, same
inputs.  This is what pins the oracle's `grid_encode_forward / backward` -- the roofline kernel of this repository -- to
the reference itself (rows a12, a13 of SURVEY.md section 8); the builder-written second statements (fp64 torch grid,
linear-field reproduction) stay in tests/test_gpu_parity.py beside it.

What "equal" means here:
  * forward and Jacobian, fp32 table: BIT FOR BIT, reference kernel == C oracle == product, level by level -- with one
    documented platform effect: the reference evaluates `exp2f(level * S)` on the device (gridencoder.cu:124), i.e. with
    ROCm's libm here and CUDA's on the reference's own hardware, and ROCm's value is 1 ulp from glibc's at some levels
    (level 11 of the bound-2 BASELINE table).  Oracle and product take glibc's (the product computes the per-level
    constants on the host).  At such a level the test requires bit-equality with the oracle's exp2f moved by one ulp
    (`O.grid_exp2f_nudged`) -- i.e. the ONLY difference is that one constant -- and counts how many levels needed it;
  * forward and Jacobian, at::Half table: bit for bit against the product (the reference accumulates in Half, rounding
    product and running sum at every corner; so does the product);
  * backward: atomics in both, so 1e-4 relative with an absolute floor of a few fp32 ulps of the largest sums.
"""
import importlib
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.fixture(scope="module")
def ref():
    from oracle import build_ref as br
    br.build()
    try:
        return br.load("gridencoder")
    except ImportError as e:                                     # never built: /root/reference is not on this machine
        pytest.skip(f"oracle/_ref not built: {e}")


@pytest.fixture(scope="module")
def prod():
    from enerf_amd import ext as e
    from enerf_amd.ext import build as eb
    eb.build(verbose=False)
    e.activate()
    yield importlib.import_module("_gridencoder")
    for n in e.MODULES:
        sys.modules.pop(n, None)


def _forward(m, x, emb, offsets, D, C, L, S, Hb, gridtype, jac=True):
    B = x.shape[0]
    out = torch.empty(L, B, C, device=DEV, dtype=emb.dtype)
    j = torch.empty(B, L * D * C, device=DEV, dtype=emb.dtype) if jac else torch.empty(1, device=DEV, dtype=emb.dtype)
    m.grid_encode_forward(x, emb, offsets, out, B, D, C, L, S, Hb, jac, j, gridtype)
    return out, j


def _oracle_as_the_reference_platform(ref_out, x, emb, offsets, S, Hb, gridtype, L):
    """The oracle's forward with, per level, the exp2f nudge (0 / +1 / -1 ulp) under which it equals the reference kernel
    bit for bit.  Returns (outputs, dy_dx, {level: ulps})."""
    out, jac = O.grid_encode_forward(x, emb, offsets, S, Hb, True, gridtype)
    nudged = {}
    for l in range(L):
        if np.array_equal(out[l], ref_out[l]):
            continue
        for ulps in (1, -1):
            with O.grid_exp2f_nudged(l, ulps):
                o2, j2 = O.grid_encode_forward(x, emb, offsets, S, Hb, True, gridtype)
            if np.array_equal(o2[l], ref_out[l]):
                B, D = x.shape
                C = emb.shape[1]
                out[l] = o2[l]
                jac.reshape(B, L, D * C)[:, l] = j2.reshape(B, L, D * C)[:, l]
                nudged[l] = ulps
                break
        else:
            raise AssertionError(f"level {l}: the reference kernel's output equals the oracle's under no exp2f within 1 ulp "
                                 f"(max diff {np.abs(out[l] - ref_out[l]).max():.3e})")
    return out, jac, nudged


@pytest.mark.parametrize("D,C,gridtype", [(3, 2, 0), (3, 1, 0), (3, 4, 0), (3, 8, 1), (2, 2, 0), (2, 4, 1), (3, 2, 1)])
def test_small_tables_three_ways(ref, prod, D, C, gridtype):
    offsets, pls = O.grid_offsets(input_dim=D, num_levels=8, level_dim=C, base_resolution=4, log2_hashmap_size=10,
                                  desired_resolution=160)
    S = float(np.log2(pls)); Hb = 4; L = 8; B = 777
    rng = np.random.default_rng(50 + C + 10 * D)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, D)).astype(np.float32)
    x[0] = 0.0; x[1] = 1.0; x[2, 0] = 1.2; x[3, 1] = -0.1           # corners + out-of-range (rows of zeros)
    cx, ce, co = cu(x), cu(emb), cu(offsets)
    r_out, r_jac = (t.cpu().numpy() for t in _forward(ref, cx, ce, co, D, C, L, S, Hb, gridtype))
    p_out, p_jac = (t.cpu().numpy() for t in _forward(prod, cx, ce, co, D, C, L, S, Hb, gridtype))
    o_out, o_jac, nudged = _oracle_as_the_reference_platform(r_out, x, emb, offsets, S, Hb, gridtype, L)
    assert len(nudged) <= 1, nudged
    assert np.array_equal(r_out, o_out) and np.array_equal(r_jac, o_jac)
    g_out, g_jac = O.grid_encode_forward(x, emb, offsets, S, Hb, True, gridtype)          # glibc's constants: the product's
    assert np.array_equal(p_out, g_out) and np.array_equal(p_jac, g_jac)
    # backward (both scatter with atomics) and the input gradient, on the reference platform's constants
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    ge_r, gi_r = torch.zeros_like(ce), torch.zeros(B, D, device=DEV)
    ref.grid_encode_backward(cu(g), cx, ce, co, ge_r, B, D, C, L, S, Hb, True, cu(r_jac), gi_r, gridtype)
    ge_p, gi_p = torch.zeros_like(ce), torch.zeros(B, D, device=DEV)
    prod.grid_encode_backward(cu(g), cx, ce, co, ge_p, B, D, C, L, S, Hb, True, cu(p_jac), gi_p, gridtype)
    ge_o, gi_o = O.grid_encode_backward(g, x, emb, offsets, S, Hb, g_jac, gridtype)
    scale = np.abs(ge_o).max()
    np.testing.assert_allclose(ge_p.cpu().numpy(), ge_o, rtol=1e-4, atol=4e-6 * scale)
    np.testing.assert_array_equal(gi_p.cpu().numpy(), gi_o)                                # one thread per element, in order
    if not nudged:
        np.testing.assert_allclose(ge_r.cpu().numpy(), ge_o, rtol=1e-4, atol=4e-6 * scale)
        np.testing.assert_allclose(gi_r.cpu().numpy(), gi_o, rtol=1e-5, atol=1e-6 * np.abs(gi_o).max())
    else:
        for l, ulps in nudged.items():
            with O.grid_exp2f_nudged(l, ulps):
                ge_n, _ = O.grid_encode_backward(g, x, emb, offsets, S, Hb, None, gridtype)
            np.testing.assert_allclose(ge_r.cpu().numpy(), ge_n, rtol=1e-4, atol=4e-6 * scale)


@pytest.mark.parametrize("bound", [2, 3])
def test_baseline_tables_three_ways(ref, prod, bound):
    """BASELINE's tables (L16 F2 T2^19, desired resolution 2048 * bound; 6.3 / 6.5 M rows), 100 000 random points."""
    offsets, pls = O.grid_offsets(desired_resolution=2048 * bound)
    S = float(np.log2(pls)); L = 16; C = 2; B = 100000
    rng = np.random.default_rng(70 + bound)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float32)
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    cx, ce, co = cu(x), cu(emb), cu(offsets)
    r_out, r_jac = (t.cpu().numpy() for t in _forward(ref, cx, ce, co, 3, C, L, S, 16, 0))
    p_out, p_jac = (t.cpu().numpy() for t in _forward(prod, cx, ce, co, 3, C, L, S, 16, 0))
    o_out, o_jac, nudged = _oracle_as_the_reference_platform(r_out, x, emb, offsets, S, 16, 0, L)
    assert len(nudged) <= 2, nudged
    assert np.array_equal(r_out, o_out) and np.array_equal(r_jac, o_jac)
    g_out, g_jac = O.grid_encode_forward(x, emb, offsets, S, 16, True, 0)
    assert np.array_equal(p_out, g_out) and np.array_equal(p_jac, g_jac)
    for l in range(L):                                           # product == reference kernel wherever the constant agrees
        assert l in nudged or np.array_equal(p_out[l], r_out[l]), l
    g = rng.normal(size=(L, B, C)).astype(np.float32)
    dummy = torch.empty(1, device=DEV)
    ge_r, ge_p = torch.zeros_like(ce), torch.zeros_like(ce)
    ref.grid_encode_backward(cu(g), cx, ce, co, ge_r, B, 3, C, L, S, 16, False, dummy, dummy, 0)
    prod.grid_encode_backward(cu(g), cx, ce, co, ge_p, B, 3, C, L, S, 16, False, dummy, dummy, 0)
    ge_r, ge_p = ge_r.cpu().numpy(), ge_p.cpu().numpy()
    for l in range(L):
        a, b = int(offsets[l]), int(offsets[l + 1])
        if l in nudged:
            continue
        scale = np.abs(ge_r[a:b]).max()
        np.testing.assert_allclose(ge_p[a:b], ge_r[a:b], rtol=1e-4, atol=4e-6 * scale, err_msg=f"level {l}")


def test_training_shaped_batch_binned_backward_vs_reference_kernel(ref, prod):
    """133 120 samples in ray order (runs of neighbouring points: the product's binned backward with in-wave run
    aggregation and record lists) against the reference's one-atomic-per-corner kernel."""
    bound = 3
    offsets, pls = O.grid_offsets(desired_resolution=2048 * bound)
    S = float(np.log2(pls)); L = 16; C = 2; R = 4096; K = 65; B = R * K // 2
    rng = np.random.default_rng(5)
    o = rng.uniform(0.3, 0.7, (R // 2, 1, 3)); d = rng.normal(size=(R // 2, 1, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    t = (np.arange(K)[None, :, None] + rng.uniform(0, 1, (R // 2, 1, 1))) * (1.0 / 1024)
    x = np.clip(o + d * t, 0, 1).reshape(-1, 3).astype(np.float32)[:B]
    emb = rng.uniform(-1e-4, 1e-4, (int(offsets[-1]), C)).astype(np.float32)
    g = (rng.normal(size=(L, B, C)) * 1e-3).astype(np.float32)
    cx, ce, co, cg = cu(x), cu(emb), cu(offsets), cu(g)
    r_out, _ = _forward(ref, cx, ce, co, 3, C, L, S, 16, 0, jac=False)
    p_out, _ = _forward(prod, cx, ce, co, 3, C, L, S, 16, 0, jac=False)
    same = [bool(torch.equal(r_out[l], p_out[l])) for l in range(L)]
    assert sum(same) >= L - 2, same
    dummy = torch.empty(1, device=DEV)
    ge_r, ge_p = torch.zeros_like(ce), torch.zeros_like(ce)
    ref.grid_encode_backward(cg, cx, ce, co, ge_r, B, 3, C, L, S, 16, False, dummy, dummy, 0)
    prod.grid_encode_backward(cg, cx, ce, co, ge_p, B, 3, C, L, S, 16, False, dummy, dummy, 0)
    ge_r, ge_p = ge_r.cpu().numpy(), ge_p.cpu().numpy()
    for l in range(L):
        if not same[l]:
            continue
        a, b = int(offsets[l]), int(offsets[l + 1])
        scale = np.abs(ge_r[a:b]).max()
        np.testing.assert_allclose(ge_p[a:b], ge_r[a:b], rtol=1e-4, atol=1e-5 * scale, err_msg=f"level {l}")


@pytest.mark.parametrize("C", [2, 4])
def test_half_table_forward_bit_equal_backward_close(ref, prod, C):
    """The at::Half instantiation (grid.py:38-39: the table's half copy under autocast): the reference accumulates the
    eight corners in Half, rounding the product and the running sum every time; the product does the same."""
    offsets, pls = O.grid_offsets(level_dim=C, desired_resolution=2048)
    S = float(np.log2(pls)); L = 16; B = 50000
    rng = np.random.default_rng(61)
    emb = rng.uniform(-1, 1, (int(offsets[-1]), C)).astype(np.float16)
    x = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    cx, ce, co = cu(x), cu(emb), cu(offsets)
    r_out, r_jac = _forward(ref, cx, ce, co, 3, C, L, S, 16, 0)
    p_out, p_jac = _forward(prod, cx, ce, co, 3, C, L, S, 16, 0)
    same = [bool(torch.equal(r_out[l], p_out[l])) for l in range(L)]
    assert sum(same) >= L - 2, same                               # (the exp2f levels again)
    rj, pj = r_jac.view(B, L, -1), p_jac.view(B, L, -1)
    for l in range(L):
        assert not same[l] or torch.equal(rj[:, l], pj[:, l]), l
    g = rng.normal(size=(L, B, C)).astype(np.float16)
    ge_r, ge_p = torch.zeros_like(ce), torch.zeros_like(ce)
    gi_r, gi_p = torch.zeros(B, 3, device=DEV, dtype=torch.half), torch.zeros(B, 3, device=DEV, dtype=torch.half)
    ref.grid_encode_backward(cu(g), cx, ce, co, ge_r, B, 3, C, L, S, 16, True, r_jac, gi_r, 0)
    prod.grid_encode_backward(cu(g), cx, ce, co, ge_p, B, 3, C, L, S, 16, True, r_jac, gi_p, 0)
    # grad_inputs in Half (gridencoder.cu:331-339, `result += grad * dy_dx` on at::Half): the SOURCE rounds the product to
    # Half, then the sum (Half's operators go through float) -- that is what the product does, exactly.  The reference as
    # hipcc builds it does not: -ffp-contract=fast narrows both operations to f16 and fuses them into one v_fma_f16
    # (single rounding; tests/refcheck/half_diag.py: 99.99 % of its entries equal that emulation, 48 % the source's).  A
    # compiler's choice, not the reference's arithmetic: held to a few half ulps.
    G, J = cu(g), r_jac.view(B, L, 3, C)
    src = torch.zeros(B, 3, device=DEV, dtype=torch.half)
    mass = torch.zeros(B, 3, device=DEV)                           # sum of |terms|: what a rounding of a partial sum scales with
    for l in range(L):
        for ch in range(C):
            term = G[l, :, ch].float()[:, None] * J[:, l, :, ch].float()
            src = (src.float() + term.half().float()).half()
            mass += term.abs()
    assert torch.equal(gi_p, src)
    assert bool(((gi_r.float() - gi_p.float()).abs() <= 2.0 ** -9 * mass + 1e-3).all())
    ge_o, _ = O.grid_encode_backward(g.astype(np.float32), x, emb.astype(np.float32), offsets, S, 16)
    # packed-half atomics: every add rounds to half, in whatever order -- both sides are held to the fp32 oracle
    for got in (ge_r, ge_p):
        err = np.abs(got.float().cpu().numpy() - ge_o)
        assert float(np.quantile(err, 0.999)) < 3e-2 and np.median(err) < 2e-3
