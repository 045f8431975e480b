"""BASELINE.json configs at their stated sizes, HIP path (through the C ABI) against the CPU oracle.

configs[2]  mocapDesk2-shaped event step: bound 2, two renders of 4096 rays from poses one degree apart sharing one
            backward -- images, sample counters (bit-exact), loss and every parameter gradient against the same step run
            on the oracle backend (autograd-driven, the reference's structure).
configs[4]  eds00-shaped full frame: bound 2, 640 x 480 = 307 200 rays through network_ff (FFMLP, bf16 MFMA): the
            marched samples of all rays bit-exact against the oracle marcher, sigma / rgb against the oracle's rounded
            FFMLP on a strided subsample, the compositing against the oracle's round-by-round inference loop fed with the
            same network outputs, and the whole-frame schedule against the reference's round schedule, bit for bit.
"""
import math

import numpy as np
import pytest
import torch

from oracle import oracle as O
from util import synthetic_density_grid, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda"
H = 128


def _scene_bits(bound):
    return O.packbits(synthetic_density_grid(bound, H).reshape(-1), 0.01)


def _fill(model, seed):
    g = torch.Generator().manual_seed(seed)
    model.encoder.embeddings.data.copy_(torch.rand(model.encoder.embeddings.shape, generator=g) * 2 - 1)


# ---------------------------------------------------------------------------------------------------- configs[2]
def _event_batch(n, dev):
    from enerf_amd import scene
    g = torch.Generator().manual_seed(77)
    inds = torch.randint(0, scene.H * scene.W, (n,), generator=g)
    k = 5
    (o1, d1) = scene.pixel_rays(scene.pose(k), inds, "cpu")
    (o2, d2) = scene.pixel_rays(scene.pose(k + 1.0 / (360.0 / 32)), inds, "cpu")       # one degree further on the circle
    pols = torch.where(torch.rand(1, n, generator=g) < 0.5, -1.0, 1.0)
    data = {"images": torch.zeros(1, n, 3), "rays_evs_o1": o1, "rays_evs_d1": d1, "rays_evs_o2": o2, "rays_evs_d2": d2,
            "pols": pols}
    return {k_: v.to(dev) for k_, v in data.items()}


def test_config2_event_step_4096_rays_vs_oracle(monkeypatch, mlp32_mode):
    import enerf_amd.raymarching as rmod, enerf_amd.gridencoder as gmod, enerf_amd.shencoder as smod
    from oracle import backend as ob
    from enerf_amd import events, fused_render
    from enerf_amd.events import EventOptions
    from enerf_amd.network import NeRFNetwork
    bound, N = 2, 4096
    bits = _scene_bits(bound)
    opt = EventOptions(C_thres=0.2, use_luma=True, linlog=True, event_only=True)
    bg = torch.tensor([[[0.35, 0.6, 0.15]]])

    # the oracle side: the reference's structure (two autograd renders + loss.backward) on the CPU backend
    with monkeypatch.context() as mp:
        mp.setattr(rmod, "_backend", ob.raymarching_backend); mp.setattr(rmod, "_DEVICE", "cpu")
        mp.setattr(gmod, "_backend", ob.gridencoder_backend); mp.setattr(smod, "_backend", ob.shencoder_backend)
        torch.manual_seed(0)
        ref = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
        _fill(ref, 5)
        ref.density_bitfield.copy_(torch.from_numpy(bits))
        state = {k: v.clone() for k, v in ref.state_dict().items()}
        ref.train()
        data = _event_batch(N, "cpu")
        out = {}
        orig_loss = events.event_loss
        mp.setattr(events, "event_loss", lambda a, b, p, o: (out.update(im1=a.detach().clone(), im2=b.detach().clone()),
                                                             orig_loss(a, b, p, o))[1])
        # every nn.Linear call's input X and output gradient dY (both renders): the weight gradient is dY^T X summed
        # over ~2 x 130 k samples; its fp64 reduction and the sum of the terms' magnitudes bound what fp32 summation in
        # ANY order may differ by (see below)
        terms = {}
        knife = {}
        hooks = []
        for net_name in ("sigma_net", "color_net"):
            for li, layer in enumerate(getattr(ref, net_name)):
                hidden = li != len(getattr(ref, net_name)) - 1

                def fwd_hook(mod, inp, outp, key=f"{net_name}.{li}.weight", hidden=hidden):
                    x = inp[0].detach()
                    outp.register_hook(lambda g, key=key, x=x: terms.setdefault(key, []).append((x, g.detach())))
                    if hidden:
                        # knife-edge samples: a ReLU input closer to zero than the device's forward error can be -- half
                        # the rigorous bound u x sum |w| |x| of that unit (u = 2^-16 per split-bf16 product, 4 eps32 for
                        # the fp32 chains); with 192 hidden units per sample that flags ~0.5 % of the samples
                        pre = outp.detach()
                        mag = x.abs() @ mod.weight.detach().abs().t()
                        u_ = 2.0 ** -16 if mlp32_mode == "split-bf16" else 4 * 2.0 ** -23
                        knife.setdefault(key.split(".")[0], []).append(((pre.abs() < 0.5 * u_ * mag) & (pre != 0)).any(dim=1))
                hooks.append(layer.register_forward_hook(fwd_hook))
        # every encoder call's (normalised) query points and the gradient of its output: the table gradient of a row is
        # the sum of corner weight x dL/dfeature over the samples of both renders whose cells touch it
        enc_terms = []

        enc_fwd = []                     # the query points in FORWARD call order (the gradient hooks fire in reverse)

        def enc_hook(mod, inp, outp):
            xq = inp[0].detach()
            enc_fwd.append(xq)
            outp.register_hook(lambda g, xq=xq: enc_terms.append((xq, g.detach())))
        hooks.append(ref.encoder.register_forward_hook(enc_hook))
        loss_ref, _ = events.train_step_events(ref, data, opt, bg_color=bg)
        loss_ref.backward()
        for h_ in hooks:
            h_.remove()
        grads_ref = {n: p.grad.clone() for n, p in ref.named_parameters()}
        counter_ref = ref.step_counter.clone()

    # the product side: the closed render backward of the event step, cold window (no sample budget yet)
    model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
    model.load_state_dict(state)
    model.to(DEV).train()
    data = _event_batch(N, DEV)
    got = {}
    orig_raw = fused_render.render_train_raw

    def spy(*a, **k):
        im, ctx = orig_raw(*a, **k)
        got.setdefault("images", []).append(im)
        return im, ctx

    monkeypatch.setattr(fused_render, "render_train_raw", spy)
    loss, _ = events.train_step_events_manual(model, data, opt, bg_color=bg.to(DEV))
    torch.cuda.synchronize()
    assert len(got["images"]) == 2
    assert torch.equal(model.step_counter.cpu(), counter_ref)                       # samples / rays of both renders
    assert int(counter_ref[0, 0]) > 50000 and int(counter_ref[1, 0]) > 50000 and int(counter_ref[0, 1]) == N
    assert_close(got["images"][0].view(-1, 3), out["im1"].view(-1, 3), rtol=1e-4, atol=2e-5)
    assert_close(got["images"][1].view(-1, 3), out["im2"].view(-1, 3), rtol=1e-4, atol=2e-5)
    assert abs(float(loss.detach()) - float(loss_ref.detach())) <= 1e-4 * abs(float(loss_ref.detach()))
    # MLP weight gradients: dW = sum over 2 x ~130 k samples of dY_i x_i^T, whose terms cancel to ~1 % of their
    # magnitudes.  The oracle side's own per-sample terms are reduced in fp64 here (G64) together with the sum of their
    # magnitudes (A) and the largest single term (T): a correct fp32 evaluation stays within a few eps32 x A of G64
    # (summation order) plus a few whole terms (a ReLU whose pre-activation is within an ulp of zero opens on one side
    # and not on the other: with 2 x 130 k x 64 hidden units per layer a handful always are; the split-bf16 arithmetic,
    # whose forward error is ~10x the fp32 chains', leaves a few more).  The bar per entry is north_star's 1e-4 relative
    # + 64 eps32 x A + n_flip x T with n_flip = 2 (fp32 MFMA) / 8 (split-bf16).
    eps32 = float(np.finfo(np.float32).eps)
    n_flip = 8 if mlp32_mode == "split-bf16" else 2
    worst = {}

    def largest_term(pairs):
        out = None
        for x, dy in pairs:
            x, dy = x.to(DEV).abs(), dy.to(DEV).abs()
            for k in range(0, x.shape[0], 8192):
                m = (dy[k:k + 8192, :, None] * x[k:k + 8192, None, :]).amax(0)
                out = m if out is None else torch.maximum(out, m)
        return out.double().cpu()

    for n, p in model.named_parameters():
        r = grads_ref[n]
        got_g = p.grad.cpu()
        if n in terms:
            assert len(terms[n]) == 2, (n, len(terms[n]))                  # two renders
            g64 = sum(dy.double().t() @ x.double() for x, dy in terms[n])
            mag = sum(dy.double().abs().t() @ x.double().abs() for x, dy in terms[n])
            top = largest_term(terms[n])
            assert float((g64 - r.double()).abs().max()) <= 64 * eps32 * float(mag.max())     # (the hooks saw the real terms)
            err = (got_g.double() - g64).abs()
            bar = 1e-4 * g64.abs() + 64 * eps32 * mag + n_flip * top + 1e-12
            worst[n] = {"err/bar": float((err / bar).max()), "bar/max|G|": float(bar.max() / g64.abs().max()),
                        "err/max|G|": float(err.max() / g64.abs().max())}
            assert bool((err <= bar).all()), (n, worst[n])
        else:
            # the table: per row a sum over the few hundred (coarse levels) to a handful (fine levels) of samples in its
            # cells, accumulated by fp32 atomics in an arbitrary order on the oracle side, in fp64 per tile here
            # Bounded per entry like the MLP gradients, from the oracle side's own terms: A = sum over the samples of
            # corner weight x (largest |dL/dfeature| of that sample) -- the scale any evaluation of the sample's
            # feature gradient is accurate against (its 32 entries are sums with cancellation of the same chain).  Every
            # entry must hold north_star's 1e-4 against |reference| + A, except the rows of the few samples whose ReLU
            # pre-activation lies within round-off of zero (a knife-edge sample changes its whole feature gradient, i.e.
            # 8 rows x 16 levels per render); those are counted, and held to A itself.
            assert len(enc_terms) == 2, len(enc_terms)
            enc = ref.encoder
            emb_np = enc.embeddings.detach().numpy()
            offs = enc.offsets.numpy()
            S_ = float(np.log2(enc.per_level_scale))
            A = np.zeros_like(emb_np)
            for xq, gq in enc_terms:
                xn = ((xq.reshape(-1, 3) + bound) / (2 * bound)).numpy().astype(np.float32)
                gmax = gq.reshape(-1, 32).abs().amax(dim=1, keepdim=True).expand(-1, 32)
                g_lbc = gmax.reshape(-1, 16, 2).permute(1, 0, 2).contiguous().numpy()
                A += O.grid_encode_backward(g_lbc, xn, emb_np, offs, S_, enc.base_resolution)[0]
            A = torch.from_numpy(A)
            # rows a knife-edge sample touches (per render: its 8 corners on each of the 16 levels)
            exempt = np.zeros(emb_np.shape[0], bool)
            n_knife = 0
            for k_, xq in enumerate(enc_fwd):                 # render k_ of the forward pass, with its own hidden layers
                flags = None
                for net_name in ("sigma_net", "color_net"):
                    per_call = knife[net_name]
                    per_render = len(per_call) // 2
                    for f_ in per_call[k_ * per_render:(k_ + 1) * per_render]:
                        flags = f_ if flags is None else (flags | f_)
                xn = ((xq.reshape(-1, 3) + bound) / (2 * bound)).numpy().astype(np.float32)
                ind = flags.reshape(-1, 1).float().expand(-1, 32).reshape(-1, 16, 2).permute(1, 0, 2).contiguous().numpy()
                n_knife += int(flags.sum())
                exempt |= O.grid_encode_backward(ind, xn, emb_np, offs, S_, enc.base_resolution)[0].any(axis=1)
            exempt = torch.from_numpy(exempt)
            err = (got_g - r).abs()
            tight = 1e-4 * (r.abs() + A) + 1e-12
            over = (err > tight).any(dim=1)
            worst[n] = {"knife_edge_samples": n_knife, "rows_they_touch": int(exempt.sum()),
                        "rows_over_1e-4_elsewhere": int((over & ~exempt).sum()),
                        "max err/(|r|+A) elsewhere": float((err / (r.abs() + A + 1e-30))[~exempt].max()),
                        "max err/max|r|": float(err.max() / r.abs().max())}
            assert n_knife <= 4000, worst[n]                            # (of ~270 k samples)
            assert int((over & ~exempt).sum()) == 0, (n, worst[n])          # every other row: 1e-4, no allowance
            assert bool((err <= 1e-4 * r.abs() + A + 1e-12).all()), (n, worst[n])
    print("configs[2] gradient bars (MLP: per-entry error / bar, and both relative to the largest entry; table: / max):", worst)
    assert float(grads_ref["encoder.embeddings"].abs().max()) > 0


# ---------------------------------------------------------------------------------------------------- configs[4]
def _frame_rays(dev):
    from enerf_amd import scene
    inds = torch.arange(scene.H * scene.W)
    o, d = scene.pixel_rays(scene.pose(3), inds, "cpu")
    return o[0].contiguous().to(dev), d[0].contiguous().to(dev)


def _oracle_ff_forward(model, x, d, bound):
    """sigma, rgb of nerf/network_ff.py's forward on the oracle (bf16-rounded FFMLP, SURVEY.md Appendix A: sigma net
    32 -> 64 -> 64 -> 16, colour net [SH 16 | geo 15 | 0] -> 64 -> 64 -> 64 -> 16 (3 used), exp / sigmoid heads)."""
    enc = model.encoder
    emb = enc.embeddings.detach().cpu().numpy()
    offsets = enc.offsets.cpu().numpy()
    S = float(np.log2(enc.per_level_scale))
    xn = ((x + bound) / (2 * bound)).astype(np.float32)
    feats, _ = O.grid_encode_forward(xn, emb, offsets, S, enc.base_resolution)                # [L,B,C]
    feats = np.ascontiguousarray(feats.transpose(1, 0, 2).reshape(len(x), -1))
    bf = lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()                      # noqa: E731
    ws = model.sigma_net.weights.detach().to(torch.bfloat16).float().cpu().numpy()
    h, _ = O.ffmlp_forward(bf(feats), ws, 32, 16, 64, 2, 0, 6, rnd=1, want_buffer=False)
    sigma = np.exp(h[:, 0])
    sh, _ = O.sh_encode_forward(d.astype(np.float32), 4)
    cin = np.concatenate([sh, h[:, 1:], np.zeros((len(x), 1), np.float32)], axis=1)
    wc = model.color_net.weights.detach().to(torch.bfloat16).float().cpu().numpy()
    c, _ = O.ffmlp_forward(bf(cin), wc, 32, 16, 64, 3, 0, 6, rnd=1, want_buffer=False)
    return sigma, 1.0 / (1.0 + np.exp(-c[:, :3]))


def test_config4_full_frame_307200_rays_ffmlp_vs_oracle():
    from enerf_amd import frame
    from enerf_amd.network_ff import NeRFNetwork
    bound = 2
    C = 1 + math.ceil(math.log2(bound))
    bits = _scene_bits(bound)
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True).to(DEV).eval()
    model.encoder.embeddings.data.uniform_(-1, 1)
    g = torch.Generator(device=DEV).manual_seed(9)
    # weights large enough that some rays saturate (the termination rule is exercised), small enough to stay finite
    model.sigma_net.weights.data.copy_((torch.rand(model.sigma_net.weights.shape, generator=g, device=DEV) - 0.5) * 0.9)
    model.color_net.weights.data.copy_((torch.rand(model.color_net.weights.shape, generator=g, device=DEV) - 0.5) * 0.6)
    model.density_bitfield.copy_(torch.from_numpy(bits))
    model.density_scale = 40.0                      # dense enough that rays saturate: the termination rule is exercised
    ro, rd = _frame_rays(DEV)
    N = ro.shape[0]
    assert N == 307200
    tr = {}
    with torch.no_grad():
        depth, image = frame.render_frame(model, ro, rd, 1, trace=tr.update)
        torch.cuda.synchronize()
        # (a) the reference's round schedule on the same model.  Not bit for bit, and no two round schedules are: at
        # every round boundary the marcher restarts from rays_t, which composite_rays re-derives as a running fp32 sum
        # of the samples' real step lengths (raymarching.cu:869,897) -- occasionally an ulp off the marcher's own t, after
        # which that ray's remaining samples sit an ulp further along.  Measured on this frame: 1 pixel in 10^4 differs
        # between the reference schedule (~130 rounds) and either the 8-round schedule or the single pass, by <= 4e-7
        # with fp32 networks and <= 3e-5 here (a bf16 activation flips).
        d2, im2 = frame.render_rounds(model, ro, rd, 1)
        model.infer_batch_mult = 8
        d3, im3 = frame.render_rounds(model, ro, rd, 1)
    for other_d, other_im in ((d2, im2), (d3, im3)):
        diff = (image - other_im).abs()
        assert float(diff.max()) < 2e-3 and float((diff > 0).float().mean()) < 2e-3
        dd = (depth - other_d).abs()
        dd = dd[torch.isfinite(dd)]
        assert float(dd.max()) < 2e-4 and float((dd > 0).float().mean()) < 2e-3

    # (b) marching: all 307 200 rays against the oracle marcher, bit-exact
    o_np, d_np = ro.cpu().numpy(), rd.cpu().numpy()
    aabb = np.array([-bound] * 3 + [bound] * 3, np.float32)
    nears, fars = O.near_far_from_aabb(o_np, d_np, aabb, 0.2)
    assert np.array_equal(tr["nears"].cpu().numpy(), nears) and np.array_equal(tr["fars"].cpu().numpy(), fars)
    total = int(tr["counter"][0])
    M = tr["M"]
    ref = O.march_rays_train(o_np, d_np, bits, bound, 0.0, 1024, C, H, M, nears, fars, 0)
    assert np.array_equal(tr["counter"].cpu().numpy(), ref[4]) and np.array_equal(tr["rays"].cpu().numpy(), ref[3])
    assert total > 5_000_000 and int(ref[3][:, 2].max()) < 1024          # no ray at the loop's step cap
    for name, want in zip(("xyzs", "dirs", "deltas"), ref[:3]):
        assert np.array_equal(tr[name].cpu().numpy(), want), name

    # (c) the networks: every 97th sample against the oracle's bf16-rounded FFMLP
    pick = np.arange(0, total, 97)
    sig_ref, rgb_ref = _oracle_ff_forward(model, ref[0][pick], ref[1][pick], bound)
    sig = tr["sigmas"].cpu().numpy()[pick] / np.float32(model.density_scale)        # the frame stores scaled densities
    rgb = tr["rgbs"].cpu().numpy()[pick]
    ulp = 2.0 ** -8
    rel = np.abs(sig - sig_ref) / np.maximum(np.abs(sig_ref), 1e-6)
    assert np.mean(rel > 4 * ulp) < 0.02 and rel.max() < 0.15, (np.mean(rel > 4 * ulp), rel.max())
    err = np.abs(rgb - rgb_ref)
    assert np.mean(err > 4 * ulp) < 0.02 and err.max() < 0.06, (np.mean(err > 4 * ulp), err.max())
    assert sig.std() > 0 and rgb.std() > 0.01

    # (d) compositing: the oracle's round-by-round inference loop (march_rays / composite_rays / compact_rays) fed with
    # the GPU's sigma / rgb for exactly the samples it visits
    sig_all, rgb_all = tr["sigmas"].cpu().numpy(), tr["rgbs"].cpu().numpy()
    offs = ref[3][:, 1].astype(np.int64)
    ws = np.zeros(N, np.float32); dp = np.zeros(N, np.float32); im = np.zeros((N, 3), np.float32)
    alive = np.arange(N, dtype=np.int32); rt = nears.copy()
    taken_by_ray = np.zeros(N, np.int64)
    n_alive, taken = N, 0
    while taken < 1024 and n_alive > 0:
        n_step = max(min(8 * N // n_alive, 64), 1)
        Mi = n_alive * n_step
        x, dd, dl = O.march_rays(n_alive, n_step, alive, rt, o_np, d_np, bound, 0.0, 1024, C, H, bits, nears, fars, Mi, 0)
        # slot (n, k) of this round is sample taken_by_ray[ray] + k of that ray
        ray = alive[:n_alive].astype(np.int64)
        idx = (offs[ray] + taken_by_ray[ray])[:, None] + np.arange(n_step)[None, :]
        valid = dl.reshape(n_alive, n_step, 2)[:, :, 0] != 0
        idx = np.where(valid, idx, 0)
        assert np.array_equal(x.reshape(n_alive, n_step, 3)[valid], ref[0][idx[valid]])      # same samples, again
        s_round = np.where(valid, sig_all[idx], 0).astype(np.float32).reshape(-1)
        c_round = np.where(valid[..., None], rgb_all[idx], 0).astype(np.float32).reshape(-1, 3)
        O.composite_rays(n_alive, n_step, alive, rt, s_round, c_round, dl, ws, dp, im)
        taken_by_ray[ray] += valid.sum(1)
        taken += n_step
        alive, rt, n_alive = O.compact_rays(n_alive, alive, rt)
        alive, rt = alive[:n_alive].copy(), rt[:n_alive].copy()
    im_ref = im + (1 - ws)[:, None]
    assert_close(tr["weights_sum"], ws, rtol=1e-4, atol=2e-6)
    assert_close(image, im_ref, rtol=1e-4, atol=2e-5)
    hit = (fars < 1e30)
    dp_ref = np.clip(dp - nears, 0, None)[hit] / (fars - nears)[hit]
    assert_close(depth.cpu().numpy()[hit], dp_ref, rtol=1e-4, atol=2e-5)
    assert float((ws > 0.999).mean()) > 0.001                           # some rays did saturate
    assert int(tr["used"][0]) <= total


def test_frame_schedule_equals_round_schedule_fp32_nets():
    """Whole-frame pass vs the round schedule for the nn.Linear (fp32 MFMA) networks and a tensor background, 20 000
    rays, bound 3."""
    from enerf_amd import frame, scene
    from enerf_amd.network import NeRFNetwork
    torch.manual_seed(2)
    model = NeRFNetwork(encoding="hashgrid", bound=3, cuda_ray=True, out_dim_color=3).to(DEV).eval()
    model.encoder.embeddings.data.uniform_(-1, 1)
    scene.install_occupancy(model)
    g = torch.Generator(device=DEV).manual_seed(4)
    (ro, rd), _ = scene.training_batch(2, 20000, DEV, generator=g)
    ro, rd = ro[0].contiguous(), rd[0].contiguous()
    bg = torch.rand(3, device=DEV)
    with torch.no_grad():
        d1, i1 = frame.render_frame(model, ro, rd, bg)
        d0, i0 = frame.render_rounds(model, ro, rd, bg)
        out = model.render(ro[None], rd[None], staged=False, bg_color=bg, perturb=False)
    # (equal up to the round schedule's own t hand-over, see test_config4_...: a few pixels, <= 1e-6 with fp32 networks)
    assert float((i1 - i0).abs().max()) < 2e-6 and float(((i1 - i0).abs() > 0).float().mean()) < 2e-3
    dd = (d1 - d0).abs()
    assert float(dd[torch.isfinite(dd)].max()) < 2e-6
    assert torch.equal(out["image"].view(-1, 3), i1)                      # run_cuda takes the whole-frame path
    assert float(i1.std()) > 0.01
