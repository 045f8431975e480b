"""enerf_amd's host-side mirror of the reference interface (wrappers, networks, renderer.run, event loss, rays) against
fixtures minted from the reference's own Python (oracle/make_golden.py).  Both sides use the C oracle for the native
calls, so what is compared here is the Python-level contract: layouts, permutes, padding, init, sequencing."""
import numpy as np
import pytest
import torch

from util import golden, det_fill_, t, assert_close


def test_grid_encoder_wrapper(cpu_oracle_backend):
    from enerf_amd.gridencoder import GridEncoder
    g = golden("ref_grid_wrapper")
    enc = GridEncoder(input_dim=3, num_levels=6, level_dim=2, base_resolution=4, log2_hashmap_size=9,
                      desired_resolution=96)
    assert (enc.offsets.numpy() == g["offsets"]).all()
    assert enc.per_level_scale == float(g["per_level_scale"])
    det_fill_([enc.embeddings], 11)
    x = t(g["x"]).requires_grad_(True)
    y = enc(x, bound=1)
    (y * t(g["w"])).sum().backward()
    assert_close(y, g["y"], rtol=0, atol=0)
    assert_close(enc.embeddings.grad, g["grad_embeddings"], rtol=0, atol=0)
    assert_close(x.grad, g["grad_x"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("bound", [1, 2, 3])
def test_grid_offsets_match_reference_sizing(bound):
    from enerf_amd.gridencoder import level_offsets
    g = golden(f"ref_grid_offsets_b{bound}")
    pls = np.exp2(np.log2(2048 * bound / 16) / 15)
    assert pls == float(g["per_level_scale"])
    off = level_offsets(3, 16, pls, 16, 19)
    assert (off == g["offsets"]).all() and int(off[-1]) == int(g["n_rows"])
    if bound == 3:
        assert int(off[-1]) == 6507840
    if bound == 2:
        assert int(off[-1]) == 6328848


@pytest.mark.parametrize("deg", [4, 8])
def test_sh_encoder_wrapper(cpu_oracle_backend, deg):
    from enerf_amd.shencoder import SHEncoder
    g = golden(f"ref_sh_wrapper_d{deg}")
    enc = SHEncoder(degree=deg)
    d = t(g["d"]).requires_grad_(True)
    y = enc(d)
    (y * t(g["w"])).sum().backward()
    assert_close(y, g["y"], rtol=0, atol=0)
    assert_close(d.grad, g["grad_d"], rtol=0, atol=0)


@pytest.mark.parametrize("k,i,o,B", [(2, 32, 16, 100), (3, 32, 3, 128)])
def test_ffmlp_wrapper(cpu_oracle_backend, k, i, o, B):
    from enerf_amd.ffmlp import FFMLP
    g = golden(f"ref_ffmlp_wrapper_k{k}")
    net = FFMLP(i, o, 64, k)
    assert_close(net.weights, g["weights"], rtol=0, atol=0)     # manual_seed(42) U(+-sqrt(3/64)) init
    net.train()
    x = t(g["x"]).requires_grad_(True)
    y = net(x)
    assert y.shape == (B, o)
    (y * t(g["gw"])).sum().backward()
    assert_close(y, g["y"], rtol=1e-6, atol=1e-6)
    assert_close(net.weights.grad, g["grad_weights"], rtol=1e-5, atol=1e-5)
    assert_close(x.grad, g["grad_x"], rtol=1e-5, atol=1e-6)
    net.eval()
    with torch.no_grad():
        assert_close(net(x.detach()), g["y_inf"], rtol=1e-6, atol=1e-6)


def _make_network():
    from enerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=False, out_dim_color=3)
    det_fill_(list(model.parameters()), 41)
    return model


def test_network_forward_density_color(cpu_oracle_backend):
    g = golden("ref_network")
    model = _make_network().eval()
    names = [n for n, _ in model.named_parameters()]
    assert names == ["encoder.embeddings", "sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight",
                     "color_net.1.weight", "color_net.2.weight"]
    assert set(model.state_dict().keys()) >= {"aabb_train", "aabb_infer", "encoder.offsets", "encoder.embeddings"}
    x, d = t(g["x"]), t(g["d"])
    with torch.no_grad():
        sigma, color = model(x, d)
        dens = model.density(x)
        cm = model.color(x, d, mask=t(g["mask"]), geo_feat=dens["geo_feat"])
    assert_close(sigma, g["sigma"], rtol=1e-5, atol=1e-6)
    assert_close(color, g["color"], rtol=1e-5, atol=1e-6)
    assert_close(dens["geo_feat"], g["geo_feat"], rtol=1e-5, atol=1e-6)
    assert_close(cm, g["color_masked"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("up", [0, 8])
def test_renderer_run_eval(cpu_oracle_backend, up):
    g = golden(f"ref_run_up{up}")
    model = _make_network().eval()
    with torch.no_grad():
        out = model.render(t(g["rays_o"]), t(g["rays_d"]), staged=False, bg_color=None, perturb=False, num_steps=24,
                           upsample_steps=up, out_dim_color=3)
    assert_close(out["image"], g["image"], rtol=1e-4, atol=1e-5)
    assert_close(out["depth"], g["depth"], rtol=1e-4, atol=1e-5)


def test_renderer_run_train_gradients(cpu_oracle_backend):
    g = golden("ref_run_train")
    model = _make_network().train()
    out = model.render(t(g["rays_o"]), t(g["rays_d"]), staged=False, bg_color=torch.full((3,), 0.25), perturb=False,
                       num_steps=24, upsample_steps=0, out_dim_color=3)
    ((out["image"] ** 2).sum() + out["depth"].sum()).backward()
    assert_close(out["image"], g["image"], rtol=1e-4, atol=1e-5)
    assert_close(model.sigma_net[0].weight.grad, g["g_sigma0"], rtol=1e-3, atol=1e-5)
    assert_close(model.color_net[2].weight.grad, g["g_color2"], rtol=1e-3, atol=1e-5)
    assert_close(model.encoder.embeddings.grad[:4920], g["g_emb_l0"], rtol=1e-3, atol=1e-6)
    assert_close(model.encoder.embeddings.grad.abs().sum(), g["g_emb_sum"], rtol=1e-4)


def test_network_ff(cpu_oracle_backend):
    from enerf_amd.network_ff import NeRFNetwork
    g = golden("ref_network_ff")
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=False)
    det_fill_([model.encoder.embeddings], 51)
    assert_close(model.sigma_net.weights, g["w_sigma"], rtol=0, atol=0)
    assert_close(model.color_net.weights, g["w_color"], rtol=0, atol=0)
    model.eval()
    with torch.no_grad():
        sigma, rgb = model(t(g["x"]), t(g["d"]))
    assert_close(sigma, g["sigma"], rtol=1e-5, atol=1e-6)
    assert_close(rgb, g["rgb"], rtol=1e-5, atol=1e-6)


CFG = {
    "luma_linlog": dict(use_luma=1, linlog=1, C_thres=0.2, event_only=1),
    "rgb_linlog": dict(use_luma=0, linlog=1, C_thres=0.2, event_only=1),
    "luma_log": dict(use_luma=1, linlog=0, C_thres=0.2, event_only=1),
    "rgb_log": dict(use_luma=0, linlog=0, C_thres=0.2, event_only=1),
    "normed": dict(use_luma=1, linlog=1, C_thres=-1, event_only=1),
    "both": dict(use_luma=1, linlog=1, C_thres=0.2, event_only=0),
}


@pytest.mark.parametrize("name", list(CFG))
def test_event_step_loss_and_grads(name):
    from enerf_amd.events import EventOptions, train_step_events
    g = golden("ref_event_loss")
    imgs = [t(g["img1"]), t(g["img2"]), t(g["img3"])]

    class FakeModel:
        def __init__(self):
            self.calls, self.last = 0, []

        def render(self, o, d, **kw):
            im = imgs[self.calls % 3].clone().requires_grad_(True)
            self.calls += 1
            self.last.append(im)
            return {"image": im, "depth": im[..., 0]}

    opt = EventOptions(**CFG[name])
    m = FakeModel()
    B, N = g["pols"].shape
    z = torch.zeros(B, N, 3)
    data = {"images": t(g["frames"]), "rays_evs_o1": z, "rays_evs_d1": z, "rays_evs_o2": z, "rays_evs_d2": z,
            "pols": t(g["pols"]), "rays_o": z, "rays_d": z}
    loss, delta = train_step_events(m, data, opt)
    loss.backward()
    assert_close(loss, g[f"{name}_loss"], rtol=1e-6, atol=1e-7)
    assert_close(delta, g[f"{name}_delta"], rtol=1e-6, atol=1e-7)
    g1 = m.last[0].grad if m.last[0].grad is not None else torch.zeros_like(imgs[0])
    g2 = m.last[1].grad if m.last[1].grad is not None else torch.zeros_like(imgs[0])
    assert_close(g1, g[f"{name}_g1"], rtol=1e-5, atol=1e-8)
    assert_close(g2, g[f"{name}_g2"], rtol=1e-5, atol=1e-8)
    if not CFG[name]["event_only"]:
        assert_close(m.last[2].grad, g["both_g3"], rtol=1e-5, atol=1e-8)


def test_event_utils_and_rays():
    from enerf_amd.events import rgb_to_luma, lin_log, get_rays, get_event_rays
    g = golden("ref_event_utils")
    x = t(g["x"])
    assert_close(rgb_to_luma(x, True), g["luma_esim"], rtol=1e-6)
    assert_close(rgb_to_luma(x, False), g["luma_v2e"], rtol=1e-6)
    assert_close(lin_log(x * 255, 20), g["linlog"], rtol=1e-6)
    r = golden("ref_rays")
    out = get_rays(t(r["pose"]), tuple(r["intr"]), int(r["H"]), int(r["W"]), -1)
    assert_close(out["rays_o"], r["rays_o"], rtol=1e-6)
    assert_close(out["rays_d"], r["rays_d"], rtol=1e-6, atol=1e-7)
    ev = get_event_rays(t(r["xs"]), t(r["ys"]), t(r["c2w_b"]), t(r["c2w_a"]), tuple(r["intr"]))
    for k in ("rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2"):
        assert_close(ev[k], r[k], rtol=1e-6, atol=1e-7)


def test_trunc_exp_and_freq_encoder():
    from enerf_amd.activation import trunc_exp
    from enerf_amd.encoding import FreqEncoder
    g = golden("ref_misc")
    x = t(g["x"]).requires_grad_(True)
    y = trunc_exp(x)
    y.sum().backward()
    assert_close(y, g["trunc_exp"], rtol=1e-6)
    assert_close(x.grad, g["trunc_exp_grad"], rtol=1e-6)
    f = FreqEncoder(input_dim=3, max_freq_log2=5, N_freqs=6)
    assert f.output_dim == 39
    assert_close(f(t(g["p"])), g["freq"], rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------ checkpoint compatibility
@pytest.mark.parametrize("which", ["network", "network_ff"])
@pytest.mark.parametrize("bound", [1, 2, 3])
def test_state_dict_layout_matches_reference(which, bound):
    """tests/golden/ref_state_dict_schema.json was minted from the reference's own classes (oracle/make_golden.py):
    key -> (shape, dtype) of model.state_dict() with cuda_ray on.  This repo's mirrors must produce the identical
    layout, so a reference `.pth` ('model' entry, nerf/utils.py save_checkpoint) loads with strict=True and a
    checkpoint written here loads into the reference."""
    import json
    import os
    if which == "network":
        from enerf_amd.network import NeRFNetwork
        model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True, out_dim_color=3)
    else:
        from enerf_amd.network_ff import NeRFNetwork
        model = NeRFNetwork(encoding="hashgrid", bound=bound, cuda_ray=True)
    with open(os.path.join(os.path.dirname(__file__), "golden", "ref_state_dict_schema.json")) as f:
        schema = json.load(f)[f"{which}_bound{bound}"]
    mine = {k: [list(v.shape), str(v.dtype)] for k, v in model.state_dict().items()}
    assert mine == schema
    # a reference-format checkpoint (random contents of the reference's shapes / dtypes) loads strictly
    g = torch.Generator().manual_seed(bound)
    ckpt = {}
    for k, (shape, dtype) in schema.items():
        dt = getattr(torch, dtype.split(".")[1])
        if k == "encoder.offsets":
            ckpt[k] = model.state_dict()[k].clone()
        elif dt.is_floating_point:
            ckpt[k] = torch.rand(shape, generator=g).to(dt)
        else:
            ckpt[k] = torch.randint(0, 100, shape, generator=g).to(dt)
    missing, unexpected = model.load_state_dict(ckpt, strict=True)
    assert not missing and not unexpected
    for k, v in model.state_dict().items():
        assert torch.equal(v, ckpt[k]), k


NEG_CFG = {
    "neg_luma": dict(use_luma=1, C_thres=0.2, event_only=1, w_no_ev=0.7, epoch=2, epoch_start_noEvLoss=0),
    "neg_rgb": dict(use_luma=0, C_thres=0.2, event_only=1, w_no_ev=1.0, epoch=2, epoch_start_noEvLoss=0),
    "neg_normed": dict(use_luma=1, C_thres=-1, event_only=1, w_no_ev=2.0, epoch=2, epoch_start_noEvLoss=0),
    "neg_both": dict(use_luma=1, C_thres=0.2, event_only=0, w_no_ev=0.7, epoch=2, epoch_start_noEvLoss=0),
    "neg_gated": dict(use_luma=1, C_thres=0.2, event_only=1, w_no_ev=0.7, epoch=1, epoch_start_noEvLoss=1),
}


@pytest.mark.parametrize("name", list(NEG_CFG))
def test_event_step_with_negative_event_sampling(name):
    """--negative_event_sampling (nerf/utils.py:548-565): the no-event pair of renders and its hinge on the (lin-)log
    intensity change, against the reference's own Trainer.train_step_events (loss, delta, the gradient reaching every
    render's image, how many renders were made)."""
    from enerf_amd.events import EventOptions, train_step_events
    g = golden("ref_no_event_loss")
    imgs = [t(g["img1"]), t(g["img2"]), t(g["img3"])]

    class FakeModel:
        def __init__(self):
            self.calls, self.last = 0, []

        def render(self, o, d, **kw):
            im = imgs[self.calls % 3][:, : o.shape[1]].clone().requires_grad_(True)
            self.calls += 1
            self.last.append(im)
            return {"image": im, "depth": im[..., 0]}

    opt = EventOptions(negative_event_sampling=True, linlog=True, **NEG_CFG[name])
    m = FakeModel()
    B, N = g["pols"].shape
    z, zn = torch.zeros(B, N, 3), torch.zeros(B, int(g["Nn"]), 3)
    data = {"images": t(g["frames"]), "rays_evs_o1": z, "rays_evs_d1": z, "rays_evs_o2": z, "rays_evs_d2": z,
            "pols": t(g["pols"]), "rays_o": z, "rays_d": z, "rays_no_evs_o1": zn, "rays_no_evs_d1": zn,
            "rays_no_evs_o2": zn, "rays_no_evs_d2": zn}
    loss, delta = train_step_events(m, data, opt)
    loss.backward()
    assert m.calls == int(g[f"{name}_calls"])
    assert_close(loss, g[f"{name}_loss"], rtol=1e-6, atol=1e-7)
    assert_close(delta, g[f"{name}_delta"], rtol=1e-6, atol=1e-7)
    for i, im in enumerate(m.last):
        gi = im.grad if im.grad is not None else torch.zeros_like(im)
        assert_close(gi, g[f"{name}_g{i}"], rtol=1e-5, atol=1e-8)
