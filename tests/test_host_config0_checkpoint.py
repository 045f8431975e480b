"""Host-side closure of two gaps the round-2 review named:

* BASELINE configs[0] end to end: `NeRFNetwork(encoding="frequency", encoding_dir="frequency", cuda_ray=False)`, 256 rays x
  512 stratified samples through `run()` + MSE + backward + Adam for three steps, against a fixture minted by running
  the reference's own classes (nerf/network.py:10-101, encoding.py:5-43, nerf/renderer.py:150-278; minted by
  oracle/make_golden.py:gold_config0).
* the reference's checkpoint dict (nerf/utils.py:1295-1415): a `.pth` written by the reference's own
  `Trainer.save_checkpoint(full=True)` (gold_checkpoint) loads into the harness here, the next optimizer step
  reproduces the reference's next step, and what the harness saves has the reference's structure.
"""
import gzip
import io
import os

import numpy as np
import pytest
import torch

from util import GOLDEN, golden, det_fill_, t, assert_close


def test_configs0_frequency_network_three_training_steps(cpu_oracle_backend):
    from enerf_amd.network import NeRFNetwork
    g = golden("ref_config0_steps")
    model = NeRFNetwork(encoding="frequency", encoding_dir="frequency", bound=3, cuda_ray=False, out_dim_color=1)
    assert model.in_dim == int(g["in_dim"]) == 39 and model.in_dim_dir == int(g["in_dim_dir"]) == 39
    det_fill_(list(model.parameters()), 91, -0.25, 0.25)
    opt = torch.optim.Adam(model.get_params(0.005), betas=(0.9, 0.99), eps=1e-15)
    o, d, target = t(g["rays_o"]), t(g["rays_d"]), t(g["target"])
    assert o.shape == (1, 256, 3)
    model.train()
    losses = []
    for it in range(3):
        torch.manual_seed(500 + it)                     # the jitter comes from torch's host generator (renderer.py:186)
        opt.zero_grad(set_to_none=True)
        out = model.render(o, d, staged=False, bg_color=None, perturb=True, num_steps=512, upsample_steps=0,
                           out_dim_color=1)
        loss = torch.nn.functional.mse_loss(out["image"], target)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if it == 0:
            assert_close(out["image"], g["image0"], rtol=1e-5, atol=1e-6)
            assert_close(out["depth"], g["depth0"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(losses, g["losses"], rtol=1e-5)
    for k, v in model.state_dict().items():
        if k.endswith(".weight"):
            # three Adam steps with eps = 1e-15 turn a last-bit difference of a tiny gradient into a visible one
            # on isolated weights: compare in bulk (lr = 0.005, so an un-updated weight would be off by 0.015)
            ref = g["p_" + k.replace(".", "_")]
            diff = np.abs(v.detach().numpy() - ref)
            assert np.median(diff) <= 1e-6 and diff.max() <= 2e-3, (k, np.median(diff), diff.max())


def _reference_checkpoint():
    with gzip.open(os.path.join(GOLDEN, "ref_checkpoint_freq.pth.gz")) as f:
        return torch.load(io.BytesIO(f.read()), map_location="cpu", weights_only=False)


def _freq_harness(optimizer=None):
    from enerf_amd.network import NeRFNetwork
    from enerf_amd.trainer import TrainHarness
    model = NeRFNetwork(encoding="frequency", encoding_dir="frequency", bound=1, cuda_ray=True, out_dim_color=3)
    h = TrainHarness(model, lr=0.01, occupancy="learned", optimizer=optimizer)
    h.set_lr_scheduler(lambda o: torch.optim.lr_scheduler.LambdaLR(o, lambda it: 0.1 ** min(it / 10, 1)))
    return model, h


def _one_step(model, h, g):
    x, d = t(g["x"]), t(g["d"])
    h.opt.zero_grad(set_to_none=True)
    sigma, color = model(x, d)
    (sigma.mean() + (color ** 2).mean()).backward()
    step = getattr(h.opt, "step_now", h.opt.step)
    step()
    h.lr_scheduler.step()


@pytest.mark.parametrize("optimizer", ["torch", "fused"])
def test_reference_checkpoint_resumes_and_round_trips(cpu_oracle_backend, tmp_path, optimizer):
    from enerf_amd.optim import FusedAdam
    ref = _reference_checkpoint()
    g = golden("ref_checkpoint_resumed")
    model, h = _freq_harness(FusedAdam if optimizer == "fused" else None)
    missing, unexpected = h.load_checkpoint(ref)
    assert not missing and not unexpected
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref["model"][k]), k
    assert model.mean_count == 1200 and model.mean_density == 0.37
    assert (h.epoch, h.global_step) == (2, 3) and h.stats["loss"] == [0.5, 0.25]
    assert h.lr_scheduler.last_epoch == 3
    st = h.opt.state[model.sigma_net[0].weight]
    assert int(st["step"]) == 3 and torch.equal(st["exp_avg"], ref["optimizer"]["state"][0]["exp_avg"])
    # what the harness writes back is the reference's dict: same keys at every level, same tensors
    path = h.save_checkpoint(str(tmp_path / "resaved.pth"), full=True)
    mine = torch.load(path, map_location="cpu", weights_only=False)
    assert list(mine.keys()) == list(ref.keys())
    assert list(mine["model"].keys()) == list(ref["model"].keys())
    assert all(torch.equal(mine["model"][k], ref["model"][k]) for k in ref["model"])
    assert mine["optimizer"]["state"].keys() == ref["optimizer"]["state"].keys()
    for k, st_ref in ref["optimizer"]["state"].items():
        st_mine = mine["optimizer"]["state"][k]
        assert st_mine.keys() == st_ref.keys()
        assert torch.is_tensor(st_mine["step"]) and st_mine["step"].dtype == st_ref["step"].dtype
        assert all(torch.equal(st_mine[n], st_ref[n]) for n in st_ref)
    for gm, gr in zip(mine["optimizer"]["param_groups"], ref["optimizer"]["param_groups"]):
        assert set(gr) <= set(gm) and gm["params"] == gr["params"]
        assert gm["lr"] == gr["lr"] and tuple(gm["betas"]) == tuple(gr["betas"]) and gm["eps"] == gr["eps"]
    assert mine["lr_scheduler"]["last_epoch"] == ref["lr_scheduler"]["last_epoch"]
    assert mine["lr_scheduler"]["_last_lr"] == ref["lr_scheduler"]["_last_lr"]
    assert (mine["epoch"], mine["global_step"], mine["mean_count"], mine["mean_density"]) == \
        (ref["epoch"], ref["global_step"], ref["mean_count"], ref["mean_density"])
    # ... and a plain torch.optim.Adam (what the reference would resume with) accepts it and can step
    ref_model, _ = _freq_harness()
    ref_opt = torch.optim.Adam(ref_model.get_params(0.01), betas=(0.9, 0.99), eps=1e-15)
    ref_opt.load_state_dict(mine["optimizer"])
    for p in ref_model.parameters():
        p.grad = torch.zeros_like(p)
    ref_opt.step()
    # the resumed run's next step is the reference's next step (learning rate from the restored schedule)
    _one_step(model, h, g)
    assert abs(h.opt.param_groups[0]["lr"] - float(g["lr_after"])) <= 1e-12
    for k, v in model.state_dict().items():
        if k.endswith(".weight"):
            assert_close(v, g["p_" + k.replace(".", "_")], rtol=1e-5, atol=1e-6, msg=k)


def test_model_only_and_bare_state_dict_loads(cpu_oracle_backend):
    ref = _reference_checkpoint()
    model, h = _freq_harness()
    h.load_checkpoint(ref, model_only=True)
    assert model.mean_count == 1200 and h.global_step == 0 and not h.opt.state
    model2, h2 = _freq_harness()
    h2.load_checkpoint(dict(ref["model"]))              # 'model' not in the dict: a bare state_dict (utils.py:1369-1372)
    assert all(torch.equal(v, ref["model"][k]) for k, v in model2.state_dict().items())
    assert model2.mean_count != 1200


def test_lr_schedule_reaches_every_optimizer_route():
    """LambdaLR (main_nerf.py:212) writes param_groups[...]['lr']; FusedAdam.step_now's host fall-back (CPU tensors)
    must follow it step by step exactly as torch.optim.Adam does."""
    from enerf_amd.optim import FusedAdam
    torch.manual_seed(0)
    w0 = torch.randn(37, 5)
    grads = [torch.randn(37, 5) for _ in range(6)]
    outs = []
    for cls in (torch.optim.Adam, FusedAdam):
        w = torch.nn.Parameter(w0.clone())
        opt = cls([{"params": [w], "lr": 1e-2}], betas=(0.9, 0.99), eps=1e-15)
        sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** min(it / 4, 1))
        opt._opt_called = True
        lrs = []
        for gr in grads:
            w.grad = gr.clone()
            (opt.step_now if cls is FusedAdam else opt.step)()
            sched.step()
            lrs.append(opt.param_groups[0]["lr"])
        outs.append((w.detach().clone(), lrs))
    assert outs[0][1] == outs[1][1] and outs[0][1][-1] == pytest.approx(1e-3)
    assert_close(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-7)


def test_reference_checkpoint_file_loads_without_the_unpickler_and_a_bad_optimizer_only_warns(cpu_oracle_backend, tmp_path):
    """load_checkpoint(path) reads with weights_only=True: the file the reference's Trainer wrote holds nothing but
    tensors, numbers, strings, lists and dicts.  An optimizer state that does not fit warns and leaves the rest loaded
    (nerf/utils.py:1396-1401), and an update_extra_state read-back left in flight by the run before the load must not
    overwrite the loaded mean_density."""
    path = str(tmp_path / "ref.pth")
    with gzip.open(os.path.join(GOLDEN, "ref_checkpoint_freq.pth.gz")) as f, open(path, "wb") as out:
        out.write(f.read())
    model, h = _freq_harness()
    model._pending_density_stats = ("stale",)            # (whatever it holds: the load drops it)
    missing, unexpected = h.load_checkpoint(path)
    assert not missing and not unexpected and model._pending_density_stats is None
    assert model.mean_count == 1200 and model.mean_density == 0.37 and h.global_step == 3
    ref = _reference_checkpoint()
    ref["optimizer"]["param_groups"] = ref["optimizer"]["param_groups"][:1]       # a group short: load_state_dict raises
    model2, h2 = _freq_harness()
    with pytest.warns(UserWarning, match="optimizer"):
        h2.load_checkpoint(ref)
    assert h2.global_step == 3 and h2.lr_scheduler.last_epoch == 3 and not h2.opt.state


def test_harness_precision_switches_are_named_apart(cpu_oracle_backend):
    """fp16=True is the reference's regime (autocast + GradScaler); the bf16 regime has its own name and is refused
    where the fused path cannot serve it -- no silent remap of one onto the other."""
    from enerf_amd.network import NeRFNetwork
    from enerf_amd.trainer import TrainHarness
    model = NeRFNetwork(encoding="frequency", encoding_dir="frequency", bound=1, cuda_ray=True, out_dim_color=3)
    h = TrainHarness(model, occupancy="learned", fp16=True)
    assert h.fp16 and not h.amp_bf16 and h.scaler is not None and "mlp_precision" not in model.__dict__
    with pytest.raises(ValueError):
        TrainHarness(model, occupancy="learned", amp="bf16")
    with pytest.raises(ValueError):
        TrainHarness(model, occupancy="learned", amp="fp8")
