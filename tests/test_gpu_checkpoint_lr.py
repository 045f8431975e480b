"""Rows f1 / f4 loose ends on the device: the reference's LambdaLR schedule (main_nerf.py:212) stepped through the fused
table optimizer (`enerf_grid_adam_from_records`), and a training run saved in the reference's checkpoint format
(nerf/utils.py:1295-1351), resumed in a fresh harness and continued."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(fuse_table, lr_lambda=None, seed=0):
    from enerf_amd.network import NeRFNetwork
    from enerf_amd.trainer import TrainHarness
    torch.manual_seed(seed)
    model = NeRFNetwork(encoding="hashgrid", bound=2, cuda_ray=True, out_dim_color=3).to(DEV)
    h = TrainHarness(model, lr=1e-2, occupancy="synthetic")
    h.fuse_table_adam = fuse_table
    if lr_lambda is not None:
        h.set_lr_scheduler(lambda o: torch.optim.lr_scheduler.LambdaLR(o, lr_lambda))
    return model, h


def _params(m):
    return {n: p.detach().clone() for n, p in m.named_parameters()}


def test_lambda_lr_steps_through_the_fused_table_optimizer():
    """LambdaLR(0.1 ** min(iter / iters, 1)) over 24 steps: the fused record-list optimizer (one GPU) and the dense
    gradient + multi-tensor Adam route read the scheduled rate every step and train the same weights; with the
    schedule the weights differ from a constant-rate run (i.e. the rate really arrives in the kernel)."""
    from test_gpu_training import _batches
    data = _batches(4, 4096, 2)
    sched = lambda it: 0.1 ** min(it / 12, 1)                                   # noqa: E731
    runs = {}
    for tag, fuse, lam in (("fused", True, sched), ("dense", False, sched), ("constant", True, None)):
        model, h = _setup(fuse, lam)
        losses = [float(h.step_rgb(*data[i % 4])) for i in range(24)]
        torch.cuda.synchronize()
        runs[tag] = (losses, _params(model), h.opt.param_groups[0]["lr"])
    assert runs["fused"][2] == runs["dense"][2] == pytest.approx(1e-3) and runs["constant"][2] == 1e-2
    la, lb = np.array(runs["fused"][0]), np.array(runs["dense"][0])
    assert np.abs(la - lb).max() <= 1e-4 * np.abs(la).max()
    for n, a in runs["fused"][1].items():
        b, c = runs["dense"][1][n], runs["constant"][1][n]
        assert float((a - b).abs().mean()) <= 1e-3 * float(a.abs().mean()), n
        assert float((a - c).abs().mean()) > 20 * float((a - b).abs().mean()), n


def test_checkpoint_in_reference_format_resumes_training(tmp_path):
    """20 steps -> save_checkpoint(full=True) -> fresh model + harness -> load_checkpoint -> 12 more steps, against the
    uninterrupted 32 steps: model, occupancy state, sample budget, Adam moments / step counts and the schedule all
    travel through the reference's dict, so the continued run is the same training (to the coarse levels' atomics)."""
    from test_gpu_training import _batches
    data = _batches(4, 2048, 2)
    sched = lambda it: 0.1 ** min(it / 100, 1)                                  # noqa: E731
    model, h = _setup(True, sched)
    first = [float(h.step_rgb(*data[i % 4])) for i in range(20)]
    path = h.save_checkpoint(str(tmp_path / "ngp_ep0001.pth"), full=True)
    at_save = (model.local_step, model.iter_density)
    rest = [float(h.step_rgb(*data[i % 4])) for i in range(20, 32)]
    torch.cuda.synchronize()
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert {"epoch", "global_step", "stats", "mean_count", "mean_density", "optimizer", "lr_scheduler", "scaler",
            "model"} == set(ckpt)
    assert ckpt["global_step"] == 20 and ckpt["mean_count"] > 0
    st0 = ckpt["optimizer"]["state"][0]
    assert torch.is_tensor(st0["step"]) and float(st0["step"]) == 20.0 and st0["exp_avg"].shape == model.encoder.embeddings.shape

    model2, h2 = _setup(True, sched, seed=123)                                  # different random init: all of it is replaced
    missing, unexpected = h2.load_checkpoint(path)
    assert not missing and not unexpected
    assert h2.global_step == 20 and model2.mean_count == ckpt["mean_count"]
    assert h2.opt.param_groups[0]["lr"] == pytest.approx(1e-2 * 0.1 ** 0.2)
    # (the reference's dict does not carry the renderer's two Python counters -- a resumed reference run restarts its
    # step-counter ring and its density-update count; set here so that the continuation is the uninterrupted run)
    model2.local_step, model2.iter_density = at_save
    rest2 = [float(h2.step_rgb(*data[i % 4])) for i in range(20, 32)]
    torch.cuda.synchronize()
    assert np.abs(np.array(rest) - np.array(rest2)).max() <= 1e-3 * np.abs(rest).max(), (rest, rest2)
    pa, pb = _params(model), _params(model2)
    for n, a in pa.items():
        assert float((a - pb[n]).abs().mean()) <= 2e-3 * float(a.abs().mean()), n
    assert first[0] > rest2[-1]
