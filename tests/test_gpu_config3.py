"""BASELINE configs[3] (spiral1-shaped: bound 3, 65 536 rays per step ray-sharded over 8 GPUs = 8192 rays per rank, RCCL
all-reduce of the hash-grid / MLP gradients) at its PER-RANK shape on the one GPU a test box has:

  * 8192 rays through both data-parallel tails (chunked all-reduce + Adam per piece; reduce-scatter -> Adam on the slice
    -> all-gather), each driven natively (`_finish_native`: csrc/dp_tail.hip on the library's own RCCL communicator) and
    through torch.distributed (`_finish_distributed` / `_finish_sharded`), on the real backend (RCCL) with a world of
    one rank, against the single-process step (which takes the fused record-list optimizer instead);
  * 2 ranks x 8192 rays (gloo, both ranks on this device) against the single-process 16 384-ray step: per-step sample /
    ray counters add up bit for bit, the losses average to the whole batch's loss, the post-Adam parameters agree and
    an evaluation render of the replicas gives the single-process image.  The jitter of a training march is seeded by
    the ray's index in its own batch (raymarching.cu:349-350), so shard-vs-whole equality is tested with
    `TrainHarness.perturb = False`, and over the window before the first sample budget exists (with a budget the
    marcher drops the rays that overflow it -- per rank in a sharded run, per batch in the whole one: not the same rays).
    Past that window the two tails are compared with each other and the replicas must stay bit-identical.

What is left of configs[3] that no test here can reach: eight real GPUs (xGMI, RCCL rings)."""
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BOUND, RAYS_PER_RANK = 3, 8192


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(n_batches, n_rays, seed=21):
    from test_gpu_training import _batches
    return _batches(n_batches, n_rays, BOUND, seed=seed)


def _model():
    from enerf_amd.network import NeRFNetwork
    torch.manual_seed(0)
    return NeRFNetwork(encoding="hashgrid", bound=BOUND, cuda_ray=True, out_dim_color=3).to(DEV)


def _run(h, data, steps, first=0):
    losses, counters = [], []
    m = h.model
    for i in range(first, first + steps):
        nxt = data[(i + 1) % len(data)]
        losses.append(float(h.step_rgb(*data[i % len(data)], next_rays=(nxt[0], nxt[1]))))
        slot = getattr(m, "rendered_counter_slot", None)
        slot = (m.local_step - 1) % 16 if slot is None else slot
        counters.append(m.step_counter[slot].cpu().numpy().copy())
    torch.cuda.synchronize()
    return losses, np.stack(counters)


def _params(m):
    return {n: p.detach().cpu().clone() for n, p in m.named_parameters()}


def _eval_image(m):
    from enerf_amd import scene
    inds = torch.arange(0, scene.H * scene.W, 53, device=DEV)
    ro, rd = scene.pixel_rays(scene.pose(3), inds, DEV)
    m.eval()
    with torch.no_grad():
        img = m.render(ro, rd, staged=False, bg_color=None, perturb=False)["image"].cpu()
    m.train()
    return img


def _rccl_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from enerf_amd.trainer import TrainHarness
        data = _data(4, RAYS_PER_RANK)
        # every tail of the harness on torch.distributed (the library's own RCCL tail of rounds 3 / 4 was retired in round 5:
        # slower than this one in every measurement and untestable at world 2 on one GPU)
        for tag, dp, mode in (("single", 1, None), ("allreduce_torch", 2, "allreduce"),
                              ("sharded_torch", 2, "sharded"), ("sharded_fused", 2, "sharded")):
            model = _model()
            h = TrainHarness(model, lr=1e-2, occupancy="synthetic", world=dp)   # dp = 2: the data-parallel tail runs
            if mode:
                h.comm_mode = mode
                h.fused_sharded = tag.startswith("sharded_fused")   # (the other sharded row: the dense reduce-scatter tail)
            losses, counters = _run(h, data, 40)
            out[tag] = (losses, counters, _params(model), int(model.mean_count))
    finally:
        dist.destroy_process_group()


def test_configs3_rank_shape_through_both_tails_on_rccl():
    """8192 rays at bound 3 -- one rank's share of configs[3] -- for 40 steps (two sample-budget windows) through the
    chunked all-reduce tail and the sharded tail on RCCL (world of one rank: averaging is the identity), against the
    single-process step: the same rays are marched (counters bit-exact), the same losses and weights come out."""
    import torch.multiprocessing as mp
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, _port(), out), nprocs=1, join=True)
    la, ca, pa, ma = out["single"]
    assert ca[:, 1].max() == RAYS_PER_RANK and ma > 0
    for tag in ("allreduce_torch", "sharded_torch", "sharded_fused"):
        lb, cb, pb, mb = out[tag]
        assert np.array_equal(ca, cb) and ma == mb, tag
        assert np.abs(np.array(la) - np.array(lb)).max() <= 1e-4 * np.abs(la).max(), tag
        for n, a in pa.items():
            assert float((a - pb[n]).abs().mean()) <= 1e-3 * float(a.abs().mean()), (tag, n)


def _gloo_worker(rank, world, port, mode, cold_steps, more_steps, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from enerf_amd.trainer import TrainHarness
        torch.cuda.set_device(0)
        lo, hi = rank * RAYS_PER_RANK, (rank + 1) * RAYS_PER_RANK
        data = [tuple((t[:, lo:hi] if t.dim() == 3 else t[lo:hi]).contiguous() for t in b)
                for b in _data(4, world * RAYS_PER_RANK)]
        model = _model()
        h = TrainHarness(model, lr=1e-2, occupancy="synthetic", world=world)
        h.perturb = False
        h.comm_mode = "sharded" if mode.startswith("sharded") else mode
        h.fused_sharded = mode != "sharded_dense"         # "sharded": this rank's slice keeps its record lists (OwnerRange)
        seen = []
        if mode == "sharded":
            inner = h._finish_sharded_fused
            h._finish_sharded_fused = lambda *a, **k: (seen.append(1), inner(*a, **k))[1]
        l0, c0 = _run(h, data, cold_steps)
        p0, img0 = _params(model), _eval_image(model)
        l1, c1 = _run(h, data, more_steps, first=cold_steps)
        if mode == "sharded":
            assert len(seen) == cold_steps + more_steps, len(seen)      # every step went through the fused sharded tail
        out[(mode, rank)] = (l0, c0, p0, img0, l1, c1, _params(model), int(model.mean_count))
    finally:
        dist.destroy_process_group()


def test_configs3_two_ranks_of_8192_rays_equal_the_16384_ray_step():
    import torch.multiprocessing as mp
    from enerf_amd.trainer import TrainHarness
    cold, more = 15, 25
    out = mp.Manager().dict()
    for mode in ("allreduce", "sharded", "sharded_dense"):
        mp.spawn(_gloo_worker, args=(2, _port(), mode, cold, more, out), nprocs=2, join=True)
    # the whole batch in one process
    data = _data(4, 2 * RAYS_PER_RANK)
    model = _model()
    h = TrainHarness(model, lr=1e-2, occupancy="synthetic")
    h.perturb = False
    ls, cs = _run(h, data, cold)
    ps, imgs = _params(model), _eval_image(model)
    for mode in ("allreduce", "sharded", "sharded_dense"):
        (l0, c0, p0, i0, l1, c1, q0, m0), (lr1, cr1, p1, i1, lr2, cr2, q1, m1) = out[(mode, 0)], out[(mode, 1)]
        # replicas: identical weights after every window, one agreed sample budget
        assert all(torch.equal(p0[n], p1[n]) for n in p0) and all(torch.equal(q0[n], q1[n]) for n in q0), mode
        assert m0 == m1 > 0 and torch.equal(i0, i1)
        # shards add up to the whole batch: samples and rays marched per step, bit for bit
        assert np.array_equal(c0 + cr1, cs), mode
        assert int(cs[:, 1].max()) == 2 * RAYS_PER_RANK
        # the mean of the two shard losses is the whole batch's loss; same weights after 15 Adam steps; same picture
        both = 0.5 * (np.array(l0) + np.array(lr1))
        assert np.abs(both - np.array(ls)).max() <= 1e-4 * np.abs(ls).max(), mode
        for n, a in ps.items():
            assert float((a - p0[n]).abs().mean()) <= 1e-3 * float(a.abs().mean()), (mode, n)
        assert float((imgs - i0).abs().max()) <= 2e-3, mode
    # past the cold window (sample budget agreed by MAX all-reduce): both tails train the same model
    for other in ("sharded", "sharded_dense"):
        qa, qb = out[("allreduce", 0)][6], out[(other, 0)][6]
        for n, a in qa.items():
            assert float((a - qb[n]).abs().mean()) <= 1e-3 * float(a.abs().mean()), (other, n)
        la, lb = np.array(out[("allreduce", 0)][4]), np.array(out[(other, 0)][4])
        assert np.isfinite(la).all() and np.abs(la - lb).max() <= 1e-3 * np.abs(la).max(), other
