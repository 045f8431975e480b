"""Ray-sharded data parallelism (enerf_amd/parallel.py) with world_size 2 over gloo on CPU: the native calls go through
the C oracle backend (tests only), so what is exercised is the N > 1 logic -- shard ranges, the two-bucket gradient
average, identical replicas after the optimizer step, and equivalence with the single-process full-batch step."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inject_oracle():
    import enerf_amd.raymarching as rm
    import enerf_amd.gridencoder as ge
    import enerf_amd.shencoder as sh
    from oracle import backend as ob
    rm._backend, rm._DEVICE, ge._backend, sh._backend = (ob.raymarching_backend, "cpu", ob.gridencoder_backend,
                                                         ob.shencoder_backend)


def _model_and_data(n_rays):
    from enerf_amd.network import NeRFNetwork
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import synthetic_density_grid, camera_rays
    from oracle import oracle as O
    torch.manual_seed(0)
    m = NeRFNetwork(encoding="hashgrid", bound=1, cuda_ray=True, out_dim_color=3)
    g = torch.Generator().manual_seed(1)
    m.encoder.embeddings.data.copy_(torch.rand(m.encoder.embeddings.shape, generator=g) * 0.2 - 0.1)
    bits = O.packbits(synthetic_density_grid(1).reshape(-1), 0.01)
    m.density_bitfield.copy_(torch.from_numpy(bits))
    o, d = camera_rays(n_rays, 5, 1)
    target = torch.rand(1, n_rays, 3, generator=g)
    return m, torch.from_numpy(o)[None], torch.from_numpy(d)[None], target


def _step(m, ro, rd, target, opt, avg, scale):
    m.train()
    opt.zero_grad(set_to_none=True)
    out = m.render(ro, rd, staged=False, bg_color=None, perturb=False, force_all_rays=True)
    # sum-reduced loss scaled so that the average over ranks equals the full-batch mean
    loss = ((out["image"] - target) ** 2).sum() * scale
    loss.backward()
    if avg is not None:
        avg()
    opt.step()
    return loss.detach()


def _worker(rank, world, port, n_rays, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    _inject_oracle()
    from enerf_amd import parallel
    r, w, _ = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    m, ro, rd, target = _model_and_data(n_rays)
    if rank == 1:   # perturb rank 1's replica: broadcast_state must repair it
        with torch.no_grad():
            m.sigma_net[0].weight.add_(1.0)
    parallel.broadcast_state(m, src=0)
    lo, hi = parallel.shard_range(n_rays, rank, world)
    opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
    avg = parallel.GradAverager(list(m.parameters()))
    for _ in range(2):
        _step(m, ro[:, lo:hi], rd[:, lo:hi], target[:, lo:hi], opt, avg, scale=world / (3.0 * n_rays))
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from enerf_amd.parallel import shard_range
    for n in (1, 7, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(600)
def test_two_rank_step_equals_single_process_step(tmp_path):
    n_rays, world = 48, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rays, str(tmp_path)), nprocs=world, join=True)
    s0 = torch.load(tmp_path / "rank0.pt")
    s1 = torch.load(tmp_path / "rank1.pt")
    for k in s0:
        if k in ("step_counter", "density_grid"):
            continue       # per-rank sample counters differ by construction (each rank marches its own rays)
        assert torch.equal(s0[k], s1[k]), f"replicas diverged in {k}"
    # single process, full batch, mean loss == the 2-rank average of per-shard sums
    sys.path.insert(0, ROOT)
    _inject_oracle()
    try:
        m, ro, rd, target = _model_and_data(n_rays)
        opt = torch.optim.Adam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        for _ in range(2):
            _step(m, ro, rd, target, opt, None, scale=1.0 / (3.0 * n_rays))
        ref = m.state_dict()
        for k in ("encoder.embeddings", "sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight",
                  "color_net.2.weight"):
            a, b = s0[k], ref[k]
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 2e-6, k
    finally:
        import importlib
        import enerf_amd.raymarching as rm, enerf_amd.gridencoder as ge, enerf_amd.shencoder as sh
        importlib.reload(rm); importlib.reload(ge); importlib.reload(sh)


def _tune_worker(rank, world, port, outdir):
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from enerf_amd import parallel
    from enerf_amd.trainer import TrainHarness
    parallel.init_from_env(backend="gloo")
    try:
        h = object.__new__(TrainHarness)                 # only what tune_comm reads: no renderer, no GPU
        h.model = torch.nn.Linear(2, 2)
        h.avg = object()
        h.update_interval = 16
        h.comm_chunks = 4
        h.comm_mode = "allreduce"
        h.prefetch_at = "forward"
        # rank 0 is slow with 1 piece, rank 1 with 8: the slowest rank decides, so both must settle on 2; the sharded
        # tail is slow on rank 1 only -- still rejected everywhere
        cost = {0: {1: 0.020, 2: 0.004, 8: 0.002}, 1: {1: 0.002, 2: 0.004, 8: 0.020}}[rank]
        seen = []

        def step(i):
            seen.append((i, h.comm_chunks, h.comm_mode))
            time.sleep(0.012 * rank if h.comm_mode == "sharded" else cost[h.comm_chunks])
        timings = h.tune_comm(step, candidates=(1, 2, 8), window=3)
        torch.save({"timings": timings, "chosen": h.comm_chunks, "mode": h.comm_mode, "tuned": h.tuned, "seen": seen},
                   os.path.join(outdir, f"tune{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_comm_tuning_slowest_rank_decides(tmp_path):
    """tune_comm over gloo, world_size 2: the timing of a candidate is the MAX over ranks, every rank sees the same
    numbers and makes the same choice; candidates run `window` consecutive steps each after one warm-up window."""
    import torch.multiprocessing as mp
    mp.spawn(_tune_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (torch.load(os.path.join(str(tmp_path), f"tune{r}.pt")) for r in (0, 1))
    assert a["timings"] == b["timings"] and set(a["timings"]) == {1, 2, 8}
    assert a["chosen"] == b["chosen"] == 2
    assert a["timings"][1] >= 19.0 and a["timings"][8] >= 19.0 and a["timings"][2] < 15.0       # ms per step
    # warm-up + 3 candidates, two windows of the sharded tail, then three windows (march placement) with what was chosen
    assert [c for _, c, _ in a["seen"]] == [1] * 6 + [2] * 3 + [8] * 3 + [2] * 15
    assert [m for _, _, m in a["seen"]] == ["allreduce"] * 12 + ["sharded"] * 6 + ["allreduce"] * 9
    assert [i for i, _, _ in a["seen"]] == list(range(27))
    assert set(a["tuned"]["prefetch_at_ms_per_step"]) == {"forward", "mlp_backward", "collectives"}
    assert a["mode"] == b["mode"] == "allreduce" and a["tuned"]["sharded_ms_per_step"] == b["tuned"]["sharded_ms_per_step"] >= 11.0


def _render_worker(rank, world, port, n_rays, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    _inject_oracle()
    from enerf_amd import parallel
    parallel.init_from_env(backend="gloo")
    m, ro, rd, _ = _model_and_data(n_rays)
    m.eval()
    out = parallel.render_sharded(m, ro, rd, bg_color=None, perturb=False)
    torch.save({k: v.clone() for k, v in out.items()}, os.path.join(outdir, f"render{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_render_equals_single_process_render(tmp_path):
    """Inference at N = 2: each rank renders its slice of the rays, all_gather of the tiles -> every rank holds the
    full frame, equal to the single-process render (an odd ray count exercises the padded last tile)."""
    n_rays, world = 51, 2
    port = _free_port()
    mp.spawn(_render_worker, args=(world, port, n_rays, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "render0.pt"), torch.load(tmp_path / "render1.pt")
    sys.path.insert(0, ROOT)
    _inject_oracle()
    try:
        m, ro, rd, _ = _model_and_data(n_rays)
        m.eval()
        with torch.no_grad():
            ref = m.render(ro, rd, staged=False, bg_color=None, perturb=False)
        for got in (r0, r1):
            assert got["image"].shape == (1, n_rays, 3) and got["depth"].shape == (1, n_rays)
            assert torch.equal(got["image"], ref["image"])
            assert torch.equal(got["depth"].nan_to_num(-7.0), ref["depth"].nan_to_num(-7.0))
    finally:
        import importlib
        import enerf_amd.raymarching as rm, enerf_amd.gridencoder as ge, enerf_amd.shencoder as sh
        importlib.reload(rm); importlib.reload(ge); importlib.reload(sh)


def _pack_dw(model):
    """flat dW buffer in fused_network's parameter order (sigma 0, sigma 1, colour 0, colour 1, colour 2)"""
    ps = [model.sigma_net[0].weight, model.sigma_net[1].weight, model.color_net[0].weight, model.color_net[1].weight,
          model.color_net[2].weight]
    return torch.cat([p.grad.reshape(-1) for p in ps]).clone()


def _tail_worker(rank, world, port, n_rays, mode, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(2)
    _inject_oracle()
    from enerf_amd import parallel
    from enerf_amd.optim import FusedAdam
    from enerf_amd.trainer import TrainHarness
    parallel.init_from_env(backend="gloo")
    m, ro, rd, target = _model_and_data(n_rays)
    h = TrainHarness(m, lr=1e-2, occupancy=None, world=world, optimizer=FusedAdam)
    h.comm_chunks, h.comm_mode = 3, mode
    lo, hi = parallel.shard_range(n_rays, rank, world)
    for _ in range(2):
        m.train()
        for p in m.parameters():
            p.grad = None
        out = m.render(ro[:, lo:hi], rd[:, lo:hi], staged=False, bg_color=None, perturb=False, force_all_rays=True)
        (((out["image"] - target[:, lo:hi]) ** 2).sum() * (world / (3.0 * n_rays))).backward()
        # hand the tail what the closed-form step hands it: (table gradient, flat MLP dW)
        h._raw_grads = (m.encoder.embeddings.grad, _pack_dw(m))
        (h._finish_sharded if mode == "sharded" else h._finish_distributed)()
    torch.save({k: v.clone() for k, v in m.state_dict().items()}, os.path.join(outdir, f"tail_{mode}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["allreduce", "sharded"])
def test_two_rank_closed_form_tails_equal_single_process_adam(tmp_path, mode):
    """TrainHarness._finish_distributed (table gradient all-reduced in 3 pieces, Adam per piece) and _finish_sharded
    (reduce-scatter -> Adam on the rank's slice -> all-gather) on 2 real ranks over gloo: replicas identical, and equal
    to one process stepping the full batch (the oracle backend computes the gradients; the tails are what is tested)."""
    n_rays, world = 40, 2
    mp.spawn(_tail_worker, args=(world, _free_port(), n_rays, mode, str(tmp_path)), nprocs=world, join=True)
    s0, s1 = (torch.load(tmp_path / f"tail_{mode}_{r}.pt") for r in (0, 1))
    for k in s0:
        if k not in ("step_counter", "density_grid"):
            assert torch.equal(s0[k], s1[k]), f"replicas diverged in {k}"
    sys.path.insert(0, ROOT)
    _inject_oracle()
    try:
        from enerf_amd.optim import FusedAdam
        m, ro, rd, target = _model_and_data(n_rays)
        opt = FusedAdam(m.get_params(1e-2), betas=(0.9, 0.99), eps=1e-15)
        for _ in range(2):
            _step(m, ro, rd, target, opt, None, scale=1.0 / (3.0 * n_rays))
        ref = m.state_dict()
        for k in ("encoder.embeddings", "sigma_net.0.weight", "sigma_net.1.weight", "color_net.0.weight",
                  "color_net.1.weight", "color_net.2.weight"):
            assert float((s0[k] - ref[k]).abs().max()) <= 1e-5 * float(ref[k].abs().max()) + 2e-6, k
    finally:
        import importlib
        import enerf_amd.raymarching as rm, enerf_amd.gridencoder as ge, enerf_amd.shencoder as sh
        importlib.reload(rm); importlib.reload(ge); importlib.reload(sh)
