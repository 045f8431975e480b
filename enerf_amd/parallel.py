"""Ray-sharded data parallelism for the train step: one process per GPU, every rank holds a full
replica (hash table + MLPs + occupancy state), renders its own slice of the step's rays, and the gradients are
averaged with ONE collective exchange per step over RCCL/xGMI (`torch.distributed`, backend "nccl" on ROCm; "gloo"
in the CPU tests).  There is no collective in the forward/backward data path itself.

Bucketing: the hash-table gradient (52 MB fp32 at bound 3) is reduced in place as its own bucket; all MLP gradients
(37 KB) are flattened into a second bucket so that two collectives are issued per step regardless of layer count.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank); a single-process run needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver stack
        # a rank that never arrives (or a collective that never completes) must end the job with an error on every rank,
        # not hold the node: RCCL's watchdog tears the communicator down when a collective times out.  The timeout itself
        # is torch's default (10 min nccl / 30 min gloo: long single-rank phases -- full-resolution evaluation, an export
        # on rank 0, a first build -- must not kill a training job) unless ENERF_DIST_TIMEOUT_S says otherwise; bench.py,
        # whose phases are seconds long, sets it to 300 for its own run.
        import datetime
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        kw = {}
        if os.environ.get("ENERF_DIST_TIMEOUT_S"):
            kw["timeout"] = datetime.timedelta(seconds=int(os.environ["ENERF_DIST_TIMEOUT_S"]))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def shard_range(n, rank, world):
    """Contiguous slice [lo, hi) of n rays owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradAverager:
    """Averages gradients of `params` across ranks with two buckets (big tensors in place, small ones flattened)."""

    def __init__(self, params, big_threshold=1 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.big = [p for p in self.params if p.numel() >= big_threshold]
        self.small = [p for p in self.params if p.numel() < big_threshold]
        self._flat = None
        self._works = []
        self._avg_in_op = False

    def start(self):
        """Launch the two collectives (async).  Work queued on the current stream afterwards overlaps with them."""
        self._works = []
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        # RCCL averages in the collective; gloo (CPU tests) only sums
        self._avg_in_op = dist.get_backend() == "nccl"
        op = dist.ReduceOp.AVG if self._avg_in_op else dist.ReduceOp.SUM
        for p in self.big:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            self._works.append(dist.all_reduce(p.grad, op=op, async_op=True))
        small = self.small
        if small:
            n = sum(p.numel() for p in small)
            if self._flat is None or self._flat.numel() != n or self._flat.device != small[0].device:
                self._flat = torch.empty(n, dtype=torch.float32, device=small[0].device)
            off = 0
            for p in small:
                k = p.numel()
                if p.grad is None:
                    self._flat[off:off + k].zero_()
                else:
                    self._flat[off:off + k].copy_(p.grad.reshape(-1))
                off += k
            self._works.append(dist.all_reduce(self._flat, op=op, async_op=True))

    def finish(self):
        """Wait for the collectives and leave the averaged gradients in p.grad."""
        if not self._works:
            return
        for w in self._works:
            w.wait()
        self._works = []
        inv = 1.0 / dist.get_world_size()
        if not self._avg_in_op:
            for p in self.big:
                p.grad.mul_(inv)
        if self.small:
            off = 0
            for p in self.small:
                k = p.numel()
                if p.grad is None:
                    p.grad = torch.empty_like(p)
                p.grad.copy_(self._flat[off:off + k].view_as(p))
                if not self._avg_in_op:
                    p.grad.mul_(inv)
                off += k

    def __call__(self):
        self.start()
        self.finish()


@torch.no_grad()
def render_sharded(model, rays_o, rays_d, **render_kw):
    """Inference at N > 1: rank r renders the contiguous slice shard_range(N, r, world) of the rays
    [1,N,3] and the tiles are all-gathered (the reference's evaluation loop does the same with its predictions,
    nerf/utils.py:1069-1071): every rank returns the full {"image": [1,N,3], "depth": [1,N]}.  One collective of
    16 B per ray (4.9 MB for a 640x480 frame); slices are padded to equal length so that a single
    all_gather_into_tensor serves any N.  A single process (no process group) just renders."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model.render(rays_o, rays_d, staged=False, **render_kw)
    world, rank = dist.get_world_size(), dist.get_rank()
    N = rays_o.shape[1]
    lo, hi = shard_range(N, rank, world)
    out = model.render(rays_o[:, lo:hi].contiguous(), rays_d[:, lo:hi].contiguous(), staged=False, **render_kw)
    width = -(-N // world)
    tile = torch.zeros(width, 4, dtype=torch.float32, device=rays_o.device)          # (r, g, b, depth) per ray
    tile[:hi - lo, :3] = out["image"].reshape(-1, 3)
    tile[:hi - lo, 3] = out["depth"].reshape(-1)
    full = torch.empty(world * width, 4, dtype=torch.float32, device=rays_o.device)
    dist.all_gather_into_tensor(full, tile)
    rows = torch.cat([full[r * width:r * width + (shard_range(N, r, world)[1] - shard_range(N, r, world)[0])]
                      for r in range(world)]) if world * width != N else full
    return {"image": rows[:, :3].reshape(1, N, 3), "depth": rows[:, 3].reshape(1, N)}


def broadcast_state(model, src=0):
    """Make replicas identical (parameters and buffers, incl. the occupancy grid / bitfield)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src)
