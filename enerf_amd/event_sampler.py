"""Event-pair sampling on the device: the part of EventNeRFDataset that feeds the event step.

The reference prepares, per event batch, a pixel-grouped event list with Python dict / list loops
(nerf/provider.py:1147-1199) and, per training step, draws `batch_size_evs` event pairs with a Python loop over the
batch (provider.py:1367-1410: successor filter, random accumulation window, polarity sum) before handing pixel
coordinates and interpolated poses to get_event_rays.  At 4096 pairs per step that loop, not the GPU, bounds the step.
Here both are tensor programs that run where the tensors live (HIP device in production):

  build_event_tables   events (x, y, t, p) -> pixel-grouped list + successor tables + polarity prefix sums
  sample_event_pairs   one step's pairs: start / end indices, summed polarities, pixel coordinates
  event_pair_batch     ... + pose gather + get_event_rays = the `rays_evs_*` / `pols` entries of the data dict

Given the same random draws the results are those of the reference's loops (oracle/event_collate.py restates them;
tests/test_event_sampler.py).  The draws themselves come from a torch generator instead of numpy's global state.
"""
import torch

from .events import get_event_rays


def build_event_tables(events):
    """events [E, 4] float tensor, columns (x, y, t_ns, polarity), any order.

    provider.py:1152-1199: sort by time; group by pixel in order of each pixel's first event, events of a pixel in
    time order; drop pixels with a single event; flatten.  Returns a dict of tensors on events.device:
      events        [N, 4]  the grouped list
      num_at_xy     [P]     events per kept pixel (> 1)             (xy_numEvs_Idx[:, 0])
      first_at_xy   [P]     index of the pixel's first event        (xy_numEvs_Idx[:, 1])
      no_successor  [N] bool  last event of its pixel               (idx_no_successor as a mask)
      num_successor [N] int64 later events at the same pixel        (num_successor_evs)
      pol_cumsum    [N+1]   exclusive prefix sum of the polarities (sum over (a, b] = pol_cumsum[b+1] - pol_cumsum[a+1])
    """
    ev = events
    order_t = torch.argsort(ev[:, 2], stable=True)
    ev = ev[order_t]
    key = ev[:, 1].to(torch.int64) * (1 << 20) + ev[:, 0].to(torch.int64)          # pixel id (x, y < 2^20)
    uniq, inverse, counts = torch.unique(key, return_inverse=True, return_counts=True)
    E = ev.shape[0]
    # rank of every pixel by the position of its first event in the time-sorted stream (dict insertion order)
    first_pos = torch.full((uniq.shape[0],), E, dtype=torch.int64, device=ev.device)
    first_pos.scatter_reduce_(0, inverse, torch.arange(E, device=ev.device), reduce="amin")
    rank = torch.empty_like(first_pos)
    rank[torch.argsort(first_pos)] = torch.arange(uniq.shape[0], device=ev.device)
    keep = counts[inverse] > 1
    ev, grp = ev[keep], rank[inverse][keep]
    order_g = torch.argsort(grp, stable=True)                                     # stable: time order inside a pixel
    ev, grp = ev[order_g], grp[order_g]
    N = ev.shape[0]
    _, num_at_xy = torch.unique_consecutive(grp, return_counts=True)
    ends = torch.cumsum(num_at_xy, 0)                                             # one past each pixel's last event
    first_at_xy = ends - num_at_xy
    idx = torch.arange(N, device=ev.device)
    group_end = torch.repeat_interleave(ends, num_at_xy)
    num_successor = group_end - idx - 1
    pol_cumsum = torch.cat([ev.new_zeros(1, dtype=torch.float64), torch.cumsum(ev[:, 3].double(), 0)])
    return {"events": ev, "num_at_xy": num_at_xy, "first_at_xy": first_at_xy, "no_successor": num_successor == 0,
            "num_successor": num_successor, "pol_cumsum": pol_cumsum}


def sample_event_pairs(tables, batch_size, accumulate=True, acc_max_num_evs=0, generator=None, draws=None):
    """One step's event pairs.  provider.py:1367-1410.

    accumulate=True : a uniformly drawn event (moved one back if it is the last at its pixel) paired with a uniformly
                      drawn successor among its next min(num_successor, acc_max_num_evs + 1) events; polarity = sum
                      over the events in between (inclusive of the end).
    accumulate=False: per pixel a random event that has a successor, then `batch_size` of those pixels; the pair is
                      the event and its direct successor.
    `draws` (testing): dict of the uniform variates to use instead of the generator -- `start` int64 [M] and `u_end`
    float64 [M] in [0,1) for accumulate; `u_xy` [P] and `choice` int64 [M] otherwise.
    Returns start [M], end [M] (int64), pols [1, M] float32, xs, ys [1, M] float32."""
    ev = tables["events"]
    dev = ev.device
    N = ev.shape[0]
    M = batch_size
    draws = draws or {}
    if accumulate:
        start = draws["start"].to(dev) if "start" in draws else torch.randint(0, N, (M,), device=dev, generator=generator)
        start = start - tables["no_successor"][start].to(torch.int64)             # events without a successor: step back
        ns = tables["num_successor"][start]
        if acc_max_num_evs:
            ns = torch.clamp(ns, max=acc_max_num_evs + 1)
        u = draws["u_end"].to(dev) if "u_end" in draws else torch.rand(M, device=dev, generator=generator,
                                                                       dtype=torch.float64)
        end = start + 1 + torch.clamp((u * ns.double()).floor().to(torch.int64), max=ns - 1)
        cs = tables["pol_cumsum"]
        pols = (cs[end + 1] - cs[start + 1]).float()
    else:
        num, first = tables["num_at_xy"], tables["first_at_xy"]
        u = draws["u_xy"].to(dev) if "u_xy" in draws else torch.rand(num.shape[0], device=dev, generator=generator,
                                                                     dtype=torch.float64)
        # (np.random.rand(P) * num - 1).astype(int) + first: truncation toward zero, so -1 < v < 0 maps to 0
        per_xy = (u * num.double() - 1).trunc().to(torch.int64) + first
        P = per_xy.shape[0]
        if "choice" in draws:
            choice = draws["choice"].to(dev)
        elif M > P:
            choice = torch.randint(0, P, (M,), device=dev, generator=generator)
        else:
            choice = torch.randperm(P, device=dev, generator=generator)[:M]
        start = per_xy[choice]
        end = start + 1
        pols = ev[end, 3]
    xs = ev[start, 0].unsqueeze(0)
    ys = ev[start, 1].unsqueeze(0)
    return start, end, pols.unsqueeze(0), xs, ys


def event_pair_batch(tables, poses_evs, intrinsics, batch_size, accumulate=True, acc_max_num_evs=0, generator=None,
                     draws=None):
    """The event entries of EventNeRFDataset.collate's result (provider.py:1412-1441, pre-interpolated poses):
    poses_evs [N, 3, 4] per-event camera-to-world matrices aligned with tables["events"]."""
    start, end, pols, xs, ys = sample_event_pairs(tables, batch_size, accumulate, acc_max_num_evs, generator, draws)
    rays = get_event_rays(xs, ys, poses_evs[start].unsqueeze(0), poses_evs[end].unsqueeze(0), intrinsics)
    rays["pols"] = pols
    return rays


def event_pair_rays(tables, track, intrinsics, batch_size, acc_max_num_evs=0, generator=None, draws=None):
    """One step's event entries of collate (provider.py:1364-1441, accumulate_evs, poses computed online) as ONE launch
    on the device: pair selection + polarity sums + pose interpolation at both event times (PoseTrack) + ray generation
    (csrc/event_pairs.hip).  `tables` from build_event_tables on the device, `track` a PoseTrack on the same device.
    -> {"rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2"} [1,M,3], "pols" [1,M], "start" / "end" [M]."""
    from . import _lib as L
    ev = tables["events"]
    dev = ev.device
    if not ev.is_cuda:
        raise RuntimeError("event_pair_rays runs on the device; the host route is event_pair_batch")
    N, M = ev.shape[0], int(batch_size)
    draws = draws or {}
    start = draws["start"].to(dev) if "start" in draws else torch.randint(0, N, (M,), device=dev, generator=generator)
    u = draws["u_end"].to(dev) if "u_end" in draws else torch.rand(M, device=dev, generator=generator, dtype=torch.float64)
    if "_packed" not in tables:        # the layout the kernel reads, built once per event batch
        tables["_packed"] = (ev.float().contiguous(), tables["no_successor"].to(torch.uint8).contiguous(),
                             tables["num_successor"].to(torch.int64).contiguous(),
                             tables["pol_cumsum"].to(torch.float64).contiguous())
    evf, nos, nsucc, cs = tables["_packed"]
    f32 = dict(dtype=torch.float32, device=dev)
    o1, d1, o2, d2 = (torch.empty(M, 3, **f32) for _ in range(4))
    pols = torch.empty(M, **f32)
    s_out = torch.empty(M, dtype=torch.int64, device=dev)
    e_out = torch.empty(M, dtype=torch.int64, device=dev)
    outside = torch.zeros(1, dtype=torch.int32, device=dev)
    fx, fy, cx, cy = (float(v) for v in intrinsics)
    L.check(L.lib().enerf_event_pair_rays(
        evf.data_ptr(), nos.data_ptr(), nsucc.data_ptr(), cs.data_ptr(), N, start.to(torch.int64).contiguous().data_ptr(),
        u.to(torch.float64).contiguous().data_ptr(), M, int(acc_max_num_evs), track.knots.data_ptr(), track.rot.data_ptr(),
        track.rotvec.data_ptr(), track.tcoef.data_ptr(), track.K, fx, fy, cx, cy, o1.data_ptr(), d1.data_ptr(),
        o2.data_ptr(), d2.data_ptr(), pols.data_ptr(), s_out.data_ptr(), e_out.data_ptr(), outside.data_ptr(),
        L.stream_handle()), "event_pair_rays")
    return {"rays_evs_o1": o1[None], "rays_evs_d1": d1[None], "rays_evs_o2": o2[None], "rays_evs_d2": d2[None],
            "pols": pols[None], "start": s_out, "end": e_out, "outside_track": outside}
