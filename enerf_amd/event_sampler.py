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
import numpy as np
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


# ------------------------------------------------------------------------------------------------ negative events
def build_no_event_tables(events, H, W, start_time_us, end_time_us, chunk_len_ms=20.0, rectify_map=None, generator=None,
                          keep=None):
    """Where NOTHING happened (`--negative_event_sampling`, nerf/provider.py:1283-1345): the event batch's time span
    [start_time_us, end_time_us) is cut into N = int(duration_ms / chunk_len_ms) + 1 equal chunks; for every chunk the
    pixels without an event in it are listed and a random 1/N of them is kept (the reference's memory cap).

    events [n, >=3] rows (x, y, t_ns, ...) on any device (the lists are built where the events live: one scatter into an
    H*W mask and one nonzero per chunk instead of the host's linspace / fancy-index / np.random.choice);
    rectify_map [H, W, 2] (x, y) or None for the identity map (esim).  `keep(j, candidates)` overrides the random choice
    (tests: the reference's own draw).
    -> {"coords": [N] list of [n_j, 2] float32 (x, y), "start_time_us" / "end_time_us": [N] lists, "N_ev_chunks": N,
        "dt_us": chunk length}."""
    ev = events if isinstance(events, torch.Tensor) else torch.as_tensor(np.asarray(events))
    dev = ev.device
    dur_ms = end_time_us / 1e3 - start_time_us / 1e3
    if not end_time_us > start_time_us:
        raise ValueError("no-event tables: the batch must span a positive time")
    n_chunks = int(dur_ms / chunk_len_ms) + 1
    dt_us = 1e3 * dur_ms / n_chunks
    xs = ev[:, 0].to(torch.int64)
    ys = ev[:, 1].to(torch.int64)
    t_us = ev[:, 2].to(torch.float64) * 1e-3
    out = {"coords": [], "start_time_us": [], "end_time_us": [], "N_ev_chunks": n_chunks, "dt_us": dt_us}
    ts = float(start_time_us)
    for j in range(n_chunks):
        m = (t_us >= ts) & (t_us < ts + dt_us)
        hit = torch.zeros(H * W, dtype=torch.bool, device=dev)
        hit[(ys[m] * W + xs[m])] = True
        cand = torch.nonzero(~hit).squeeze(1)                        # linear pixel index, ascending (the reference's order)
        n_keep = int(cand.numel() / n_chunks)
        if keep is not None:
            sel = torch.as_tensor(keep(j, cand), device=dev, dtype=torch.int64)
        else:
            sel = cand[torch.randperm(cand.numel(), device=dev, generator=generator)[:n_keep]]
        py, px = sel // W, sel % W
        if rectify_map is not None:
            rm = rectify_map if isinstance(rectify_map, torch.Tensor) else torch.as_tensor(np.asarray(rectify_map))
            xy = rm.to(dev)[py, px].to(torch.float32)
        else:
            xy = torch.stack([px, py], dim=1).to(torch.float32)
        if xy.shape[0] == 0:
            xy = torch.zeros(1, 2, dtype=torch.float32, device=dev)       # the reference's one dummy pixel (:1342-1343)
        out["coords"].append(xy)
        out["start_time_us"].append(ts)
        out["end_time_us"].append(ts + dt_us)
        ts += dt_us
    return out


def no_event_rays(no_evs, track, intrinsics, batch_size_evs, generator=None, draws=None):
    """One step's `rays_no_evs_*` entries of collate (nerf/provider.py:1443-1476): N = batch_size_evs / 2 pixels of one
    random chunk (with replacement), two uniform times inside the chunk in ascending order, the camera at both times
    (PoseTrack = the reference's Slerp + cubic interp1d, on the device), rays through both poses.
    `draws` = {"chunk", "idx" [N], "u" [N, 2]} replaces the random draws (tests).
    -> {"rays_no_evs_o1", "rays_no_evs_d1", "rays_no_evs_o2", "rays_no_evs_d2"} [1, N, 3] (+ "chunk", "tss_us")."""
    draws = draws or {}
    n = int(batch_size_evs * 0.5)
    n_chunks = int(no_evs["N_ev_chunks"])
    dev = no_evs["coords"][0].device
    if "chunk" in draws:
        j = int(draws["chunk"])
    else:
        # (the draws happen where the generator lives: a CPU generator with the tables on the GPU is the usual case)
        j = int(torch.randint(0, n_chunks, (1,), generator=generator,
                              device=generator.device if generator is not None else "cpu"))
    coords = no_evs["coords"][j]
    if coords.shape[0] == 0:
        raise ValueError(f"no-event chunk {j} is empty")
    gdev = generator.device if generator is not None else dev
    idx = draws["idx"].to(dev) if "idx" in draws else torch.randint(0, coords.shape[0], (n,), device=gdev,
                                                                     generator=generator).to(dev)
    u = draws["u"].to(dev, torch.float64) if "u" in draws else torch.rand(n, 2, device=gdev, generator=generator,
                                                                           dtype=torch.float64).to(dev)
    t0, t1 = no_evs["start_time_us"][j], no_evs["end_time_us"][j]
    tss = torch.sort(t0 + (t1 - t0) * u, dim=1).values                # [n, 2] microseconds, ascending per pixel
    xs = coords[idx, 0].unsqueeze(0)
    ys = coords[idx, 1].unsqueeze(0)
    p1 = track.poses_at(tss[:, 0] * 1000).to(dev)                     # the track is in nanoseconds
    p2 = track.poses_at(tss[:, 1] * 1000).to(dev)
    r = get_event_rays(xs, ys, p1.unsqueeze(0), p2.unsqueeze(0), intrinsics)
    return {"rays_no_evs_o1": r["rays_evs_o1"], "rays_no_evs_d1": r["rays_evs_d1"],
            "rays_no_evs_o2": r["rays_evs_o2"], "rays_no_evs_d2": r["rays_evs_d2"], "chunk": j, "tss_us": tss}
