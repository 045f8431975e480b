"""CPU restatement (plain Python loops / numpy, TEST INFRASTRUCTURE ONLY) of the reference's event-pair preparation:

  group_events       nerf/provider.py:1147-1199   dict of pixel -> event list in first-occurrence order, > 1 event per
                                                  pixel, flattened; xy_numEvs_Idx, idx_no_successor, num_successor_evs
  collate_pairs      nerf/provider.py:1367-1410   the per-step Python loop (successor filter, random window end,
                                                  polarity sum) -- with the uniform draws passed in instead of taken
                                                  from numpy's global generator

  no_event_tables    nerf/provider.py:1283-1351   --negative_event_sampling: per 20 ms chunk of an event batch, the pixels
                                                  without an event, subsampled to 1 / N_chunks of them
  no_event_rays      nerf/provider.py:1443-1476   the per-step no-event entries of collate: pixels of one chunk, two
                                                  sorted uniform times, scipy Slerp / cubic interp1d poses, get_event_rays
                                                  (draws passed in)

Parity status: pinned (round 4) by the reference's OWN EventNeRFDataset, constructed and collated with only its file
reading stubbed and numpy's draws recorded (oracle/make_golden.py gold_collate -> tests/golden/ref_collate.npz,
tests/test_collate_vs_reference.py): the product's tables equal the constructor's grouping loop entry for entry, its
no-event tables the loader's, its CPU and device routes reproduce collate -- ends and polarity sums exactly, rays to 1e-5.
These functions remain the line-by-line transcription the product is ALSO compared with (tests/test_event_sampler.py, with
an independent brute-force definition beside them).
"""
import numpy as np


def group_events(events):
    events_in = np.asarray(events, dtype=np.float32)
    events_in = np.asarray(sorted(events_in, key=lambda x: x[2]))                 # provider.py:1152
    evs_dict_xy = {}
    for ev in events_in:                                                          # :1156-1161
        key_xy = (ev[0], ev[1])
        if key_xy in evs_dict_xy:
            evs_dict_xy[key_xy].append(ev.tolist())
        else:
            evs_dict_xy[key_xy] = [ev.tolist()]
    evs_dict_xy = dict((k, v) for k, v in evs_dict_xy.items() if len(v) > 1)       # :1163
    xys = list(evs_dict_xy.keys())
    num_evs_at_xy = np.asarray([len(evs_dict_xy[xy]) for xy in xys])              # :1168
    xy_numEvs_Idx = np.concatenate((num_evs_at_xy[:, None],
                                    np.append(0, np.cumsum(num_evs_at_xy)[:-1])[:, None]), axis=1)   # :1171
    cum = np.cumsum(num_evs_at_xy)
    num_evs = int(cum[-1])
    idx_no_successor = cum - 1                                                    # :1178
    num_successor_evs = np.zeros(num_evs).astype(np.int64)                        # :1181-1186
    j = 0
    for i in range(num_evs):
        if i >= cum[j]:
            j += 1
        num_successor_evs[i] = cum[j] - i - 1
    flat = []
    for xy in xys:                                                                # :1190-1197
        for ev in evs_dict_xy[xy]:
            flat.append(ev)
    return {"events": np.asarray(flat, dtype=np.float32), "xy_numEvs_Idx": xy_numEvs_Idx,
            "idx_no_successor": idx_no_successor, "num_successor_evs": num_successor_evs}


def collate_pairs(g, eidx, u_end, acc_max_num_evs=0):
    """provider.py:1369-1398 with eidx (the randint draw) and u_end in [0,1) (mapped onto randint's range) given."""
    no_succ = set(int(v) for v in g["idx_no_successor"])
    eidx = np.asarray([e - 1 if int(e) in no_succ else e for e in eidx])          # :1371
    eidx_end, sum_pols = [], []
    for k, ev_id_start in enumerate(eidx):
        num_successors = g["num_successor_evs"][ev_id_start]
        if acc_max_num_evs:
            num_successors = np.minimum(num_successors, acc_max_num_evs + 1)
        # np.random.randint(ev_id_start + 1, ev_id_start + 1 + num_successors)
        ev_id_end = ev_id_start + 1 + min(int(np.floor(u_end[k] * num_successors)), int(num_successors) - 1)
        ps = g["events"][(ev_id_start + 1):(ev_id_end + 1), 3]
        sum_pols.append(ps.sum())
        eidx_end.append(ev_id_end)
    xs = g["events"][eidx, 0]
    ys = g["events"][eidx, 1]
    return eidx, np.asarray(eidx_end), np.asarray(sum_pols, dtype=np.float32), xs, ys


def collate_single(g, u_xy, choice):
    """provider.py:1400-1405 (accumulate_evs off) with the per-pixel uniforms and the np.random.choice result given."""
    num_evs_xy = g["xy_numEvs_Idx"][:, 0]
    eidx = (u_xy * num_evs_xy - 1).astype(int) + g["xy_numEvs_Idx"][:, 1]
    eidx = eidx[choice]
    pols = g["events"][eidx + 1, 3]
    return eidx, eidx + 1, pols, g["events"][eidx, 0], g["events"][eidx, 1]


# ---- event time index (utils/event_utils.py) -------------------------------------------------------------------------
def ms_to_idx_loop(t, unit_per_ms, ms_start=0):
    """utils/event_utils.py:389-408 (compute_ms_to_idx) by its defining property, one millisecond at a time:
    idx[ms] = first i with t[i] >= ms * unit_per_ms (len(t) if none); ms runs to floor(max(t)) / unit_per_ms."""
    t = np.asarray(t)
    ms_end = int(np.floor(t.max()) / unit_per_ms)
    out = []
    for ms in range(ms_start, ms_end + 1):
        i = 0
        while i < len(t) and t[i] < ms * unit_per_ms:
            i += 1
        out.append(i)
    return np.asarray(out, dtype=np.int64)


def slicer_window(t_us, ms_to_idx, t_start_us, t_end_us, t_offset=0):
    """EventSlicer.get_events' index arithmetic (utils/event_utils.py:256-300, :303-322, :324-377, :379-383): the
    conservative millisecond window, then the two linear scans.  Returns (first, one past last) or None."""
    assert t_start_us < t_end_us
    t_start_us -= t_offset
    t_end_us -= t_offset
    w0 = max(int(np.floor(t_start_us / 1000)), 0)
    w1 = int(np.ceil(t_end_us / 1000))
    if w0 >= len(ms_to_idx) or w1 >= len(ms_to_idx):
        return None
    a, b = int(ms_to_idx[w0]), int(ms_to_idx[w1])
    arr = np.asarray(t_us[a:b])
    if arr.size == 0 or arr[-1] < t_start_us:
        return a + arr.size, a + arr.size
    i0 = 0
    while arr[i0] < t_start_us:
        i0 += 1
    i1 = arr.size
    for k in range(arr.size - 1, -1, -1):
        if arr[k] >= t_end_us:
            i1 = k
        else:
            break
    return a + i0, a + i1


def no_event_tables(evs_batch_ns, coords, H_ev, W_ev, start_time_us, end_time_us, rectify_map, choice, chunk_len_ms=20):
    """provider.py:1283-1351 for ONE event batch.  evs_batch_ns [n, >=3] (x, y, t_ns, ...), coords [n, 2] the pixel
    coordinates used for the lookup (:1305-1306), rectify_map [H, W, 2].  `choice(j, idxs, size)` stands for
    np.random.choice(idxs, size=size, replace=False) (:1327)."""
    out = {"coords": [], "tss_bds": {"N_ev_chunks": [], "start_time_us": [], "end_time_us": [], "dt_us": []}}
    assert end_time_us > start_time_us
    dur_ms = end_time_us / 1e3 - start_time_us / 1e3                                # :1299
    N_ev_chunks = int(dur_ms / chunk_len_ms) + 1                                    # :1302
    dt_us = 1e3 * dur_ms / N_ev_chunks
    xsall = coords[:, 0].astype(np.uint32)
    ysall = coords[:, 1].astype(np.uint32)
    ts_iter = start_time_us
    for j in range(N_ev_chunks):
        ts_mask = (evs_batch_ns[:, 2] * 1e-3 >= ts_iter) & (evs_batch_ns[:, 2] * 1e-3 < (ts_iter + dt_us))   # :1309
        xstmp, ystmp = xsall[ts_mask], ysall[ts_mask]
        idxs_no_evs = np.linspace(1, H_ev * W_ev, H_ev * W_ev).astype(np.uint32)    # :1316
        idxs_evs = ystmp * W_ev + xstmp
        idxs_no_evs[idxs_evs] = 0
        N_noevs = (idxs_no_evs > 0).sum()
        idxs_no_evs = idxs_no_evs[idxs_no_evs > 0]
        idxs_no_evs = choice(j, idxs_no_evs, int(N_noevs / N_ev_chunks))            # :1327
        N_noevs = (idxs_no_evs > 0).sum()
        ys, xs = (idxs_no_evs - 1) // W_ev, (idxs_no_evs - 1) % W_ev                # :1331
        rect = rectify_map[ys, xs]
        no_evs_batch = np.zeros((N_noevs, 2))
        no_evs_batch[:, 0] = rect[:, 0]
        no_evs_batch[:, 1] = rect[:, 1]
        if len(no_evs_batch) == 0:
            no_evs_batch = np.zeros((1, 2))                                          # :1342
        out["coords"].append(no_evs_batch.astype(np.float32))                        # (float32 cast: :1231-1232)
        out["tss_bds"]["start_time_us"].append(ts_iter)
        out["tss_bds"]["end_time_us"].append(ts_iter + dt_us)
        ts_iter += dt_us
    out["tss_bds"]["N_ev_chunks"].append(N_ev_chunks)
    out["tss_bds"]["dt_us"].append(dt_us)
    return out


def no_event_rays(no_evs, rot_interpolator, trans_interpolator, get_event_rays, intrinsics, batch_size_evs, chunk_j, neidx,
                  u01):
    """provider.py:1443-1476 with the three random draws (chunk, pixel indices, the [N, 2] uniforms) passed in.
    rot_interpolator / trans_interpolator: scipy Slerp / interp1d(kind="cubic") over the pose track in nanoseconds."""
    import torch
    N_noevs = int(batch_size_evs * 0.5)
    assert len(neidx) == N_noevs and u01.shape == (N_noevs, 2)
    c = torch.from_numpy(np.asarray(no_evs["coords"][chunk_j], dtype=np.float32))
    xsno = c[neidx, 0].unsqueeze(0)
    ysno = c[neidx, 1].unsqueeze(0)
    t0, t1 = no_evs["tss_bds"]["start_time_us"][chunk_j], no_evs["tss_bds"]["end_time_us"][chunk_j]
    tss = np.sort(t0 + (t1 - t0) * u01, axis=1)                                      # :1457-1458
    poses = []
    for col in (0, 1):
        ts_ns = tss[:, col] * 1000                                                   # :1461 / :1467
        rots = rot_interpolator(ts_ns).as_matrix()
        trans = trans_interpolator(ts_ns)
        hom = np.concatenate([rots, trans[:, :, None]], axis=2)                      # get_hom_trafos(...)[:, :3, :]
        poses.append(torch.from_numpy(hom.astype(np.float32)))
    return get_event_rays(xsno, ysno, poses[0].unsqueeze(0), poses[1].unsqueeze(0), intrinsics), tss
