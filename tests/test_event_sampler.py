"""Event-pair sampling (enerf_amd/event_sampler.py) against the loop restatement of the reference
(oracle/event_collate.py) and against a brute-force definition, on CPU; on the GPU against the CPU result."""
import numpy as np
import pytest
import torch

from oracle import event_collate as EC


def _events(n, w, h, seed):
    rng = np.random.default_rng(seed)
    xs = rng.integers(0, w, n)
    ys = rng.integers(0, h, n)
    ts = rng.permutation(n * 3)[:n].astype(np.float64) * 1000.0           # distinct timestamps, shuffled order
    ps = rng.choice([-1.0, 1.0], n)
    return np.stack([xs, ys, ts, ps], 1).astype(np.float32)


@pytest.mark.parametrize("n,w,h", [(400, 6, 5), (5000, 40, 30), (64, 64, 64)])
def test_tables_match_reference_loops(n, w, h):
    from enerf_amd.event_sampler import build_event_tables
    ev = _events(n, w, h, n)
    if n == 64:
        ev[:8, :2] = ev[0, :2]                                             # make sure some pixel has > 1 event
    g = EC.group_events(ev)
    t = build_event_tables(torch.from_numpy(ev))
    assert np.array_equal(t["events"].numpy(), g["events"])
    assert np.array_equal(t["num_at_xy"].numpy(), g["xy_numEvs_Idx"][:, 0])
    assert np.array_equal(t["first_at_xy"].numpy(), g["xy_numEvs_Idx"][:, 1])
    assert np.array_equal(t["no_successor"].nonzero().flatten().numpy(), g["idx_no_successor"])
    assert np.array_equal(t["num_successor"].numpy(), g["num_successor_evs"])
    # brute force: successors = later events at the same pixel in the flattened list
    e = g["events"]
    for i in range(0, len(e), max(1, len(e) // 50)):
        same = (e[i + 1:, 0] == e[i, 0]) & (e[i + 1:, 1] == e[i, 1])
        assert int(same.sum()) == int(t["num_successor"][i])
        assert np.all(e[i + 1:][same][:, 2] > e[i, 2])


@pytest.mark.parametrize("acc_max", [0, 3])
def test_accumulated_pairs_match_reference_loop(acc_max):
    from enerf_amd.event_sampler import build_event_tables, sample_event_pairs
    ev = _events(6000, 32, 24, 7)
    g = EC.group_events(ev)
    t = build_event_tables(torch.from_numpy(ev))
    N, M = len(g["events"]), 2048
    rng = np.random.default_rng(8)
    eidx = rng.integers(0, N, M)
    u = rng.random(M)
    rs, re_, rp, rx, ry = EC.collate_pairs(g, eidx, u, acc_max)
    s, e, p, xs, ys = sample_event_pairs(t, M, True, acc_max, draws={"start": torch.from_numpy(eidx),
                                                                      "u_end": torch.from_numpy(u)})
    assert np.array_equal(s.numpy(), rs) and np.array_equal(e.numpy(), re_)
    assert np.array_equal(p[0].numpy(), rp)                               # sums of +-1: exact
    assert np.array_equal(xs[0].numpy(), rx) and np.array_equal(ys[0].numpy(), ry)
    ee = g["events"]
    assert np.all(ee[rs, 0] == ee[re_, 0]) and np.all(ee[rs, 1] == ee[re_, 1]) and np.all(ee[rs, 2] < ee[re_, 2])
    if acc_max:
        assert int((e - s).max()) <= acc_max + 1


def test_single_successor_pairs_match_reference():
    from enerf_amd.event_sampler import build_event_tables, sample_event_pairs
    ev = _events(3000, 20, 20, 9)
    g = EC.group_events(ev)
    t = build_event_tables(torch.from_numpy(ev))
    P = len(g["xy_numEvs_Idx"])
    rng = np.random.default_rng(10)
    u = rng.random(P)
    choice = rng.integers(0, P, 512)
    rs, re_, rp, rx, ry = EC.collate_single(g, u, choice)
    s, e, p, xs, ys = sample_event_pairs(t, 512, False, draws={"u_xy": torch.from_numpy(u),
                                                                "choice": torch.from_numpy(choice)})
    assert np.array_equal(s.numpy(), rs) and np.array_equal(e.numpy(), re_) and np.array_equal(p[0].numpy(), rp)
    assert np.array_equal(xs[0].numpy(), rx) and np.array_equal(ys[0].numpy(), ry)


def test_batch_has_the_reference_data_dict_entries():
    from enerf_amd.event_sampler import build_event_tables, event_pair_batch
    ev = _events(2000, 16, 12, 11)
    t = build_event_tables(torch.from_numpy(ev))
    N = t["events"].shape[0]
    g = torch.Generator().manual_seed(3)
    poses = torch.eye(4)[:3].repeat(N, 1, 1) + 0.01 * torch.randn(N, 3, 4, generator=g)
    out = event_pair_batch(t, poses, (10.0, 10.0, 8.0, 6.0), 256, True, 4, generator=g)
    assert set(out) == {"rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2", "pols"}
    assert out["rays_evs_o1"].shape == (1, 256, 3) and out["rays_evs_d2"].shape == (1, 256, 3)
    assert out["pols"].shape == (1, 256) and float(out["pols"].abs().max()) <= 5
    assert torch.allclose(out["rays_evs_d1"].norm(dim=-1), (out["rays_evs_d1"] * 0 + 1).sum(-1) / 3, atol=0.2)


@pytest.mark.gpu
def test_device_tables_and_pairs_match_cpu():
    from enerf_amd.event_sampler import build_event_tables, sample_event_pairs
    ev = torch.from_numpy(_events(200000, 346, 260, 12))
    tc = build_event_tables(ev)
    tg = build_event_tables(ev.cuda())
    for k in tc:
        assert torch.equal(tc[k], tg[k].cpu()), k
    N = tc["events"].shape[0]
    g = torch.Generator().manual_seed(5)
    draws = {"start": torch.randint(0, N, (4096,), generator=g), "u_end": torch.rand(4096, generator=g, dtype=torch.float64)}
    a = sample_event_pairs(tc, 4096, True, 8, draws=draws)
    b = sample_event_pairs(tg, 4096, True, 8, draws=draws)
    for x, y in zip(a, b):
        assert torch.equal(x, y.cpu())
    # free-running generator on the device: pairs are valid (same pixel, later time, window bound)
    s, e, p, xs, ys = sample_event_pairs(tg, 4096, True, 8, generator=torch.Generator(device="cuda").manual_seed(1))
    evg = tg["events"]
    assert bool((evg[s, 0] == evg[e, 0]).all()) and bool((evg[s, 1] == evg[e, 1]).all())
    assert bool((evg[s, 2] < evg[e, 2]).all()) and int((e - s).max()) <= 9 and int((e - s).min()) >= 1


def _track(K, t_lo, t_hi, seed):
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(seed)
    gaps = rng.uniform(0.5, 2.0, K - 1)
    t = t_lo - 1.0 + np.concatenate([[0.0], np.cumsum(gaps)]) * ((t_hi - t_lo + 2.0) / gaps.sum())
    R = Rotation.from_rotvec(np.cumsum(rng.normal(size=(K, 3)) * 0.04, 0))
    p = np.cumsum(rng.normal(size=(K, 3)) * 0.02, 0) + np.array([1.5, 0.3, 0.0])
    return t, R, p


def test_pose_track_equals_scipy_slerp_and_cubic_interp1d():
    """enerf_amd/pose_interp.PoseTrack (per-segment rotation vectors + cubic coefficients, evaluated as a tensor
    program) against what the reference calls per step: scipy's Slerp and interp1d(kind="cubic")
    (nerf/provider.py:1142-1143) -- the reference's own dependency is the oracle here."""
    from scipy.interpolate import interp1d
    from scipy.spatial.transform import Slerp
    from enerf_amd.pose_interp import PoseTrack
    t, R, p = _track(60, 1.0e9, 1.2e9, 3)
    tr = PoseTrack(t, R.as_matrix(), p)
    rng = np.random.default_rng(4)
    q = rng.uniform(t[0], t[-1], 5000)
    q[:4] = [t[0], t[-1], t[17], np.nextafter(t[18], 0)]                   # knots and a hair before a knot
    got = tr.poses_at(torch.from_numpy(q)).double().numpy()
    ref_R = Slerp(t, R)(q).as_matrix()
    ref_p = interp1d(x=t, y=p, axis=0, kind="cubic", bounds_error=True)(q)
    assert np.abs(got[:, :, :3] - ref_R).max() < 2e-7 and np.abs(got[:, :, 3] - ref_p).max() < 2e-7
    with pytest.raises(ValueError):
        tr.poses_at(torch.tensor([t[-1] + 1.0]))


@pytest.mark.gpu
@pytest.mark.parametrize("acc_max", [0, 6])
def test_event_pair_rays_kernel_vs_reference_loop_scipy_and_get_event_rays(acc_max):
    """csrc/event_pairs.hip (one launch: pair selection, polarity sums, pose interpolation, rays) at 4096 pairs of a
    200 k-event batch, against the host route the reference takes per step: the collate loop (oracle/event_collate.py),
    scipy's Slerp + cubic interp1d at the event times, get_event_rays."""
    from scipy.interpolate import interp1d
    from scipy.spatial.transform import Slerp
    from enerf_amd.event_sampler import build_event_tables, event_pair_rays
    from enerf_amd.events import get_event_rays
    from enerf_amd.pose_interp import PoseTrack
    ev = _events(200000, 346, 260, 12)
    g = EC.group_events(ev)
    tables = build_event_tables(torch.from_numpy(ev).cuda())
    assert np.array_equal(tables["events"].cpu().numpy(), g["events"])
    N, M = len(g["events"]), 4096
    t, R, p = _track(500, float(ev[:, 2].min()), float(ev[:, 2].max()), 5)
    track = PoseTrack(t, R.as_matrix(), p, device="cuda")
    rng = np.random.default_rng(6)
    eidx, u = rng.integers(0, N, M), rng.random(M)
    intr = (320.0, 320.0, 173.0, 130.0)
    out = event_pair_rays(tables, track, intr, M, acc_max, draws={"start": torch.from_numpy(eidx),
                                                                   "u_end": torch.from_numpy(u)})
    torch.cuda.synchronize()
    rs, re_, rp, rx, ry = EC.collate_pairs(g, eidx, u, acc_max)
    assert np.array_equal(out["start"].cpu().numpy(), rs) and np.array_equal(out["end"].cpu().numpy(), re_)
    assert np.array_equal(out["pols"][0].cpu().numpy(), rp) and int(out["outside_track"]) == 0

    def host_pose(idx):                                                    # provider.py:1411-1415
        ts = g["events"][idx, 2].astype(np.float64)
        rots = Slerp(t, R)(ts).as_matrix()
        trans = interp1d(x=t, y=p, axis=0, kind="cubic", bounds_error=True)(ts)
        return torch.Tensor(np.concatenate([rots, trans[:, :, None]], -1)).unsqueeze(0)

    ref = get_event_rays(torch.from_numpy(rx)[None], torch.from_numpy(ry)[None], host_pose(rs), host_pose(re_), intr)
    for k in ("rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k].numpy(), rtol=0, atol=2e-6, err_msg=k)
    # times outside the track are counted, as interp1d(bounds_error=True) would have raised
    short = PoseTrack(t[:250], R[:250].as_matrix(), p[:250], device="cuda")
    assert int(event_pair_rays(tables, short, intr, M, acc_max)["outside_track"]) > 0


# ---- event time index / esim loader (enerf_amd/event_window.py) ------------------------------------------------------
def test_ms_to_idx_known_answer_of_the_reference_docstring():
    """The worked example in utils/event_utils.py:241-249."""
    from enerf_amd.event_window import EventTimeIndex
    from oracle.event_collate import ms_to_idx_loop
    t = [0, 500, 2100, 5000, 5000, 7100, 7200, 7200, 8100, 9000]
    want = [0, 2, 2, 3, 3, 3, 5, 5, 8, 9]
    assert ms_to_idx_loop(t, 1000).tolist() == want
    assert EventTimeIndex(torch.tensor(t, dtype=torch.int64)).ms_to_idx.tolist() == want


def _window_cases(device):
    from enerf_amd.event_window import EventTimeIndex
    from oracle.event_collate import ms_to_idx_loop, slicer_window
    rng = np.random.default_rng(5)
    t = np.sort(rng.integers(0, 60_000, 4000)).astype(np.int64)          # microseconds, many ties
    t[:3] = 0
    index = EventTimeIndex(torch.from_numpy(t).to(device), t_offset=17)
    ref_idx = ms_to_idx_loop(t, 1000)
    assert index.ms_to_idx.cpu().numpy().tolist() == ref_idx.tolist()
    for _ in range(300):
        a = int(rng.integers(-2000, 62_000))
        b = a + int(rng.integers(1, 9000))
        want = slicer_window(t, ref_idx, a + 17, b + 17, t_offset=17)
        got = index.window(a + 17, b + 17)
        assert got == want, (a, b, got, want)
        if want is not None:                                           # the defining property, by brute force
            lo, hi = want
            inside = (t >= a) & (t < b)
            # the conservative millisecond window may cut the range at its far end only when ms_to_idx stops short
            assert inside[lo:hi].all() and not inside[:lo].any()
    assert index.t_final == int(t[-1]) + 17


def test_event_window_matches_the_slicer_restatement():
    _window_cases("cpu")


@pytest.mark.gpu
def test_event_window_on_the_device():
    _window_cases("cuda")


def test_esim_loader_batches_and_polarities(tmp_path):
    from enerf_amd.event_window import load_esim_event_batches
    rng = np.random.default_rng(2)
    files = []
    for k in range(5):
        n = 20 + k
        ev = np.stack([rng.integers(0, 64, n), rng.integers(0, 48, n), np.sort(rng.integers(0, 10**6, n)) + k * 10**6,
                       rng.integers(0, 2, n), np.zeros(n)], axis=1).astype(np.float64)
        np.save(tmp_path / f"{k:04d}.npy", ev)
        files.append(ev)
    (tmp_path / "notes.txt").write_text("ignored")
    out = load_esim_event_batches(str(tmp_path), [0, 2, 3], hwf=(48, 64, 50.0))
    assert [b.shape for b in out] == [(41, 4), (22, 4), (23, 4)]
    assert np.array_equal(out[0][:, :3], np.concatenate([files[0], files[1]])[:, :3])
    assert np.array_equal(out[0][:, 3], 2 * np.concatenate([files[0], files[1]])[:, 3] - 1)      # {0,1} -> {-1,+1}
    assert np.array_equal(out[2][:, :3], files[3][:, :3])
    one = load_esim_event_batches(str(tmp_path), [4], hwf=(48, 64, 50.0))
    assert len(one) == 1 and one[0].shape == (24, 4)
    with pytest.raises(AssertionError):
        load_esim_event_batches(str(tmp_path), [0, 2], hwf=(32, 64, 50.0))                       # y up to 47 > sensor
    signed = files[0].copy()
    signed[:, 3] = 2 * signed[:, 3] - 1
    signed[0, 3] = -1
    np.save(tmp_path / "0000.npy", signed)
    assert np.array_equal(load_esim_event_batches(str(tmp_path), [0], hwf=(48, 64, 1.0))[0][:, 3], signed[:, 3])


# ------------------------------------------------------------------------------------------------ negative events
def _no_event_case(seed, W=40, H=30, n=6000, dur_ms=57.0):
    rng = np.random.default_rng(seed)
    t0_us = 1.0e6
    ts_ns = np.sort(rng.uniform(t0_us * 1e3, (t0_us + dur_ms * 1e3) * 1e3, n))
    ev = np.stack([rng.integers(0, W, n), rng.integers(0, H, n), ts_ns, rng.choice([-1.0, 1.0], n)], 1)
    rect = np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=2).astype(np.float64) + 0.25      # a "rectified" map
    return ev, rect, W, H, t0_us, t0_us + dur_ms * 1e3


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_no_event_tables_match_the_reference_loop(device):
    """build_no_event_tables (one scatter + one nonzero per chunk, where the events live) against the restated loop of
    nerf/provider.py:1283-1351 with the same subsampling draw; float64 event times (ns) as the loader holds them."""
    from enerf_amd.event_sampler import build_no_event_tables
    ev, rect, W, H, t0, t1 = _no_event_case(3)
    picks = {}

    def choice(j, idxs, size):
        picks[j] = np.random.default_rng(100 + j).choice(idxs, size=size, replace=False)
        return picks[j]

    ref = EC.no_event_tables(ev, ev[:, :2], H, W, t0, t1, rect, choice)
    assert ref["tss_bds"]["N_ev_chunks"][0] == 3                       # 57 ms -> three 19 ms chunks
    got = build_no_event_tables(torch.from_numpy(ev).to(device), H, W, t0, t1, rectify_map=rect,
                                keep=lambda j, cand: torch.from_numpy(picks[j].astype(np.int64) - 1))
    assert got["N_ev_chunks"] == 3 and abs(got["dt_us"] - ref["tss_bds"]["dt_us"][0]) < 1e-9
    for j in range(3):
        assert np.array_equal(got["coords"][j].cpu().numpy(), ref["coords"][j])
        assert got["start_time_us"][j] == ref["tss_bds"]["start_time_us"][j]
        assert got["end_time_us"][j] == ref["tss_bds"]["end_time_us"][j]
    # the candidates themselves (before subsampling) are exactly the pixels without an event in the chunk, ascending
    seen = []
    build_no_event_tables(torch.from_numpy(ev).to(device), H, W, t0, t1,
                          keep=lambda j, cand: (seen.append(cand.cpu().numpy()), cand[:5])[1])
    for j in range(3):
        lo, hi = ref["tss_bds"]["start_time_us"][j], ref["tss_bds"]["end_time_us"][j]
        m = (ev[:, 2] * 1e-3 >= lo) & (ev[:, 2] * 1e-3 < hi)
        hit = np.zeros(H * W, bool); hit[(ev[m, 1] * W + ev[m, 0]).astype(np.int64)] = True
        assert np.array_equal(seen[j], np.nonzero(~hit)[0])
    # its own random choice: the right number, no duplicates, all event-free
    own = build_no_event_tables(torch.from_numpy(ev).to(device), H, W, t0, t1)
    for j in range(3):
        c = own["coords"][j].cpu().numpy().astype(np.int64)
        lin = c[:, 1] * W + c[:, 0]
        assert len(lin) == int(len(seen[j]) / 3) and len(np.unique(lin)) == len(lin) and np.isin(lin, seen[j]).all()


def test_no_event_tables_degenerate_chunk_keeps_the_dummy_pixel():
    from enerf_amd.event_sampler import build_no_event_tables
    W, H = 4, 3
    xs, ys = np.meshgrid(np.arange(W), np.arange(H))
    ev = np.stack([xs.ravel(), ys.ravel(), np.linspace(1e9, 1.004e9, W * H), np.ones(W * H)], 1)   # every pixel fires
    got = build_no_event_tables(torch.from_numpy(ev), H, W, 1e6, 1.005e6)
    assert got["N_ev_chunks"] == 1 and got["coords"][0].shape == (1, 2) and float(got["coords"][0].abs().sum()) == 0


@pytest.mark.parametrize("device", ["cpu", pytest.param("cuda", marks=pytest.mark.gpu)])
def test_no_event_rays_match_the_reference_collate_with_scipy_poses(device):
    """no_event_rays against provider.py:1443-1476 restated with the reference's own scipy interpolators and the
    reference-pinned get_event_rays, same draws."""
    from scipy.interpolate import interp1d
    from scipy.spatial.transform import Slerp
    from enerf_amd.event_sampler import build_no_event_tables, no_event_rays
    from enerf_amd.events import get_event_rays
    from enerf_amd.pose_interp import PoseTrack
    ev, rect, W, H, t0, t1 = _no_event_case(5)
    t, R, p = _track(50, t0 * 1e3 - 5e6, t1 * 1e3 + 5e6, 8)            # nanoseconds, covering the batch
    picks = {}

    def choice(j, idxs, size):
        picks[j] = np.random.default_rng(7 + j).choice(idxs, size=size, replace=False)
        return picks[j]

    ref_tab = EC.no_event_tables(ev, ev[:, :2], H, W, t0, t1, rect, choice)
    tab = build_no_event_tables(torch.from_numpy(ev).to(device), H, W, t0, t1, rectify_map=rect,
                                keep=lambda j, cand: torch.from_numpy(picks[j].astype(np.int64) - 1))
    rng = np.random.default_rng(11)
    B = 512
    intr = (35.0, 34.0, 19.5, 14.5)
    track = PoseTrack(t, R.as_matrix(), p, device=device)
    for chunk in (0, 2):
        neidx = rng.integers(0, len(ref_tab["coords"][chunk]), B // 2)
        u = rng.random((B // 2, 2))
        ref, tss = EC.no_event_rays(ref_tab, Slerp(t, R), interp1d(x=t, y=p, axis=0, kind="cubic", bounds_error=True),
                                    get_event_rays, intr, B, chunk, neidx, u)
        got = no_event_rays(tab, track, intr, B, draws={"chunk": chunk, "idx": torch.from_numpy(neidx),
                                                       "u": torch.from_numpy(u)})
        assert np.allclose(got["tss_us"].cpu().numpy(), tss, rtol=0, atol=1e-6)
        for a, b in (("rays_no_evs_o1", "rays_evs_o1"), ("rays_no_evs_d1", "rays_evs_d1"),
                     ("rays_no_evs_o2", "rays_evs_o2"), ("rays_no_evs_d2", "rays_evs_d2")):
            assert got[a].shape == (1, B // 2, 3)
            np.testing.assert_allclose(got[a].cpu().numpy(), ref[b].numpy(), rtol=2e-6, atol=2e-6)
    own = no_event_rays(tab, track, intr, B)                           # its own draws: shapes, chunk range, ordered times
    assert 0 <= own["chunk"] < 3 and bool((own["tss_us"][:, 0] <= own["tss_us"][:, 1]).all())
    lo, hi = tab["start_time_us"][own["chunk"]], tab["end_time_us"][own["chunk"]]
    assert float(own["tss_us"].min()) >= lo and float(own["tss_us"].max()) <= hi
