"""enerf_amd's event-pair sampling against the reference's OWN EventNeRFDataset (nerf/provider.py:1105-1500), constructed and
collated by oracle/make_golden.py: gold_collate with only its input/output stubbed (the parent's file-reading __init__, the
.npy reader, the pose plot): the constructor's per-pixel grouping loop, the no-event tables of load_events_at_frame_idxs and
collate are the reference's code, numpy's global draws recorded: tests/golden/ref_collate.npz.  The same draws go into event_sampler.sample_event_pairs / no_event_rays / PoseTrack /
get_event_rays on the CPU, and -- marked gpu -- into the one-launch device route (csrc/event_pairs.hip): pair ends and
polarity sums exact, rays to 1e-5."""
import numpy as np
import pytest
import torch

from util import golden

INTR = (14.0, 13.0, 8.0, 6.0)


@pytest.fixture(scope="module")
def z():
    return golden("ref_collate")


def _tables_and_track(z, dev="cpu"):
    from enerf_amd.event_sampler import build_event_tables
    from enerf_amd.pose_interp import PoseTrack
    t = build_event_tables(torch.from_numpy(z["events"]).to(dev))
    return t, PoseTrack(z["pose_ts"], z["pose_R"], z["pose_t"], device=dev)


def test_event_tables_equal_the_reference_constructor(z):
    """The grouping loop of the constructor (:1147-1199): events sorted by time, grouped per pixel in first-occurrence
    order, pixels with one event dropped; per-pixel counts and offsets, the last event of every pixel, successors."""
    t, _ = _tables_and_track(z)
    assert np.array_equal(t["events"].numpy(), z["tab_events"])
    assert np.array_equal(t["num_at_xy"].numpy(), z["tab_xy_numEvs_Idx"][:, 0])
    assert np.array_equal(t["first_at_xy"].numpy(), z["tab_xy_numEvs_Idx"][:, 1])
    assert np.array_equal(t["no_successor"].nonzero().flatten().numpy(), z["tab_idx_no_successor"])
    assert np.array_equal(t["num_successor"].numpy(), z["tab_num_successor_evs"])
    assert t["events"].shape[0] == int(z["tab_num_evs"])


def test_no_event_tables_equal_the_reference_loader(z):
    """load_events_at_frame_idxs' no-event tables (:1283-1351): chunks of ~20 ms between this batch's first stamp and the
    next batch's, the pixels without an event per chunk, thinned to 1 / N by np.random.choice (the recorded draws)."""
    from enerf_amd.event_sampler import build_no_event_tables
    ev = z["events"].astype(np.float64)
    start_us, end_us = 1e-3 * ev[0, 2], 1e-3 * float(z["events_next_first_ns"])
    n = int(z["noev_n_chunks"])
    out = build_no_event_tables(torch.from_numpy(ev), 12, 16, start_us, end_us,
                                keep=lambda j, cand: z[f"noev_choice{j}"] - 1)       # the reference numbers pixels from 1
    assert out["N_ev_chunks"] == n
    np.testing.assert_allclose(out["start_time_us"], z["noev_start_us"], rtol=1e-6)   # (the reference keeps them as fp32)
    np.testing.assert_allclose(out["end_time_us"], z["noev_end_us"], rtol=1e-6)
    for j in range(n):
        assert np.array_equal(out["coords"][j].numpy(), z[f"noev_coords{j}"]), j
    # and the candidates the reference drew from are the ones the restatement lists: every recorded choice is a candidate
    seen = []
    build_no_event_tables(torch.from_numpy(ev), 12, 16, start_us, end_us,
                          keep=lambda j, cand: (seen.append(cand.numpy()), cand[:0])[1])
    for j in range(n):
        assert np.all(np.isin(z[f"noev_choice{j}"] - 1, seen[j])) and len(z[f"noev_choice{j}"]) == int(len(seen[j]) / n)


def _u_end(t, starts, ends, acc_max):
    """The uniform that lands on the recorded np.random.randint(start + 1, start + 1 + n) draw."""
    nos = t["no_successor"].cpu().numpy().astype(bool)
    s = np.where(nos[starts], starts - 1, starts)
    n = t["num_successor"].cpu().numpy()[s]
    if acc_max:
        n = np.minimum(n, acc_max + 1)
    return s, (ends - (s + 1) + 0.5) / n


@pytest.mark.parametrize("tag,acc_max", [("acc", 0), ("acc_max3", 3), ("acc_noev", 0)])
def test_accumulated_pairs_and_rays_equal_the_reference_collate(z, tag, acc_max):
    from enerf_amd.event_sampler import sample_event_pairs
    from enerf_amd.events import get_event_rays
    t, track = _tables_and_track(z)
    starts = z[f"{tag}_draw_randint_first"].astype(np.int64)
    ends = z[f"{tag}_draw_randint_rest"][:64].astype(np.int64)
    s_ref, u = _u_end(t, starts, ends, acc_max)
    s, e, p, xs, ys = sample_event_pairs(t, 64, True, acc_max, draws={"start": torch.from_numpy(starts),
                                                                      "u_end": torch.from_numpy(u)})
    assert np.array_equal(s.numpy(), s_ref) and np.array_equal(e.numpy(), ends)
    assert np.array_equal(p.numpy(), z[f"{tag}_pols"])                      # sums of +-1: exact
    ev = t["events"]
    p1, p2 = track.poses_at(ev[s, 2]), track.poses_at(ev[e, 2])
    rays = get_event_rays(xs, ys, p1.unsqueeze(0), p2.unsqueeze(0), INTR)
    for k in ("rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2"):
        np.testing.assert_allclose(rays[k].numpy(), z[f"{tag}_{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


def test_single_successor_pairs_equal_the_reference_collate(z):
    from enerf_amd.event_sampler import sample_event_pairs
    t, _ = _tables_and_track(z)
    u_xy = z["single_draw_rand_first"]
    # np.random.choice(eidx, size, replace): the recorded result holds the chosen event ids; the restatement takes positions
    chosen = z["single_draw_choice_first"].astype(np.int64)
    per_pixel = (u_xy * t["num_at_xy"].numpy() - 1).astype(int) + t["first_at_xy"].numpy()
    pos = np.array([int(np.nonzero(per_pixel == c)[0][0]) for c in chosen])
    s, e, p, xs, ys = sample_event_pairs(t, 64, False, draws={"u_xy": torch.from_numpy(u_xy), "choice": torch.from_numpy(pos)})
    assert np.array_equal(s.numpy(), chosen) and np.array_equal(e.numpy(), chosen + 1)
    assert np.array_equal(p.numpy(), z["single_pols"])


def test_no_event_rays_equal_the_reference_collate(z):
    from enerf_amd.event_sampler import no_event_rays
    _, track = _tables_and_track(z)
    tag = "acc_noev"
    rest = z[f"{tag}_draw_randint_rest"]
    chunk, idx = int(rest[64]), rest[65:65 + 32].astype(np.int64)         # after the 64 window ends: the chunk, then 32 pixels
    u = z[f"{tag}_draw_random_first"]                                      # np.random.random((32, 2))
    n = int(z["noev_n_chunks"])
    coords = [torch.from_numpy(z[f"noev_coords{j}"]) for j in range(n)]
    no_evs = {"coords": coords, "N_ev_chunks": n, "start_time_us": z["noev_start_us"].tolist(),
              "end_time_us": z["noev_end_us"].tolist()}
    r = no_event_rays(no_evs, track, INTR, 64, draws={"chunk": chunk, "idx": torch.from_numpy(idx), "u": torch.from_numpy(u)})
    for k in ("rays_no_evs_o1", "rays_no_evs_d1", "rays_no_evs_o2", "rays_no_evs_d2"):
        np.testing.assert_allclose(r[k].numpy(), z[f"{tag}_{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("tag,acc_max", [("acc", 0), ("acc_max3", 3)])
def test_device_launch_equals_the_reference_collate(z, tag, acc_max):
    """csrc/event_pairs.hip (pair selection + polarity sums + pose interpolation + rays in one launch)."""
    from enerf_amd.event_sampler import event_pair_rays
    t, track = _tables_and_track(z, "cuda")
    starts = z[f"{tag}_draw_randint_first"].astype(np.int64)
    ends = z[f"{tag}_draw_randint_rest"][:64].astype(np.int64)
    s_ref, u = _u_end(t, starts, ends, acc_max)
    r = event_pair_rays(t, track, INTR, 64, acc_max, draws={"start": torch.from_numpy(starts), "u_end": torch.from_numpy(u)})
    assert np.array_equal(r["start"].cpu().numpy(), s_ref) and np.array_equal(r["end"].cpu().numpy(), ends)
    assert np.array_equal(r["pols"].cpu().numpy(), z[f"{tag}_pols"])
    for k in ("rays_evs_o1", "rays_evs_d1", "rays_evs_o2", "rays_evs_d2"):
        np.testing.assert_allclose(r[k].cpu().numpy(), z[f"{tag}_{k}"], rtol=1e-5, atol=1e-6, err_msg=k)


def test_frame_rays_of_collate_equal_the_reference(z):
    """The frame entries of the same collate call (:1414-1416, :1488-1493): get_rays with 16 pixels drawn from torch's
    stream, the frame's colours gathered at them."""
    from enerf_amd.events import get_rays
    torch.manual_seed(83)
    r = get_rays(torch.eye(4).unsqueeze(0), INTR, 12, 16, 16)
    np.testing.assert_allclose(r["rays_o"].numpy(), z["acc_rays_o"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(r["rays_d"].numpy(), z["acc_rays_d"], rtol=1e-6, atol=1e-7)
    images = torch.from_numpy(z["frame_images"])[[0]]
    got = torch.gather(images.view(1, -1, 3), 1, torch.stack(3 * [r["inds"]], -1))
    assert np.array_equal(got.numpy(), z["acc_images"])
