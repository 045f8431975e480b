"""enerf_amd/event_window.py against the reference's OWN event readers, run by oracle/make_golden.py: gold_event_readers --
EventSlicer (utils/event_utils.py:223-383, with its compute_ms_to_idx index) over a dict that answers like the h5 file it
expects, and load_contiguous_evs_batches_esim_ns (nerf/provider.py:27-82) over a directory of .npy files:
tests/golden/ref_event_readers.npz.  CPU only."""
import os

import numpy as np
import pytest
import torch

from util import golden


@pytest.fixture(scope="module")
def z():
    return golden("ref_event_readers")


def test_millisecond_index_equals_compute_ms_to_idx(z):
    from enerf_amd.event_window import EventTimeIndex
    idx = EventTimeIndex(torch.from_numpy(z["t_us"]), unit_per_ms=1000)
    assert np.array_equal(idx.ms_to_idx.numpy(), z["ms_to_idx"])
    idx_ns = EventTimeIndex(torch.from_numpy(z["t_us"] * 1000), unit_per_ms=1_000_000)      # the esim unit
    assert np.array_equal(idx_ns.ms_to_idx.numpy(), z["ms_to_idx"])


@pytest.mark.parametrize("tag,stored_index", [("plain", True), ("plain", False), ("offset", True)])
def test_windows_equal_event_slicer_get_events(z, tag, stored_index):
    from enerf_amd.event_window import EventTimeIndex
    t = torch.from_numpy(z["t_us"])
    off = int(z["offset_windows"][0, 0] - z["plain_windows"][0, 0]) if tag == "offset" else 0
    idx = EventTimeIndex(t, unit_per_ms=1000, t_offset=off,
                         ms_to_idx=torch.from_numpy(z["ms_to_idx"]) if stored_index else None)
    assert idx.t_final == int(z[f"{tag}_t_final"])
    x, y, p = (z[k].astype(np.int64) for k in ("x", "y", "p"))
    none = hit = 0
    for (a, b), want in zip(z[f"{tag}_windows"], z[f"{tag}_results"]):
        w = idx.window(int(a), int(b))
        if want[0] == -1:
            assert w is None, (a, b)
            none += 1
            continue
        assert w is not None, (a, b)
        lo, hi = w
        assert hi - lo == want[0], (a, b)
        if hi > lo:
            assert int(t[lo]) + off == want[1] and int(t[hi - 1]) + off == want[2]
            assert int(x[lo:hi].sum() + 7 * y[lo:hi].sum() + 13 * p[lo:hi].sum()) == want[3]
            hit += 1
    assert none >= 1 and hit >= 25          # both the "cannot guarantee" answer and real windows occur in the fixture


def test_esim_batches_equal_the_reference_loader(z, tmp_path):
    from enerf_amd.event_window import load_esim_event_batches
    for k in range(7):
        np.save(os.path.join(tmp_path, f"{k:06d}.npy"), z[f"esim_file{k}"])
    for tag in ("esim_a", "esim_b", "esim_c"):
        out = load_esim_event_batches(str(tmp_path), [int(v) for v in z[f"{tag}_idxs"]], hwf=(48, 64, 1.0))
        assert [len(b) for b in out] == z[f"{tag}_sizes"].tolist(), tag
        assert np.array_equal(np.concatenate(out), z[f"{tag}_cat"]), tag
