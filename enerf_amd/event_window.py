"""Event streams by time: the millisecond index and window lookup of the reference's event readers, and the esim
`events/*.npy` loader, feeding `event_sampler.build_event_tables` (row f3 of SURVEY.md 8: the data either side of the
event-pair kernel).

Reference behaviour restated here:
  EventTimeIndex.ms_to_idx   utils/event_utils.py:389-408 (compute_ms_to_idx) and the contract spelled out at :236-249:
                             t[ms_to_idx[ms]] >= ms * 1000 and t[ms_to_idx[ms] - 1] < ms * 1000
  EventTimeIndex.window      EventSlicer.get_events, utils/event_utils.py:256-300: indices of the events with
                             t_start <= t < t_end, None when the window leaves the indexed range
  load_esim_event_batches    nerf/provider.py:27-82: batches between consecutive frame indices, (x, y, t_ns, p),
                             polarities mapped to -1 / +1, coordinates checked against the sensor

The reference walks the conservative millisecond window with two linear scans on the host (numba); here both borders
are binary searches over the time column wherever it lives (the device, next to the event tables), so a window lookup
is two `searchsorted` calls and no copy of the time stamps.
"""
import math
import os

import numpy as np
import torch


class EventTimeIndex:
    def __init__(self, t, unit_per_ms=1000, t_offset=0, ms_to_idx=None):
        """t: sorted 1-D time stamps (any integer / float dtype, any device); unit_per_ms = 1000 for microseconds
        (h5 files), 1e6 for nanoseconds (esim).  `ms_to_idx`: a stored index (the h5 files carry one) to use as is."""
        if t.ndim != 1 or t.numel() == 0:
            raise ValueError("EventTimeIndex: t must be a non-empty 1-D tensor")
        self.t, self.unit_per_ms, self.t_offset = t, unit_per_ms, int(t_offset)
        if ms_to_idx is None:
            ms_end = int(math.floor(float(t.max())) / unit_per_ms)
            marks = torch.arange(0, ms_end + 1, device=t.device, dtype=torch.int64) * int(unit_per_ms)
            ms_to_idx = torch.searchsorted(t, marks.to(t.dtype), right=False)
        self.ms_to_idx = ms_to_idx.to(torch.int64)

    @property
    def t_final(self):
        return int(self.t[-1]) + self.t_offset

    def window(self, t_start, t_end):
        """Indices (first, one past last) of the events with t_start <= t < t_end (times in the unit of `t`, offset
        included); None when the millisecond index does not cover the window (the reference's "cannot guarantee")."""
        if not t_start < t_end:
            raise AssertionError("EventTimeIndex.window: t_start must precede t_end")
        t_start, t_end = t_start - self.t_offset, t_end - self.t_offset
        w0 = max(math.floor(t_start / self.unit_per_ms), 0)
        w1 = math.ceil(t_end / self.unit_per_ms)
        n = self.ms_to_idx.numel()
        if w0 >= n or w1 >= n:
            return None
        edges = torch.tensor([t_start, t_end], device=self.t.device).to(self.t.dtype)
        a, b = (int(v) for v in self.ms_to_idx[[w0, w1]].tolist())
        i = torch.searchsorted(self.t[a:b].contiguous(), edges, right=False).tolist()
        return a + i[0], a + i[1]


def load_esim_event_batches(eventdir, idxs, hwf=None, microseconds=False):
    """List of float64 arrays [n_i, 4] = (x, y, t_ns, p in {-1, +1}); batch i holds every event file from idxs[i] up to,
    not including, idxs[i + 1]; the last batch is the file idxs[-1] alone."""
    if len(idxs) == 0:
        raise AssertionError("load_esim_event_batches: no indices")
    files = sorted(f for f in os.listdir(eventdir) if f.endswith(".npy"))
    read = lambda k: np.load(os.path.join(eventdir, files[k]))[:, :4]      # noqa: E731
    if any(b <= a for a, b in zip(idxs, idxs[1:])):
        raise AssertionError("load_esim_event_batches: indices must increase")
    batches = [np.concatenate([read(k) for k in range(a, b)]) for a, b in zip(idxs, idxs[1:])]
    batches.append(read(idxs[-1]))
    H, W = (hwf[0], hwf[1]) if hwf is not None else (720, 1280)
    out = []
    for ev in batches:
        ev = np.asarray(ev, dtype=np.float64)
        if microseconds:
            ev = ev * (1, 1, 1000.0, 1)
        if ev.ndim != 2 or ev.shape[1] != 4:
            raise AssertionError("event batches are [n, 4] = (x, y, t, p)")
        if ev[:, 0].min() < 0 or ev[:, 0].max() >= W or ev[:, 1].min() < 0 or ev[:, 1].max() >= H:
            raise AssertionError(f"event coordinates outside the {W} x {H} sensor")
        out.append(ev)
    if not any(np.any(ev[:, 3] == -1) for ev in out):                        # no -1 anywhere: polarities are {0, 1}
        out = [np.concatenate([ev[:, :3], 2 * ev[:, 3:] - 1], axis=1) for ev in out]
    for ev in out:
        if not set(np.unique(ev[:, 3])) <= {-1.0, 1.0}:
            raise AssertionError("polarities must be -1 / +1")
    return out
