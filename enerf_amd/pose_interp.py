"""Camera pose at arbitrary event times, on the device.

The reference interpolates the high-rate pose track with scipy on the host, per step, for 2 x batch_size_evs event times
(`Slerp` for rotations, `interp1d(kind="cubic")` for translations: nerf/provider.py:1142-1143, 1411-1420), unless it was
told to pre-interpolate a pose per event (`precompute_evs_poses`: "fast, but large memory requirement").  Here the track
is turned once into per-segment tables -- R_i, the rotation vector log(R_i^T R_{i+1}), and the cubic's four
coefficients per axis (scipy's own spline construction, so the interpolant IS interp1d's) -- and evaluated where the
events live: `PoseTrack.poses_at(t)` as a tensor program, or fused with the pair sampling and the ray generation in
csrc/event_pairs.hip (event_sampler.event_pair_rays).
"""
import numpy as np
import torch


class PoseTrack:
    def __init__(self, times_ns, rots, trans, device="cpu"):
        """times_ns [K] increasing, rots [K,3,3], trans [K,3] (numpy / tensors; K >= 4 for the cubic)."""
        from scipy.interpolate import PPoly, make_interp_spline
        from scipy.spatial.transform import Rotation
        t = np.asarray(times_ns, dtype=np.float64)
        R = Rotation.from_matrix(np.asarray(rots, dtype=np.float64))
        rotvec = (R[:-1].inv() * R[1:]).as_rotvec()                         # what Slerp scales by alpha per segment
        # interp1d(kind="cubic") == make_interp_spline(k=3) (not-a-knot); as a piecewise polynomial per segment
        trans = np.asarray(trans, dtype=np.float64)
        per_axis = []
        for ax in range(3):
            pp = PPoly.from_spline(make_interp_spline(t, trans[:, ax], k=3))
            seg = np.searchsorted(pp.x, t[:-1], side="right") - 1           # PPoly breakpoints carry repeated end knots
            per_axis.append(self._recentre(pp.c[:, seg, None], t[:-1] - pp.x[seg]))      # on the segment's own start
        coef = np.concatenate(per_axis, axis=-1)                            # [4, K-1, 3], highest power first
        dd = dict(dtype=torch.float64, device=device)
        self.knots = torch.tensor(t, **dd)
        self.rot = torch.tensor(R.as_matrix().reshape(-1, 9), **dd).contiguous()
        self.rotvec = torch.tensor(rotvec, **dd).contiguous()
        self.tcoef = torch.tensor(np.transpose(coef, (1, 0, 2)).copy(), **dd).contiguous()      # [K-1, 4, 3]
        self.K = len(t)

    @staticmethod
    def _recentre(c, s):
        """p(u + s) as a polynomial in u, for cubic coefficients c [4, S, 3] (highest first) and shifts s [S]."""
        s = s[:, None]
        a, b, cc, d = c[0], c[1], c[2], c[3]
        return np.stack([a, 3 * a * s + b, 3 * a * s * s + 2 * b * s + cc, ((a * s + b) * s + cc) * s + d])

    def to(self, device):
        for n in ("knots", "rot", "rotvec", "tcoef"):
            setattr(self, n, getattr(self, n).to(device))
        return self

    def poses_at(self, t):
        """t [M] (any float dtype, same unit as the track) -> camera-to-world [M, 3, 4] fp32 on the track's device.
        Times outside the track raise, like interp1d(bounds_error=True)."""
        t = t.to(self.knots.device, torch.float64)
        if bool(((t < self.knots[0]) | (t > self.knots[-1])).any()):
            raise ValueError("pose query outside the track")
        seg = (torch.searchsorted(self.knots, t, right=True) - 1).clamp(0, self.K - 2)
        t0 = self.knots[seg]
        alpha = (t - t0) / (self.knots[seg + 1] - t0)
        w = self.rotvec[seg] * alpha[:, None]
        th2 = (w * w).sum(-1)
        th = th2.sqrt()
        small = th < 1e-6
        safe = torch.where(small, torch.ones_like(th), th)
        a = torch.where(small, 1 - th2 / 6, torch.sin(safe) / safe)
        b = torch.where(small, 0.5 - th2 / 24, (1 - torch.cos(safe)) / (safe * safe))
        K = torch.zeros(t.shape[0], 3, 3, dtype=torch.float64, device=t.device)
        K[:, 0, 1], K[:, 0, 2], K[:, 1, 0] = -w[:, 2], w[:, 1], w[:, 2]
        K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -w[:, 0], -w[:, 1], w[:, 0]
        E = torch.eye(3, dtype=torch.float64, device=t.device) + a[:, None, None] * K + b[:, None, None] * (K @ K)
        R = self.rot[seg].view(-1, 3, 3) @ E
        u = (t - t0)[:, None]
        c = self.tcoef[seg]
        p = ((c[:, 0] * u + c[:, 1]) * u + c[:, 2]) * u + c[:, 3]
        return torch.cat([R, p[:, :, None]], dim=-1).float()
