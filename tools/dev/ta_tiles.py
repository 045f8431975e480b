"""Per-tile timing of the last k_grid_tile_adam launch of a training step (needs the -DENERF_TA_TIMING variant library:
ENERF_LIB_PATH=enerf_amd/lib/variants/lib_tatime.so python tools/dev/ta_tiles.py)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0], "--no-cpu-baseline", "--render-frames", "0", "--probe-steps", "0", "--other-legs", "0",
            "--strong-rays", "0", "--steps", "40", "--warmup", "24"] + sys.argv[1:]
import bench  # noqa
bench.main()
from enerf_amd import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_uint32 * (4 * 4096))()
assert lib.enerf_debug_ta_log(out) == 0
a = np.frombuffer(out, dtype=np.uint32).reshape(4096, 4).astype(np.int64)
a = a[a[:, 3] != 0]
t0 = a[:, 2].min()
beg, end = (a[:, 2] - t0) / 100.0, (a[:, 3] - t0) / 100.0     # us
print("tiles", len(a), "span %.1f us" % end.max())
for lv in range(16):
    m = a[:, 0] == lv
    if m.any():
        d = end[m] - beg[m]
        print("level %2d: tiles %4d  records/tile mean %7.0f max %7d   us/tile mean %6.2f max %6.2f   first begin %6.1f last end %6.1f"
              % (lv, m.sum(), a[m, 1].mean(), a[m, 1].max(), d.mean(), d.max(), beg[m].min(), end[m].max()))
late = np.argsort(-end)[:12]
print("last to finish:", [(int(a[i, 0]), int(a[i, 1]), round(float(beg[i]), 1), round(float(end[i]), 1)) for i in late])

wg = (ctypes.c_uint32 * (2 * 2048))()
assert lib.enerf_debug_ta_wg(wg) == 0
w = np.frombuffer(wg, dtype=np.uint32).reshape(2048, 2).astype(np.int64)
w = w[w[:, 1] != 0]
print("workgroups", len(w), "entry min %.1f max %.1f  exit min %.1f max %.1f (us, relative to the first tile's begin)" % (
    (w[:, 0].min() - t0) / 100., (w[:, 0].max() - t0) / 100., (w[:, 1].min() - t0) / 100., (w[:, 1].max() - t0) / 100.))

ph = (ctypes.c_uint64 * 256)()
assert lib.enerf_debug_bin_ph(ph, 0) == 0
print("binning pass, us per workgroup by phase (zero lists | load + rank | scan + reserve | stage + reservation answer | copy-out):")
for lv in range(16):
    n = max(1, ph[lv * 8 + 7])
    v = [ph[lv * 8 + k] / n / 100.0 for k in range(5)]
    print("  level %2d: %s  total %.2f   workgroups %d" % (lv, " ".join("%6.2f" % x for x in v), sum(v), ph[lv * 8 + 7]))

mk = (ctypes.c_uint64 * 8)()
assert lib.enerf_debug_ta_marks(mk) == 0
m0 = mk[0]
names = ["grid_fwd first entry", "grid_fwd last exit", "binning first entry", "binning last exit", "tile adam first entry", "tile adam last exit"]
print("last step, shader-side marks (us after the grid forward's first workgroup):")
for k in range(6):
    print("   %-24s %8.1f" % (names[k], (mk[k] - m0) / 100.0))
