"""oracle/density_update.py (numpy restatement of the deterministic parts of update_extra_state) on hand-checked cases."""
import numpy as np

from oracle import density_update as OD
from oracle import oracle as O


def test_cascade_geometry_and_cell_centres():
    assert OD.cascade_geometry(0, 3, 128) == (1 - 1 / 128, 1 / 128)
    assert OD.cascade_geometry(2, 3, 128) == (3 - 3 / 128, 3 / 128)          # bound caps the last cascade
    idx = O.morton3D(np.array([[0, 0, 0], [127, 127, 127], [64, 0, 127]], np.int32))
    c = OD.cell_centres(idx, 1, 3, 128)
    span = 2 - 2 / 128
    assert np.allclose(c[0], -span) and np.allclose(c[1], span)
    assert np.allclose(c[2], [(2 * 64 / 127 - 1) * span, -span, span])


def test_apply_update_by_hand():
    H3 = 64
    grid = np.zeros((2, H3), np.float32)
    grid[0, 1], grid[0, 2], grid[0, 3], grid[1, 5] = 1.0, 0.2, -1.0, 0.5
    indices = [np.array([1, 2, 3, 4]), np.array([5, 6])]
    sigmas = [np.array([0.5, 1.0, 9.0, 2.0], np.float32), np.array([0.1, 0.0], np.float32)]
    g, mean, bits = OD.apply_update(grid, indices, sigmas, 0.5, 0.9, 0.01)
    assert np.allclose(g[0, 1:5], [0.9, 0.5, -1.0, 1.0])      # decay wins, sample wins, untrained stays, fresh cell
    assert np.allclose(g[1, 5:7], [0.45, 0.0])
    want_mean = (0.9 + 0.5 + 1.0 + 0.45) / (2 * H3)
    assert abs(mean - want_mean) < 1e-7
    occupied = np.unpackbits(bits, bitorder="little").astype(bool)
    assert np.array_equal(np.nonzero(occupied)[0], [1, 2, 4, H3 + 5])       # > min(mean, 0.01) = 0.01
    assert OD.mean_count(np.array([[10, 1], [20, 1], [31, 1]]), 2) == 15
    assert OD.mean_count(np.zeros((16, 2)), 0) is None
