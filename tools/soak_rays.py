"""Randomised differential soak of OccupancyMap::calcRange on the device (the beam model's raycast: chessboard-distance
jumps, both minor-axis forms) against the oracle's cell-by-cell Bresenham walk: random maps (empty, sparse, dense,
walled), origins inside, on the border and outside, every direction incl. the axes and diagonals, short and long
max ranges.  Exact.
usage: python tools/soak_rays.py [maps] [seed]"""
import math
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import badger_amcl_amd as bpf
from oracle import pyoracle as orc

def run(maps=30, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    rays = 0
    for case in range(maps):
        sx, sy = int(rng.integers(4, 400)), int(rng.integers(4, 400))
        res = float(rng.choice([0.05, 0.1]))
        dens = float(rng.choice([0.0, 0.0003, 0.003, 0.03, 0.3]))
        cells = np.full((sy, sx), -1, dtype=np.int32)
        cells[rng.random(cells.shape) < dens] = 1
        cells[rng.random(cells.shape) < 0.01] = 0
        if rng.random() < 0.5:
            cells[0, :] = cells[-1, :] = 1
            cells[:, 0] = cells[:, -1] = 1
        origin = (float(np.float32(rng.uniform(-5, 5))), float(np.float32(rng.uniform(-5, 5))))
        om = orc.OccupancyMap(cells, res, origin)
        m = bpf.OccupancyMap(e, res)
        m.setCells(cells)
        m.setOrigin(origin)
        n = 1500
        w, h = sx * res, sy * res
        ox = origin[0] + rng.uniform(-0.6 * w, 0.6 * w, n)
        oy = origin[1] + rng.uniform(-0.6 * h, 0.6 * h, n)
        oa = rng.uniform(-math.pi, math.pi, n)
        special = np.array([0.0, math.pi / 2, math.pi, -math.pi / 2, math.pi / 4, 3 * math.pi / 4, -math.pi / 4, 1e-9, -1e-9])
        oa[: special.size * 20] = np.tile(special, 20)
        mr = rng.choice([0.02, 0.5, 3.0, 30.0, 200.0], n)
        got = m.calcRange(ox, oy, oa, mr)
        want = np.array([om.calc_range(float(ox[i]), float(oy[i]), float(oa[i]), float(mr[i])) for i in range(n)])
        rays += n
        if not np.array_equal(got, want):
            bad += 1
            k = int(np.flatnonzero(got != want)[0])
            print("MISMATCH map %d (%dx%d res %g dens %g): ray %d from (%.3f, %.3f) angle %.6f max %.2f: %r vs %r" %
                  (case, sx, sy, res, dens, k, ox[k], oy[k], oa[k], mr[k], got[k], want[k]), flush=True)
    print("%d maps, %d rays, %d maps mismatching, %.0f s" % (maps, rays, bad, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 30,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
