"""Randomised differential soak of the distance-LUT builders against the oracle: the 2-D builder in the reference's
order (priority-queue brushfire, tie order included) bit for bit on random maps; the exact device EDT against its
definition (<= the brushfire, capped, on the (a, b) lattice); the 3-D FIFO brushfire on the device, byte for byte.
usage: python tools/soak_lut.py [cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import badger_amcl_amd as bpf
from oracle import pyoracle as orc

def run(cases=40, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    for case in range(cases):
        # ---- 2-D
        sx, sy = int(rng.integers(8, 260)), int(rng.integers(8, 260))
        res = float(rng.choice([0.05, 0.1, 0.025]))
        max_dist = float(rng.choice([0.2, 0.5, 1.0, 2.0]))
        dens = float(rng.choice([0.0, 0.0005, 0.005, 0.05, 0.4]))
        cells = np.full((sy, sx), -1, dtype=np.int32)
        cells[rng.random(cells.shape) < dens] = 1
        cells[rng.random(cells.shape) < 0.02] = 0
        if rng.random() < 0.5:
            cells[0, :] = cells[-1, :] = 1
            cells[:, 0] = cells[:, -1] = 1
        origin = (float(np.float32(sx * res / 2)), float(np.float32(sy * res / 2)))
        want = np.asarray(orc.OccupancyMap(cells, res, origin).update_distances_lut(max_dist), dtype=np.float32).reshape(-1)
        m = bpf.OccupancyMap(e, res)
        m.setCells(cells)
        m.setOrigin(origin)
        m.updateDistancesLUTReference(max_dist)
        got = m.getDistancesLUT().reshape(-1)
        ok2 = np.array_equal(got, want)
        m.updateDistancesLUTExact(max_dist)
        edt = m.getDistancesLUT().reshape(-1)
        ok2 = ok2 and bool(np.all(edt <= want)) and bool(np.all(edt >= 0))
        # ---- 3-D
        mn = tuple(int(v) for v in (-rng.integers(3, 30), -rng.integers(3, 30), -rng.integers(1, 8)))
        mx = tuple(int(v) for v in (rng.integers(3, 30), rng.integers(3, 30), rng.integers(1, 12)))
        k = int(rng.choice([0, 1, 5, 200, 1500]))
        occ = np.stack([rng.integers(mn[d], mx[d] + 1, k) for d in range(3)], axis=1).astype(np.int32) if k else \
            np.zeros((0, 3), dtype=np.int32)
        r3 = float(rng.choice([0.05, 0.2]))
        md3 = r3 * float(rng.choice([2, 6, 8]))
        w3 = orc.OctoMapLUT(mn, mx, r3, md3)
        w3.build(occ)
        om = bpf.OctoMap(e, r3)
        om.updateDistancesLUT(occ, mn, mx, md3)
        pi, dr = om.getDistancesLUT()
        ok3 = np.array_equal(pi, w3.pose_indices) and np.array_equal(dr, w3.distance_ratios)
        if not (ok2 and ok3):
            bad += 1
            print("MISMATCH case %d: 2-D %dx%d res %g max_dist %g dens %g -> %s; 3-D %s..%s %d voxels res %g max_dist %g -> %s" %
                  (case, sx, sy, res, max_dist, dens, ok2, mn, mx, k, r3, md3, ok3), flush=True)
    print("%d cases, %d mismatching, %.0f s" % (cases, bad, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
