#!/usr/bin/env python3
"""Wall time of the "next" rows that moved to the device in round 2, device against host evaluation:
cluster statistics (tracking set and a 100 k spread set) and the 3-D distance-LUT builder (a hall of 400 x 300 x 40
voxels).  Run on the GPU box: python tools/time_next_rows.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import badger_amcl_amd as bpf  # noqa: E402
import badger_amcl_amd.pf as hpf  # noqa: E402
from badger_amcl_amd import synth  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    e = bpf.Engine(0)
    size = 2000
    pose = synth.true_pose(size)
    out = {}
    for name, s in (("tracking_1800", synth.converged_cloud(1800, pose, seed=1)),
                    ("spread_100k", synth.spread_cloud(100000, size, seed=2))):
        pf = bpf.ParticleFilter(e, 100, s.shape[0], 0.0, 0.0, 85.0)
        pf.initWithSamples(s, leaf_count=1)

        def stats():
            pf.fillWeights(1.0 / s.shape[0])  # bumps the set's epoch: the statistics are evaluated again
            pf.getMaxWeightPose()
        for mode in (0, 1):
            e.set_option(hpf.OPT_STATS_HOST, mode)
            out["stats_%s_%s_ms" % (name, "host" if mode else "device")] = timed(stats, 5 if mode else 20)
        e.set_option(hpf.OPT_STATS_HOST, 0)
    # 3-D builder: a hall with pillars
    lo, hi = (-200, -150, -2), (199, 149, 37)
    occ = []
    ii, jj = np.meshgrid(np.arange(lo[0], hi[0] + 1), np.arange(lo[1], hi[1] + 1), indexing="ij")
    for k in (lo[2], hi[2]):
        occ.append(np.stack([ii.ravel(), jj.ravel(), np.full(ii.size, k)], axis=1))
    for k in range(lo[2], hi[2] + 1):
        xs = np.arange(lo[0], hi[0] + 1)
        ys = np.arange(lo[1], hi[1] + 1)
        occ.append(np.stack([xs, np.full(xs.size, lo[1]), np.full(xs.size, k)], axis=1))
        occ.append(np.stack([xs, np.full(xs.size, hi[1]), np.full(xs.size, k)], axis=1))
        occ.append(np.stack([np.full(ys.size, lo[0]), ys, np.full(ys.size, k)], axis=1))
        occ.append(np.stack([np.full(ys.size, hi[0]), ys, np.full(ys.size, k)], axis=1))
        for px in range(lo[0] + 40, hi[0], 80):
            for py in range(lo[1] + 40, hi[1], 80):
                blk = np.stack(np.meshgrid(np.arange(px, px + 6), np.arange(py, py + 6), indexing="ij"), axis=-1)
                occ.append(np.concatenate([blk.reshape(-1, 2), np.full((36, 1), k)], axis=1))
    occ = np.ascontiguousarray(np.concatenate(occ).astype(np.int32))
    om = bpf.OctoMap(e, 0.05)
    for mode in (0, 1):
        e.set_option(hpf.OPT_LUT_HOST, mode)
        t0 = time.perf_counter()
        om.updateDistancesLUT(occ, lo, hi, 0.3)
        out["lut3d_%s_ms" % ("host" if mode else "device")] = (time.perf_counter() - t0) * 1e3
    e.set_option(hpf.OPT_LUT_HOST, 0)
    out["lut3d_occupied_voxels"] = int(occ.shape[0])
    out["lut3d_volume"] = [hi[d] - lo[d] + 1 for d in range(3)]
    import json
    print(json.dumps(out))
    e.close()


if __name__ == "__main__":
    main()
