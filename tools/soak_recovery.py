"""Randomised differential soak of the recovery branch (w_diff > 0: augmented-MCL random poses from
Node::randomFreeSpacePose, particle_filter.cpp:295-324,383-388) against the oracle over three cycles of progressively
worse scans: which draws become random poses, the poses, the interleaved drand48 consumption, the grown systematic
count, the reset of the running averages.  Serial CDF (as tests/test_gpu_parity.py::test_recovery_... runs it).
usage: python tools/soak_recovery.py [cases] [seed]"""
import os
import sys
import time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import badger_amcl_amd as bpf
import badger_amcl_amd.pf as hpf
from oracle import pyoracle as orc
from scenario import Scenario

def run(cases=60, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    e.set_option(hpf.OPT_CDF_SERIAL, 1)
    t0 = time.time()
    bad = 0
    recovered = 0
    for case in range(cases):
        n = int(rng.choice([300, 1000, 2500, 6000, 12000]))
        resampler = int(rng.integers(0, 2))
        alpha = (float(rng.choice([0.001, 0.01])), float(rng.choice([0.1, 0.5])))
        e.set_option(hpf.OPT_KLD_DEVICE_MIN, int(rng.choice([1, 8192])))
        seed = int(rng.integers(1, 100000))
        sc_ = Scenario(orc, size=200, n=n, beams=61, cloud=str(rng.choice(["mixture", "converged"])),
                       seed=int(rng.integers(1, 10000)))
        m, sc, pf, data = sc_.gpu_objects(e, 61, "lf", min_samples=int(rng.choice([10, 100])), seed=seed, alpha=alpha)
        pf.setResampleModel(resampler)
        pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
        opf = orc.ParticleFilter(pf.min_samples, n, alpha[0], alpha[1], 85.0, seed=seed)
        opf.set_resample_model(resampler)
        opf.set_samples(sc_.samples)
        opf.set_random_pose_source(sc_.omap, sc_.map_factors[2])
        scans = [sc_.ranges, np.clip(sc_.ranges * float(rng.uniform(0.4, 0.8)), 0.05, 29.0),
                 np.full(61, float(rng.uniform(0.5, 3.0)))]
        ok = True
        for cycle, ranges in enumerate(scans):
            sc.updateSensor(pf, bpf.PlanarData(ranges, sc_.angles, sc_.range_max))
            cur = pf.getCurrentSet()
            st0 = pf.getState()
            opf.set_samples(cur.samples, leaf_count=st0.leaf_count)
            opf.pf.w_slow, opf.pf.w_fast = st0.w_slow, st0.w_fast
            opf.pf.rng = pf.getRngState()
            pf.updateResample()
            out = opf.update_resample()
            st1 = pf.getState()
            M = out.sample_count
            after = pf.getCurrentSet().samples
            ok = (out.status == 0 and abs(st1.w_diff - out.w_diff) <= 1e-12 and st1.sample_count == M and
                  st1.leaf_count == out.leaf_count and st1.bin_count == out.node_count and
                  np.array_equal(after[:, :3], opf.samples[:M, :3]) and np.all(after[:, 3] == 1.0 / M) and
                  pf.getRngState() == opf.pf.rng)
            if out.w_diff > 0:
                recovered += 1
                ok = ok and st1.w_slow == 0.0 and st1.w_fast == 0.0
            if not ok:
                break
        if not ok:
            bad += 1
            print("MISMATCH case %d cycle %d: n %d resampler %d alpha %s seed %d w_diff %g/%g M %d/%d" %
                  (case, cycle, n, resampler, alpha, seed, st1.w_diff, out.w_diff, st1.sample_count, M), flush=True)
    e.set_option(hpf.OPT_CDF_SERIAL, 0)
    e.set_option(hpf.OPT_KLD_DEVICE_MIN, 8192)
    print("%d cases x 3 cycles, %d resamples with w_diff > 0, %d cases mismatching, %.0f s" % (cases, recovered, bad, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
