"""Randomised differential soak of the resident cycle -- updateSensor (scoring, normalisation, running averages), then
updateResample -- against the oracle, through whichever launches the engine picks (the two-launch tracking form, the
general path, the histogram tree's three forms).  As in tests/test_gpu_parity.py::test_update_resample_matches_oracle
the oracle resamples the weights the engine produced, so the resample must agree exactly.
usage: python tools/soak_cycle.py [cases] [seed]"""
import os
import sys
import time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import badger_amcl_amd as bpf
from oracle import pyoracle as orc
from scenario import Scenario, rel_err

def run(cases=100, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    worst = 0.0
    for case in range(cases):
        size = int(rng.choice([200, 400]))
        n = int(rng.choice([100, 257, 1000, 3000, 5000, 12000, 30000]))
        beams = int(rng.choice([30, 61, 91, 181]))
        model = str(rng.choice(["lf", "gompertz", "prob"]))
        cloud = str(rng.choice(["converged", "converged", "mixture", "spread"]))
        resampler = int(rng.integers(0, 2))
        min_s = int(rng.choice([10, 100, 500]))
        seed = int(rng.integers(1, 100000))
        sc_ = Scenario(orc, size=size, n=n, beams=beams, cloud=cloud, seed=int(rng.integers(1, 10000)))
        m, sc, pf, data = sc_.gpu_objects(e, beams, model, min_samples=min(min_s, n), seed=seed)
        pf.setResampleModel(resampler)
        cycles = int(rng.integers(1, 3))
        ok = True
        for c in range(cycles):
            before_scoring = pf.getCurrentSet().samples
            sc.updateSensor(pf, data)
            before = pf.getCurrentSet().samples
            st0 = pf.getState()
            # the sensor update against the oracle's (weights within 1e-9; one knife-edge weight allowed)
            want = before_scoring.copy()
            tot = sc_.oracle_apply(sc_.oracle_planar(beams, model), want, 0)
            if tot > 0:
                err = rel_err(before[:, 3], want[:, 3] / tot)
                ok = ok and int((err > 1e-9).sum()) <= 1
                worst = max(worst, float(err[err <= 1e-9].max()) if (err <= 1e-9).any() else 0.0)
            pf.updateResample()
            st1 = pf.getState()
            after = pf.getCurrentSet().samples
            opf = orc.ParticleFilter(min(min_s, n), n, 0.0, 0.0, 85.0, seed=1)
            opf.set_samples(before, leaf_count=st0.leaf_count)
            opf.pf.w_slow, opf.pf.w_fast = st0.w_slow, st0.w_fast
            # (the stream: srand48(seed) before the first cycle, then wherever the engine's previous resample left it)
            opf.pf.rng = rng_before if c else orc.ParticleFilter(2, 2, 0.0, 0.0, 85.0, seed=seed).pf.rng
            opf.set_resample_model(resampler)
            out = opf.update_resample()
            M = out.sample_count
            ok = ok and (out.status == 0 and st1.last_status == 0 and st1.sample_count == M and
                         st1.leaf_count == out.leaf_count and st1.bin_count == out.node_count and
                         np.array_equal(after[:, :3], opf.samples[:M, :3]) and np.all(after[:, 3] == 1.0 / M) and
                         pf.getRngState() == opf.pf.rng and st1.converged == out.converged)
            rng_before = pf.getRngState()
            if not ok:
                break
        if not ok:
            bad += 1
            print("MISMATCH case %d (cycle %d): size %d n %d beams %d model %s cloud %s resampler %d min %d seed %d" %
                  (case, c, size, n, beams, model, cloud, resampler, min_s, seed), flush=True)
        if case % 25 == 24:
            print("%d cases, %d mismatching, worst weight err %.2e, %.0f s" % (case + 1, bad, worst, time.time() - t0), flush=True)
    print("%d cases, %d mismatching, worst weight err %.2e, %.0f s" % (cases, bad, worst, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
