"""Randomised differential soak of Odom::updateAction on the device against the oracle: the five motion models, random
noise parameters, odometry poses and deltas (incl. zero and backward motion), set sizes around the 2 048-attempt tile of
the Box-Muller accounting, random drand48 states.  Exact: the stream position and the weights; poses within 1e-12.
usage: python tools/soak_motion.py [cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import badger_amcl_amd as bpf
from badger_amcl_amd import synth
from oracle import pyoracle as orc

def run(cases=300, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    worst = 0.0
    for case in range(cases):
        model = int(rng.integers(0, 5))
        n = int(rng.choice([1, 2, 341, 342, 343, 682, 683, 1000, 2047, 2048, 2049, 7000, 40000]))
        alpha = tuple(float(x) for x in rng.uniform(0.0, 0.5, 5))
        if rng.random() < 0.1:
            alpha = (0.0,) * 5
        pose = (float(rng.uniform(-20, 20)), float(rng.uniform(-20, 20)), float(rng.uniform(-3.1, 3.1)))
        kind = rng.integers(0, 4)
        if kind == 0:
            delta = (0.0, 0.0, 0.0)
        elif kind == 1:
            delta = (float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.02, 0.02)), float(rng.uniform(-0.02, 0.02)))
        else:
            delta = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.8, 0.8)))
        absm = tuple(abs(d) * float(rng.uniform(1.0, 1.5)) for d in delta)
        s = synth.spread_cloud(n, 400, seed=int(rng.integers(0, 10000)))
        s[:, 3] = rng.uniform(0.1, 1.0, n)
        rng0 = int(rng.integers(0, 1 << 48))
        pf = bpf.ParticleFilter(e, 1, n, 0.0, 0.0, 85.0)
        pf.setRngState(rng0)
        pf.initWithSamples(s, leaf_count=1)
        od = bpf.Odom(e)
        od.setModel(model, *alpha)
        od.updateAction(pf, bpf.OdomData(pose, delta, absm))
        got = pf.getCurrentSet().samples
        want = s.copy()
        st = orc.odom_update_action(model, alpha, pose, delta, absm, want, rng0)
        d = float(np.abs(got[:, :3] - want[:, :3]).max())
        ok = pf.getRngState() == st and np.array_equal(got[:, 3], want[:, 3]) and d <= 1e-12
        worst = max(worst, d)
        if not ok:
            bad += 1
            print("MISMATCH case %d: model %d n %d alpha %s delta %s rng %x: pose diff %.2e, stream %s" %
                  (case, model, n, alpha, delta, rng0, d, pf.getRngState() == st), flush=True)
    print("%d cases, %d mismatching, worst pose difference %.2e, %.0f s" % (cases, bad, worst, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 300,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
