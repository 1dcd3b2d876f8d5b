"""Randomised differential soak of computeClusterStatsForSet / getMaxWeightPose on the device against the oracle:
random clouds (one blob, many blobs, spread, mixtures), sizes on both sides of the single-block kernel's limit (4 096
samples, 64 clusters), random weights.  Labels, counts and the heaviest cluster exact, sums within 1e-12 (the checks
of tests/test_gpu_next_rows.py::_assert_stats_equal).
usage: python tools/soak_stats.py [cases] [seed]"""
import os
import sys
import time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import badger_amcl_amd as bpf
from oracle import pyoracle as orc
from test_gpu_next_rows import _assert_stats_equal, _oracle_stats

def run(cases=100, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    for case in range(cases):
        n = int(rng.choice([1, 2, 50, 1000, 4095, 4096, 4097, 9000, 30000]))
        kind = str(rng.choice(["blob", "blobs", "spread", "mix"]))
        s = np.zeros((n, 4))
        if kind == "blob":
            s[:, 0] = rng.normal(10, 0.3, n); s[:, 1] = rng.normal(5, 0.3, n); s[:, 2] = rng.normal(0.4, 0.1, n)
        elif kind == "blobs":
            k = int(rng.integers(2, 120))
            c = rng.uniform(0, 40, (k, 2)); th = rng.uniform(-3.1, 3.1, k); w = rng.integers(0, k, n)
            s[:, :2] = c[w] + rng.normal(0, 0.2, (n, 2)); s[:, 2] = th[w] + rng.normal(0, 0.05, n)
        elif kind == "spread":
            s[:, 0] = rng.uniform(0, 20, n); s[:, 1] = rng.uniform(0, 20, n); s[:, 2] = rng.uniform(-3.1, 3.1, n)
        else:
            h = n // 2
            s[:h, 0] = rng.normal(10, 0.3, h); s[:h, 1] = rng.normal(5, 0.3, h); s[:h, 2] = rng.normal(0.4, 0.1, h)
            s[h:, 0] = rng.uniform(0, 20, n - h); s[h:, 1] = rng.uniform(0, 20, n - h); s[h:, 2] = rng.uniform(-3.1, 3.1, n - h)
            s[:] = s[rng.permutation(n)]
        w = rng.uniform(0.01, 1.0, n)
        s[:, 3] = w / w.sum()
        pf = bpf.ParticleFilter(e, 1, n, 0.0, 0.0, 85.0)
        pf.initWithSamples(s)
        try:
            _assert_stats_equal(pf, _oracle_stats(orc, s, n), exact=False, set_atol=1e-9)
        except AssertionError as ex:
            bad += 1
            print("MISMATCH case %d: n %d kind %s: %s" % (case, n, kind, str(ex).splitlines()[0][:200]), flush=True)
        if case % 20 == 19:
            print("%d cases, %d mismatching, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
    print("%d cases, %d mismatching, %.0f s" % (cases, bad, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 100,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
