"""Randomised differential soak of updateResample (Seam B) against the oracle: random set sizes and clouds, weight
vectors with zeros and heavy tails, both resamplers, random min / max sample counts and KLD parameters.  The oracle
resamples the very weights the engine holds, so everything must agree exactly: sample count, poses (hence source
indices), weights 1 / M, leaf and bin counts, the drand48 state, updateConverged.
usage: python tools/soak_resample.py [cases] [seed]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import badger_amcl_amd as bpf
from badger_amcl_amd import synth
from oracle import pyoracle as orc

def run(cases=200, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    forms = {}
    for case in range(cases):
        n = int(rng.choice([3, 64, 100, 257, 1000, 2048, 2049, 5000, 20000, 60000]))
        kind = str(rng.choice(["converged", "spread", "mixture", "clusters"]))
        size = 400
        pose = synth.true_pose(size, 0.05)
        seed = int(rng.integers(1, 100000))
        if kind == "converged":
            s = synth.converged_cloud(n, pose, seed=seed)
        elif kind == "spread":
            s = synth.spread_cloud(n, size, 0.05, seed=seed)
        elif kind == "mixture":
            a = synth.converged_cloud(n - n // 2, pose, seed=seed)
            b = synth.spread_cloud(max(n // 2, 1), size, 0.05, seed=seed + 1)[: n // 2]
            s = np.ascontiguousarray(np.concatenate([a, b]))
        else:
            k = int(rng.integers(2, 40))
            c = rng.uniform(2, 18, (k, 2))
            which = rng.integers(0, k, n)
            s = np.zeros((n, 4))
            s[:, :2] = c[which] + rng.normal(0, 0.3, (n, 2))
            s[:, 2] = rng.uniform(-3.1, 3.1, n)
        w = rng.uniform(0.0, 1.0, n) ** float(rng.choice([1.0, 4.0, 12.0]))
        if rng.random() < 0.3:
            w[rng.random(n) < 0.5] = 0.0
        if w.sum() <= 0.0:
            w[:] = 1.0
        s[:, 3] = w / w.sum()
        min_s = int(rng.choice([2, 10, 100, 500]))
        min_s = min(min_s, n)
        max_s = int(rng.choice([n, max(min_s, n // 2), n + 100]))
        pop = (float(rng.choice([0.01, 0.0025, 0.05])), float(rng.choice([3.0, 0.99, 2.0])))
        resampler = int(rng.integers(0, 2))
        rs = int(rng.integers(1, 1 << 30))
        pf = bpf.ParticleFilter(e, min_s, max(max_s, n), 0.0, 0.0, 85.0)
        pf.setPopulationSizeParameters(*pop)
        pf.setResampleModel(resampler)
        pf.srand48(rs)
        pf.initWithSamples(s)
        st0 = pf.getState()
        pf.updateResample()
        st1 = pf.getState()
        after = pf.getCurrentSet().samples
        opf = orc.ParticleFilter(min_s, max(max_s, n), 0.0, 0.0, 85.0, seed=rs)
        opf.set_population_size_parameters(*pop)
        opf.set_samples(s, leaf_count=st0.leaf_count)
        opf.pf.w_slow = opf.pf.w_fast = 1.0  # (0 / 0 in the reference: SURVEY.md Appendix A, 15)
        opf.set_resample_model(resampler)
        out = opf.update_resample()
        M = out.sample_count
        ok = (out.status == 0 and st1.last_status == 0 and st1.sample_count == M and st1.leaf_count == out.leaf_count and
              st1.bin_count == out.node_count and np.array_equal(after[:, :3], opf.samples[:M, :3]) and
              np.all(after[:, 3] == 1.0 / M) and pf.getRngState() == opf.pf.rng and st1.converged == out.converged)
        f = (e.kld_last_form(), st1.sample_count == max(max_s, n))
        forms[f] = forms.get(f, 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH case %d: n %d kind %s min %d max %d pop %s resampler %d seed %d: M %d/%d leaf %d/%d bins %d/%d status %d/%d"
                  % (case, n, kind, min_s, max_s, pop, resampler, rs, st1.sample_count, M, st1.leaf_count, out.leaf_count,
                     st1.bin_count, out.node_count, st1.last_status, out.status), flush=True)
        if case % 50 == 49:
            print("%d cases, %d mismatching, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
    print("%d cases, %d mismatching, %.0f s; (tree form, ran to max): %s" % (cases, bad, time.time() - t0, forms))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 200,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
