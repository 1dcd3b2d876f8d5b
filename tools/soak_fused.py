"""Soak of the fence-free hand-offs (k_normalize_cdf's look-back slots, k_resample_block's write-through draws; a
different scan every step, so that a stale staging block would show as well):
many update + resample cycles on varying sets, fused launches against the separate launches with the host replay,
bit for bit (weights after the update, set / counts / RNG state after the resample).  A stale read between blocks
would show as a differing CDF or draw.  usage (GPU box): python3 tools/soak_fused.py [cycles]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import badger_amcl_amd as bpf
import badger_amcl_amd.pf as hpf
from badger_amcl_amd import synth

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 400
size, beams = 2000, 181
cells, origin = synth.make_map(size)
pose = synth.true_pose(size)
scans = [bpf.PlanarData(*synth.cast_scan(cells, origin, 0.05, pose, beams, seed=5 + k), 30.0) for k in range(7)]
engines = []
for fused in (1, 0):
    e = bpf.Engine(0)
    e.set_option(hpf.OPT_FUSED_RESAMPLE, fused)
    m = bpf.OccupancyMap(e, 0.05)
    m.setCells(cells); m.setOrigin(origin); m.updateDistancesLUT(2.0)
    sc = bpf.PlanarScanner(e); sc.init(beams, m); sc.setModelLikelihoodField(0.95, 0.05, 0.2, 2.0)
    sc.setMapFactors(0.95, 0.95, 0.3); sc.setPlanarScannerPose((0.1, 0.0, 0.0))
    engines.append((e, sc))
rng = np.random.default_rng(7)
bad = 0
used = 0
t0 = time.time()
for c in range(cycles):
    n = int(rng.choice([1, 7, 130, 1000, 3000, 6000, 20000, 50000, 100000]))
    sig = float(rng.choice([0.05, 0.3, 1.0, 4.0]))
    s = synth.converged_cloud(n, pose, seed=1000 + c, sigma=(sig, sig, sig / 3))
    if c % 5 == 4 and n >= 1000:  # every fifth set: half of it anywhere on the map (off-map and in-wall poses included)
        s[: n // 2] = synth.spread_cloud(n // 2, size, 0.05, seed=2000 + c)
        s[:, 3] = 1.0 / n
    s[:, 3] *= rng.uniform(0.5, 1.5, n)
    resampler = int(rng.integers(0, 2))
    min_s = int(rng.choice([1, 10, 100, 500]))
    max_s = int(rng.choice([n, max(n, 2 * min_s), max(min_s + 1, n // 3 + 1)]))
    pop = [(0.01, 0.99), (0.05, 0.99), (0.0025, 0.9975)][int(rng.integers(0, 3))]
    out = []
    for e, sc in engines:
        pf = bpf.ParticleFilter(e, min(min_s, max_s), max(max_s, n), 0.0, 0.0, 85.0)
        pf.setResampleModel(resampler)
        pf.setPopulationSizeParameters(*pop)
        pf.srand48(c)
        pf.initWithSamples(s)
        rec = []
        for k in range(3):
            sc.updateSensor(pf, scans[(c + k) % 7])  # another scan every step: a stale staging block would show
            w = pf.getCurrentSet().samples[:, 3].copy()
            pf.updateResample()
            st = pf.getState()
            rec.append((w, pf.getCurrentSet().samples.copy(), st.sample_count, st.leaf_count, st.bin_count,
                        pf.getRngState(), st.converged, st.kld_on_device))
        out.append(rec)
    for k in range(3):
        a, b = out[0][k], out[1][k]
        same = (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:7] == b[2:7])
        used += a[7] == 2
        if not same:
            bad += 1
            print("MISMATCH cycle %d step %d n %d sigma %.2f resampler %d min %d max %d pop %r: M %d/%d leaf %d/%d" % (
                c, k, n, sig, resampler, min_s, max_s, pop, a[2], b[2], a[3], b[3]))
print("%d cycles x 3 steps, %d through the single-launch resample, %d mismatches, %.0f s" % (
    cycles, used, bad, time.time() - t0))
sys.exit(1 if bad else 0)
