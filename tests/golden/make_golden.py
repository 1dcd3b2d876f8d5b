#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs and the outputs of the CPU oracle
(oracle/amcl_oracle.c) for them.  The reference itself cannot be built or run in this image
(DESIGN.md section 2), so these vectors pin the ORACLE's behaviour over time and give the GPU
tests fixed targets that do not depend on running the oracle; the vectors that come from the
reference's own tests are the known answers in tests/test_oracle_pins.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    from oracle import pyoracle as orc
    from scenario import Scenario
    for model, max_beams in (("lf", 61), ("gompertz", 61), ("prob", 61), ("beam", 31)):
        sc = Scenario(orc, size=120, n=200, beams=61, cloud="mixture", max_dist=1.0, seed=17,
                      frac_nan=0.0 if model == "beam" else 0.02)  # the beam model does not skip NaN ranges
        w = sc.samples.copy()
        total = sc.oracle_apply(sc.oracle_planar(max_beams, model), w)
        opf = orc.ParticleFilter(20, 200, 0.0, 0.0, 85.0, seed=5)
        opf.set_population_size_parameters(0.05, 2.0)
        opf.set_samples(sc.samples)
        p = sc.oracle_planar(max_beams, model)
        opf.update_sensor(lambda s, c: sc.oracle_apply(p, s, c))
        normalized = opf.samples[:200].copy()
        w_slow = opf.pf.w_slow
        out = opf.update_resample()
        np.savez_compressed(
            os.path.join(HERE, "planar_%s.npz" % model),
            cells=sc.cells.astype(np.int8), origin=np.array(sc.origin, dtype=np.float32), lut=sc.lut,
            max_dist=sc.max_dist, ranges=sc.ranges, angles=sc.angles, range_max=sc.range_max, samples=sc.samples,
            scanner_pose=np.array(sc.scanner_pose), map_factors=np.array(sc.map_factors), max_beams=max_beams,
            weights_after_apply=w[:, 3], total=total, weights_normalized=normalized[:, 3], w_slow=w_slow,
            resampled=opf.samples[:out.sample_count], sample_count=out.sample_count, leaf_count=out.leaf_count,
            bin_count=out.node_count, rng_after=np.uint64(opf.pf.rng), converged=out.converged)
        print(model, "M", out.sample_count, "leaf", out.leaf_count)
    for resampler in (0, 1):
        make_cycle(orc, Scenario, resampler)


CYCLE_ALPHA = (0.001, 0.1)                                   # the node's default decay rates
CYCLE_ODOM = (2, (0.05, 0.04, 0.03, 0.02, 0.01))             # diff-corrected
CYCLE_ODATA = ((1.0, 2.0, 0.3), (0.03, -0.01, 0.02), (0.03, 0.01, 0.02))


def cycle_scans(sc):
    return [sc.ranges, np.clip(sc.ranges * 0.6, 0.05, 29.0)]


def run_cycle(orc, sc, resampler):
    """Two full cycles motion -> sensor -> resample (the second with w_diff > 0: random free-space poses) and
    the cluster statistics of the final set, on the oracle."""
    n = sc.samples.shape[0]
    opf = orc.ParticleFilter(50, n, CYCLE_ALPHA[0], CYCLE_ALPHA[1], 85.0, seed=77)
    opf.set_resample_model(resampler)
    opf.set_samples(sc.samples)
    opf.set_random_pose_source(sc.omap, sc.map_factors[2])
    p = sc.oracle_planar(61, "lf")
    rec = {}
    for c, ranges in enumerate(cycle_scans(sc)):
        cur = opf.samples[:opf.sample_count]
        opf.pf.rng = orc.odom_update_action(CYCLE_ODOM[0], CYCLE_ODOM[1], *CYCLE_ODATA, cur, opf.pf.rng)
        rec["moved%d" % c] = cur.copy()
        opf.update_sensor(lambda s, conv: orc.planar_apply(p, sc.omap, s, ranges, sc.angles, sc.range_max, conv))
        rec["weights%d" % c] = opf.samples[:opf.sample_count, 3].copy()
        out = opf.update_resample()
        M = out.sample_count
        rec["resampled%d" % c] = opf.samples[:M].copy()
        rec["scalars%d" % c] = np.array([M, out.leaf_count, out.node_count, out.cluster_count, out.converged],
                                        dtype=np.int64)
        rec["w_diff%d" % c] = out.w_diff
        rec["rng%d" % c] = np.uint64(opf.pf.rng)
        rec["mean%d" % c] = np.array(out.mean[:])
        rec["cov%d" % c] = np.array(list(out.cov[:]) + [out.cov_theta])
    return rec


def make_cycle(orc, Scenario, resampler):
    sc = Scenario(orc, size=120, n=600, beams=61, cloud="mixture", max_dist=1.0, seed=23)
    rec = run_cycle(orc, sc, resampler)
    assert rec["w_diff1"] > 0.01
    np.savez_compressed(os.path.join(HERE, "cycle_%s.npz" % ("multinomial", "systematic")[resampler]),
                        cells=sc.cells.astype(np.int8), origin=np.array(sc.origin, dtype=np.float32), lut=sc.lut,
                        max_dist=sc.max_dist, ranges=sc.ranges, angles=sc.angles, range_max=sc.range_max,
                        samples=sc.samples, scanner_pose=np.array(sc.scanner_pose),
                        map_factors=np.array(sc.map_factors), **rec)
    print("cycle", resampler, "M", rec["scalars0"][0], rec["scalars1"][0], "w_diff", rec["w_diff1"])


if __name__ == "__main__":
    main()
