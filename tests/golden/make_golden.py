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


if __name__ == "__main__":
    main()
