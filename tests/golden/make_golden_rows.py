#!/usr/bin/env python3
"""Generates tests/golden/rows_*.npz: fixed inputs and oracle outputs for the rows of SURVEY.md section 8 that
planar_*.npz / cycle_*.npz do not cover -- beam skipping (R6), the ray walk (R5 calcRange), the 3-D models (R17),
the five motion models (next-1) and the 2-D distance-LUT builder in reference order (next-3).  Like the other
fixtures they come from the CPU oracle (the reference cannot be built here): they pin the
oracle over time and give the HIP path targets that do not need the oracle in the loop.

    python tests/golden/make_golden_rows.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

BEAMSKIP = dict(do_beamskip=1, beam_skip_distance=0.5, beam_skip_threshold=0.3, beam_skip_error_threshold=0.9)
ODOM_ALPHA = (0.2, 0.15, 0.25, 0.1, 0.3)
ODOM_DATA = ((3.0, -1.0, 0.7), (0.21, -0.08, 0.12), (0.25, 0.09, 0.15))  # pose, delta, absolute motion
ODOM_RNG0 = 0x5A5A1234330E
GOMPERTZ_3D = dict(gompertz_a=0.748, gompertz_b=5.0, gompertz_c=1.2, input_shift=-3.2, input_scale=6.7,
                   output_shift=0.25)
CLOUD_TF = ((0.2, -0.1, 0.5), (0.0, 0.0, float(np.sin(0.15)), float(np.cos(0.15))))


def beamskip_case(orc, Scenario):
    sc = Scenario(orc, size=120, n=150, beams=60, cloud="converged", max_dist=1.0, seed=31, frac_max=0.0,
                  frac_nan=0.0)
    w = sc.samples.copy()
    total = sc.oracle_apply(sc.oracle_planar(60, "prob", BEAMSKIP), w, 1)
    return dict(cells=sc.cells.astype(np.int8), origin=np.array(sc.origin, dtype=np.float32), lut=sc.lut,
                max_dist=sc.max_dist, ranges=sc.ranges, angles=sc.angles, range_max=sc.range_max,
                samples=sc.samples, scanner_pose=np.array(sc.scanner_pose), map_factors=np.array(sc.map_factors),
                weights=w[:, 3], total=total)


def ray_fan(size=120):
    """Start points on and off the map, all octants, steep / shallow / axis-parallel, zero and huge max range."""
    rng = np.random.default_rng(9)
    ext = size * 0.05
    n = 160
    ox = rng.uniform(-0.3, ext + 0.3, n)
    oy = rng.uniform(-0.3, ext + 0.3, n)
    oa = rng.uniform(-np.pi, np.pi, n)
    oa[:8] = np.arange(8) * np.pi / 4           # the octant borders themselves
    oa[8:12] = [1e-9, np.pi / 2 - 1e-9, -1e-9, np.pi - 1e-9]
    mr = rng.choice([0.0, 0.04, 1.0, 5.0, 30.0], n)
    return ox, oy, oa, mr


def calc_range_case(orc, Scenario):
    sc = Scenario(orc, size=120, n=4, beams=11, max_dist=1.0, seed=31)
    ox, oy, oa, mr = ray_fan()
    out = np.array([sc.omap.calc_range(float(a), float(b), float(c), float(d)) for a, b, c, d in zip(ox, oy, oa, mr)])
    return dict(cells=sc.cells.astype(np.int8), origin=np.array(sc.origin, dtype=np.float32), ox=ox, oy=oy, oa=oa,
                max_range=mr, ranges=out)


def lut_case(orc):
    from badger_amcl_amd import synth
    cells, origin = synth.make_map(96)
    rng = np.random.default_rng(96)
    cells[rng.random(cells.shape) < 0.004] = 1   # scattered obstacles: many equidistant ties in the queue
    lut = orc.OccupancyMap(cells, 0.05, origin).update_distances_lut(0.7)
    return dict(cells=cells.astype(np.int8), origin=np.array(origin, dtype=np.float32), max_dist=0.7,
                lut=np.asarray(lut, dtype=np.float32))


def odom_case(orc):
    from badger_amcl_amd import synth
    s = synth.spread_cloud(300, 400, seed=12)
    s[:, 3] = np.random.default_rng(12).uniform(0.1, 1.0, 300)
    rec = dict(samples=s, alpha=np.array(ODOM_ALPHA), pose=np.array(ODOM_DATA[0]), delta=np.array(ODOM_DATA[1]),
               absolute_motion=np.array(ODOM_DATA[2]), rng_start=np.uint64(ODOM_RNG0))
    for model in range(5):
        w = s.copy()
        st = orc.odom_update_action(model, ODOM_ALPHA, *ODOM_DATA, w, ODOM_RNG0 ^ (model * 0x1111))
        rec["moved%d" % model] = w
        rec["rng_after%d" % model] = np.uint64(st)
    return rec


def cloud_inputs(orc):
    from badger_amcl_amd import synth
    res, max_dist = 0.05, 0.3
    occ = synth.box_room_voxels(lo=(-14, -10, -2), hi=(14, 10, 8))
    mn, mx = occ.min(axis=0) - 3, occ.max(axis=0) + 3
    lut = orc.OctoMapLUT(mn, mx, res, max_dist).build(occ)
    pts = synth.sphere_cloud(12, 96, (0.2, 0.05, 0.45), occ, res, seed=3).astype(np.float32)
    rng = np.random.default_rng(4)
    n = 120
    s = np.zeros((n, 4))
    s[:, 0] = 0.05 + rng.normal(0, 0.12, n)
    s[:, 1] = 0.02 + rng.normal(0, 0.12, n)
    s[:, 2] = rng.normal(0, 0.06, n)
    s[:12, 0] += 30.0  # off the map
    s[:, 3] = rng.uniform(0.5, 1.5, n) / n
    return lut, pts, s, max_dist


def cloud_case(orc):
    lut, pts, s, max_dist = cloud_inputs(orc)
    rec = dict(pose_indices=lut.pose_indices, distance_ratios=lut.distance_ratios,
               min_cells=np.asarray(lut.min_cells, dtype=np.int32), max_cells=np.asarray(lut.max_cells, dtype=np.int32),
               max_dist=max_dist, points=pts, samples=s, tf_xyz=np.array(CLOUD_TF[0]), tf_quat=np.array(CLOUD_TF[1]))
    for name, model, kw in (("plain", orc.CLOUD_MODEL, dict(z_hit=0.5, z_rand=0.05, sigma_hit=0.1)),
                            ("gompertz", orc.CLOUD_MODEL_GOMPERTZ, dict(z_hit=0.5, z_rand=0.5, sigma_hit=0.1,
                                                                        **GOMPERTZ_3D))):
        op = orc.cloud(model, 128, CLOUD_TF[0], CLOUD_TF[1], **kw)
        op.off_map_factor = 0.95
        w = s.copy()
        rec["total_" + name] = orc.cloud_apply(op, lut, w, pts)
        rec["weights_" + name] = w[:, 3]
    return rec


def main():
    from oracle import pyoracle as orc
    from scenario import Scenario
    for name, rec in (("beamskip", beamskip_case(orc, Scenario)), ("calc_range", calc_range_case(orc, Scenario)),
                      ("lut_reference", lut_case(orc)), ("odom", odom_case(orc)), ("cloud3d", cloud_case(orc))):
        np.savez_compressed(os.path.join(HERE, "rows_%s.npz" % name), **rec)
        print(name, {k: getattr(v, "shape", v) for k, v in list(rec.items())[:4]})


if __name__ == "__main__":
    main()
