"""Randomised differential soak of the 3-D point-cloud seam against the oracle: random set and cloud sizes (around the
2 048-point LDS chunk and the 64-lane wave), decimation, both models, planar and pitched mountings.
usage: python tools/soak_cloud.py [cases] [seed]"""
import os
import sys
import time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import badger_amcl_amd as bpf
from oracle import pyoracle as orc
from scenario import rel_err
from test_gpu_cloud import _setup

def run(cases=60, seed=1, e=None, quiet=False):
    """Returns the number of mismatching cases."""
    rng = np.random.default_rng(seed)
    own = e is None
    if own:
        e = bpf.Engine(0)
    t0 = time.time()
    bad = 0
    worst = 0.0
    for case in range(cases):
        n = int(rng.choice([1, 2, 63, 64, 65, 300, 1000]))
        rows, cols = int(rng.choice([1, 2, 8, 16])), int(rng.choice([3, 63, 64, 65, 256, 700]))
        pitched = bool(rng.integers(0, 2))
        model = str(rng.choice(["plain", "gompertz"]))
        max_beams = int(rng.choice([2, 17, 128, 100000]))
        lut, pts, s, tf_xyz, tf_quat, max_dist = _setup(orc, max(n, 10), rows, cols, seed=int(rng.integers(0, 1000)),
                                                        pitched=pitched)
        s = np.ascontiguousarray(s[:n])
        if pts.shape[0] < 2:
            continue
        om = bpf.OctoMap(e, 0.05)
        om.setDistancesLUT(lut.pose_indices, lut.distance_ratios, lut.min_cells, lut.max_cells, max_dist)
        sc = bpf.PointCloudScanner(e)
        sc.init(max_beams, om)
        gz = dict(gompertz_a=0.748, gompertz_b=5.0, gompertz_c=1.2, input_shift=-3.2, input_scale=6.7, output_shift=0.25)
        zh, zr, sg = float(rng.uniform(0.3, 0.9)), float(rng.uniform(0.02, 0.5)), float(rng.uniform(0.05, 0.3))
        if model == "plain":
            sc.setPointCloudModel(zh, zr, sg)
            op = orc.cloud(orc.CLOUD_MODEL, max_beams, tf_xyz, tf_quat, z_hit=zh, z_rand=zr, sigma_hit=sg)
        else:
            sc.setPointCloudModelGompertz(zh, zr, sg, gz["gompertz_a"], gz["gompertz_b"], gz["gompertz_c"],
                                          gz["input_shift"], gz["input_scale"], gz["output_shift"])
            op = orc.cloud(orc.CLOUD_MODEL_GOMPERTZ, max_beams, tf_xyz, tf_quat, z_hit=zh, z_rand=zr, sigma_hit=sg, **gz)
        f = float(rng.uniform(0.5, 1.0))
        sc.setMapFactors(f, 0.95, 0.3)
        op.off_map_factor = f
        sc.setPointCloudScannerToFootprintTF(tf_xyz, tf_quat)
        got = s.copy()
        total = sc.applyModelToSampleSet(bpf.PointCloudData(pts), got)
        want = s.copy()
        want_total = orc.cloud_apply(op, lut, want, pts)
        err = rel_err(got[:, 3], want[:, 3])
        nb = int((err > 1e-9).sum())
        ok = np.array_equal(got[:, :3], want[:, :3]) and nb <= 1 and (nb > 0 or abs(total - want_total) <= 1e-9 * abs(want_total))
        worst = max(worst, float(err[err <= 1e-9].max()) if (err <= 1e-9).any() else 0.0)
        if not ok:
            bad += 1
            print("MISMATCH case %d: n %d cloud %dx%d (%d points) max_beams %d %s pitched %s: %d weights off, total %g vs %g" %
                  (case, n, rows, cols, pts.shape[0], max_beams, model, pitched, nb, total, want_total), flush=True)
    print("%d cases, %d mismatching, worst rel err %.2e, %.0f s" % (cases, bad, worst, time.time() - t0))
    if own:
        e.close()
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 60,
                      int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
