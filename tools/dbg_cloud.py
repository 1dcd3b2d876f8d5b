import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import badger_amcl_amd as bpf
from oracle import pyoracle as orc
import test_gpu_cloud as t
from scenario import rel_err
e = bpf.Engine(0)
for seed in (4, 5, 6):
    lut, pts, s, tf_xyz, tf_quat, max_dist = t._setup(orc, 2000, 8, 256, seed=seed)
    om = bpf.OctoMap(e, 0.05); om.setDistancesLUT(lut.pose_indices, lut.distance_ratios, lut.min_cells, lut.max_cells, max_dist)
    sc = bpf.PointCloudScanner(e); sc.init(128, om); sc.setPointCloudModel(0.5, 0.05, 0.1); sc.setMapFactors(0.95,0.95,0.3)
    sc.setPointCloudScannerToFootprintTF(tf_xyz, tf_quat)
    got = s.copy(); sc.applyModelToSampleSet(bpf.PointCloudData(pts), got)
    op = orc.cloud(orc.CLOUD_MODEL, 128, tf_xyz, tf_quat, z_hit=0.5, z_rand=0.05, sigma_hit=0.1); op.off_map_factor=0.95
    want = s.copy(); orc.cloud_apply(op, lut, want, pts)
    err = rel_err(got[:,3], want[:,3]); bad = np.flatnonzero(err > 1e-9)
    print("seed", seed, "points", pts.shape[0], "bad", bad, err[bad], "max ok err", err[err<=1e-9].max())
    for j in bad[:2]:
        # find which points differ: score particle j alone against single points
        diffs=[]
        for q in range(pts.shape[0]):
            g = s[j:j+1].copy(); sc.applyModelToSampleSet(bpf.PointCloudData(pts[q:q+1]), g)
            w = s[j:j+1].copy(); orc.cloud_apply(op, lut, w, pts[q:q+1])
            if abs(g[0,3]-w[0,3]) > 1e-12*abs(w[0,3]): diffs.append((q, g[0,3]/s[j,3], w[0,3]/s[j,3]))
        print(" particle", j, "pose", s[j,:3], "differing points", diffs[:5])
        for q,_,_ in diffs[:2]:
            import ctypes as C
            # oracle transform replicated in numpy float32
            print("  point", pts[q])
