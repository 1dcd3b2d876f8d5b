"""Wave lifetimes of k_cloud_score from a diagnostic build (-DBPF_PHASE_TIMING), see tools/phase_timing.py."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench


class A:
    pass


args = A(); args.map_size = 2000; args.beams = 1081; args.particles = 200000; args.cloud = "converged"
args.model = "cloud3d"; args.resampler = "multinomial"
wl = bench.build_workload(args, 0); wl["world"] = 1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
for _ in range(2):
    pf.restore(); sc.updateSensor(pf, data)
e.synchronize()
W = 6144
out = np.zeros((W, 2), dtype=np.uint64)
assert e.lib.bpf_debug_cloud_span(out.ctypes.data_as(C.c_void_p), W) == 0
t0, t1 = out[:, 0].astype(np.float64), out[:, 1].astype(np.float64)
live = t1 > 0
life = (t1 - t0) / 100.0
print("waves %d; span %.0f us; lifetime mean %.0f min %.0f max %.0f us; start skew %.1f us" %
      (live.sum(), (t1[live].max() - t0[live].min()) / 100.0, life[live].mean(), life[live].min(), life[live].max(),
       (t0[live].max() - t0[live].min()) / 100.0))
blk = np.arange(W) // 4
for r in range(6):
    sel = live & (blk // 256 == r)
    if sel.any():
        print("placement round %d: mean life %.0f us, last end %.0f us" % (r, life[sel].mean(), ((t1[sel] - t0[live].min()) / 100.0).max()))
