"""Where a wave of k_score_field spends its time, from a diagnostic build (-DBPF_PHASE_TIMING):

    mkdir -p build_exp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBPF_PHASE_TIMING \
          -o build_exp/libphase.so badger_amcl_amd/csrc/engine.hip
    BPF_LIB=$PWD/build_exp/libphase.so [GRADED=0] [BEAMS=1081] python tools/phase_timing.py

Prints the kernel's span, the distribution of wave lifetimes (by placement round of the block: the SIMD's issue
arbiter favours its oldest wave) and the share of each phase.
"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
import badger_amcl_amd.pf as hpf


class A:
    pass


args = A(); args.map_size = 2000; args.beams = int(os.environ.get("BEAMS", 1081)); args.particles = 100000
args.cloud = "converged"; args.model = "lf"; args.resampler = "multinomial"; args.lut = "exact-edt"
wl = bench.build_workload(args, 0); wl["world"] = 1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
e.set_option(hpf.OPT_GRADED_SHARES, int(os.environ.get("GRADED", "1")))
lib = e.lib
for _ in range(5):
    pf.restore(); sc.updateSensor(pf, data)
e.synchronize()
W = 4096
out = np.zeros((W, 8), dtype=np.uint64)
assert lib.bpf_debug_phase_cycles(out.ctypes.data_as(C.c_void_p), W) == 0
ph = out[:, :6].astype(np.float64)
t0, t1 = out[:, 6].astype(np.float64), out[:, 7].astype(np.float64)
live = t1 > 0
life = (t1 - t0) / 100.0
print("waves that ran: %d" % live.sum())
print("first start -> last end %.1f us; wave lifetime mean %.1f min %.1f max %.1f us; start skew %.1f us" %
      ((t1[live].max() - t0[live].min()) / 100.0, life[live].mean(), life[live].min(), life[live].max(),
       (t0[live].max() - t0[live].min()) / 100.0))
print("lifetime percentiles (us):", " ".join("%d:%.1f" % (q, np.percentile(life[live], q)) for q in (1, 10, 25, 50, 75, 90, 99)))
blk = np.arange(W) // 4
for r in range(4):
    sel = live & (blk // 256 == r)
    if sel.any():
        print("placement round %d (blocks %4d-%4d): mean life %.1f us, last end %.1f us" %
              (r, blk[sel].min(), blk[sel].max(), life[sel].mean(), ((t1[sel] - t0[live].min()) / 100.0).max()))
names = ["LDS staging + barrier", "batches (+ group start)", "remainder loop", "reduction", "epilogue", "block partial"]
tot = ph[live].sum()
for k, nm in enumerate(names):
    print("%-26s %9.0f core cycles per wave  %5.1f %%" % (nm, ph[live, k].mean(), 100.0 * ph[live, k].sum() / tot))
print("core clock %.0f MHz" % (ph[live].sum(axis=1).mean() / life[live].mean()))
# end times per placement round and per XCD (blocks b and b + 8 share one): what finer grading could still level
end = (t1 - t0[live].min()) / 100.0
for r in range(4):
    sel = live & (blk // 256 == r)
    if sel.any():
        print("round %d ends: p10 %.1f p50 %.1f p90 %.1f max %.1f" % (r, *[np.percentile(end[sel], q) for q in (10, 50, 90)], end[sel].max()))
print("ends by XCD:", " ".join("%d:%.1f/%.1f" % (x, np.median(end[live & (blk % 8 == x)]), end[live & (blk % 8 == x)].max()) for x in range(8)))
print("start by round:", " ".join("%d:%.1f" % (r, np.median((t0[live & (blk // 256 == r)] - t0[live].min()) / 100.0)) for r in range(4)))
