"""Does the ORDER of a converged set matter to k_score_field?  Same particles, same scan, same launch; the set is
handed over in index order (as the bench does), sorted by heading, by heading bucket then position, and by map tile.
A host-side experiment: what a device-side sort could buy at most (the L1 holds 32 KB, a particle's 1 081 end points
touch ~116 tiles of 128 bytes, 16 waves per CU each work on their own particles)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import badger_amcl_amd as bpf
class A: pass
args = A(); args.map_size=2000; args.beams=1081; args.particles=100000; args.cloud=sys.argv[1] if len(sys.argv) > 1 else "converged"
args.model="lf"; args.resampler="multinomial"; args.lut="reference"; args.motion="none"; args.config=None; args.strong_total=None
wl = bench.build_workload(args, 0); wl["world"]=1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
e.set_option(2, 0)
e.set_option(bpf.pf.OPT_TILE_SORT, 0)
s0 = wl["samples"].copy()
ONLY = sys.argv[2] if len(sys.argv) > 2 else None   # substring of ONE order's label, six launches: for a rocprofv3 --pmc pass
def run(samples, label):
    pf.initWithSamples(samples); pf.snapshot()
    if ONLY is not None:
        if ONLY in label:
            for _ in range(6):
                pf.restore(); sc.updateSensor(pf, data)
            e.synchronize()
        return
    for _ in range(200):
        pf.restore(); sc.updateSensor(pf, data)
    e.synchronize(); e.profile_enable(3); e.profile_reset()
    for _ in range(100):
        pf.restore(); sc.updateSensor(pf, data)
    e.synchronize(); p = e.profile_get(); e.profile_enable(0)
    print("%-44s score kernel us: %.1f" % (label, p["score"]["ms"]/p["score"]["launches"]*1e3), flush=True)
run(s0, "index order (as generated)")
run(s0[np.argsort(s0[:, 2], kind="stable")], "sorted by heading")
b = np.floor(s0[:, 2] / 0.02).astype(np.int64)
run(s0[np.lexsort((s0[:, 0], b))], "heading buckets of 0.02 rad, then x")
tx = np.floor(s0[:, 0] / 0.4).astype(np.int64); ty = np.floor(s0[:, 1] / 0.4).astype(np.int64)
run(s0[np.lexsort((s0[:, 2], tx, ty))], "0.4 m tiles (y, x), then heading")
run(s0[np.lexsort((tx, ty, b))], "heading buckets, then 0.4 m tiles")
b1 = np.floor(s0[:, 2] / 0.01).astype(np.int64)
fx = np.floor(s0[:, 0] / 0.2).astype(np.int64); fy = np.floor(s0[:, 1] / 0.2).astype(np.int64)
run(s0[np.lexsort((fx, fy, b1))], "heading buckets of 0.01, then 0.2 m tiles")
run(s0[np.lexsort((b1, fx, fy))], "0.2 m tiles, then heading buckets of 0.01")
run(s0, "index order again")
