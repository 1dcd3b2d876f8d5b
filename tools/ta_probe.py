"""How much of k_score_field's time is gather locality?  Same kernel, three scans:
real scan, all beams identical (every lane hits the same cell), beams sorted by bearing with
constant range (smooth arc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import badger_amcl_amd as bpf
class A: pass
args = A(); args.map_size=2000; args.beams=1081; args.particles=100000; args.cloud="converged"; args.model="lf"; args.resampler="multinomial"; args.lut="reference"; args.motion="none"; args.config=None; args.strong_total=None
wl = bench.build_workload(args, 0); wl["world"]=1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
e.set_option(2, 0)
ONLY = sys.argv[1] if len(sys.argv) > 1 else None   # one scan only, few launches: for a rocprofv3 --pmc pass (tools/exp/ta_pmc.sh)
def run(d, label, key):
    if ONLY is not None:
        if ONLY == key:
            for _ in range(6):
                pf.restore(); sc.updateSensor(pf, d)
            e.synchronize()
        return
    for _ in range(300):
        pf.restore(); sc.updateSensor(pf, d)
    e.synchronize(); e.profile_enable(3); e.profile_reset()
    for _ in range(100):
        pf.restore(); sc.updateSensor(pf, d)
    e.synchronize(); p = e.profile_get(); e.profile_enable(0)
    print(label, "score kernel us:", round(p["score"]["ms"]/p["score"]["launches"]*1e3,1))
run(data, "real scan", "real")
same = bpf.PlanarData(np.full(1081, 4.0), np.full(1081, 0.3), 30.0)
run(same, "all beams identical", "same")
arc = bpf.PlanarData(np.full(1081, 4.0), wl["angles"], 30.0)
run(arc, "constant range arc", "arc")
short = bpf.PlanarData(np.full(1081, 0.5), wl["angles"], 30.0)
run(short, "0.5 m arc", "short")
