import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
class A: pass
args = A(); args.map_size=2000; args.beams=1081; args.particles=100000; args.cloud="converged"; args.model="lf"; args.resampler="multinomial"
wl = bench.build_workload(args, 0); wl["world"]=1
r = wl["ranges"]; print("ranges percentiles m:", np.percentile(r, [5,25,50,75,95,100]))
print("per-chunk max range (m):", [round(float(r[i:i+64].max()),1) for i in range(0,1081,64)])
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
pf.restore(); sc.updateSensor(pf, data); e.synchronize()
print(e.window_plan())
