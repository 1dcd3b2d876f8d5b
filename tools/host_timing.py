"""Host-side wall time of each call in a bench step (diagnostic)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
class A: pass
args = A(); args.map_size=2000; args.beams=1081; args.particles=100000; args.cloud="converged"; args.model="lf"; args.resampler="multinomial"
wl = bench.build_workload(args, 0); wl["world"]=1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
for _ in range(5):
    pf.restore(); sc.updateSensor(pf, data); pf.updateResample()
e.synchronize()
T = {"restore":0,"sensor":0,"resample":0,"sync_after_sensor":0}
N=50
for _ in range(N):
    t0=time.perf_counter(); pf.restore(); t1=time.perf_counter(); sc.updateSensor(pf, data); t2=time.perf_counter()
    pf.updateResample(); t3=time.perf_counter()
    T["restore"]+=t1-t0; T["sensor"]+=t2-t1; T["resample"]+=t3-t2
e.synchronize()
print({k: round(v/N*1e6,1) for k,v in T.items()}, "us per call (pipelined)")
# now with a sync after the sensor update: how long does the GPU part of sensor take vs the resample alone
T = {"restore":0,"sensor+sync":0,"resample":0}
for _ in range(N):
    t0=time.perf_counter(); pf.restore(); t1=time.perf_counter(); sc.updateSensor(pf, data); e.synchronize(); t2=time.perf_counter()
    pf.updateResample(); t3=time.perf_counter()
    T["restore"]+=t1-t0; T["sensor+sync"]+=t2-t1; T["resample"]+=t3-t2
print({k: round(v/N*1e6,1) for k,v in T.items()}, "us per call (sync after sensor)")
