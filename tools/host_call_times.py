"""Host-side time of the three calls of a headline step (how long the host takes to enqueue; the GPU idles for
whatever part of that is not covered by kernels already running).  usage (GPU box): python3 tools/host_call_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
a = argparse.Namespace(config=None, model="lf", particles=100000, beams=1081, map_size=2000, cloud="converged",
                       resampler="multinomial")
wl = bench.build_workload(a, 0)
wl["world"] = 1
e, m, sc, pf, data, lut = bench.setup_engine(a, wl, 0)
for _ in range(100):
    pf.restore(); sc.updateSensor(pf, data); pf.updateResample()
e.synchronize()
T = [0.0, 0.0, 0.0, 0.0]
K = 300
t_all = time.perf_counter()
for _ in range(K):
    t0 = time.perf_counter(); pf.restore()
    t1 = time.perf_counter(); sc.updateSensor(pf, data)
    t2 = time.perf_counter(); pf.updateResample()
    t3 = time.perf_counter()
    T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2
e.synchronize()
tot = (time.perf_counter() - t_all) / K
print("per step %.1f us: restore call %.1f us, updateSensor call %.1f us, updateResample call (incl. wait) %.1f us"
      % (tot * 1e6, T[0] / K * 1e6, T[1] / K * 1e6, T[2] / K * 1e6))
