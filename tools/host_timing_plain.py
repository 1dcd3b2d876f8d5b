"""Host-side wall time of the calls of one unsharded step (diagnostic): how long the host is busy in each call while
the GPU runs behind it, and how long the step takes end to end."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gc
import bench
class A: pass
args = A(); args.map_size=2000; args.beams=1081; args.particles=100000
if len(sys.argv) > 3:  # particles beams map_size (configs[0]: 5000 181 400)
    args.particles, args.beams, args.map_size = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
args.cloud="converged"; args.model="lf"; args.resampler="multinomial"; args.lut="exact-edt"
wl = bench.build_workload(args, 0); wl["world"]=1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
for _ in range(10):
    pf.restore(); sc.updateSensor(pf, data); pf.updateResample()
e.synchronize(); gc.collect(); gc.freeze()
N = 300
T = [0.0, 0.0, 0.0]
t00 = time.perf_counter()
for _ in range(N):
    t0 = time.perf_counter(); pf.restore()
    t1 = time.perf_counter(); sc.updateSensor(pf, data)
    t2 = time.perf_counter(); pf.updateResample()
    t3 = time.perf_counter()
    T[0] += t1 - t0; T[1] += t2 - t1; T[2] += t3 - t2
e.synchronize()
print("step %.1f us: restore %.1f  updateSensor %.1f  updateResample %.1f (includes the wait for the keys)" %
      ((time.perf_counter() - t00) / N * 1e6, T[0] / N * 1e6, T[1] / N * 1e6, T[2] / N * 1e6))
# the same calls with the GPU idle in between: pure host cost of issuing a sensor update
T1 = 0.0
for _ in range(100):
    pf.restore(); e.synchronize()
    t1 = time.perf_counter(); sc.updateSensor(pf, data); T1 += time.perf_counter() - t1
    e.synchronize(); pf.updateResample(); e.synchronize()
print("updateSensor on an idle GPU: %.1f us of host time" % (T1 / 100 * 1e6))
