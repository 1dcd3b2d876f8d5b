"""Host-side wall time of each stage of the sharded step at world size 1 (diagnostic).
Pipelined: time the host spends IN each call while the GPU runs behind it."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist
import bench
class A: pass
args = A(); args.map_size=2000; args.beams=1081; args.particles=100000; args.cloud="converged"; args.model="lf"; args.resampler="multinomial"
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
wl = bench.build_workload(args, 0); wl["world"]=1
e, m, sc, pf, data, lut = bench.setup_engine(args, wl, 0)
from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
b = HipShardBackend(e, sc, pf, torch.device("cuda", 0)); sf = ShardedFilter(b, dist); counts = list(sf.counts)
import collections
T = collections.defaultdict(float)
def timed(name, fn, *a):
    t0 = time.perf_counter(); r = fn(*a); T[name] += time.perf_counter() - t0; return r
# instrument the backend / filter methods
for name in ["score", "normalize", "build_cdf", "draw_window", "kld_feed_window", "tail_small", "kld_reset", "kld_counts", "rng_state", "set_rng_state", "skip", "local_total"]:
    orig = getattr(b, name)
    setattr(b, name, (lambda o, n: (lambda *a: timed("b." + n, o, *a)))(orig, name))
for name in ["_all_gather", "_all_reduce_sum"]:
    orig = getattr(sf, name)
    setattr(sf, name, (lambda o, n: (lambda *a: timed("sf." + n, o, *a)))(orig, name))
def step():
    timed("restore", pf.restore); sf.restore(counts)
    timed("update_sensor", sf.update_sensor, data); timed("update_resample", sf.update_resample)
for _ in range(5): step()
torch.cuda.synchronize(); T.clear()
N = 100
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
print("step %.1f us" % ((time.perf_counter() - t0) / N * 1e6))
for k, v in sorted(T.items(), key=lambda kv: -kv[1]):
    print("  %-22s %7.1f us" % (k, v / N * 1e6))
dist.destroy_process_group()
