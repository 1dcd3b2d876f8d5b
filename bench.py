#!/usr/bin/env python3
"""Headline benchmark: particle-beam evaluations per second over (sensor update + resample).

    python bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over the resident particle set:
    restore the set (device-to-device, stands in for the motion update that rewrites every pose)
    -> PlanarScanner::updateSensor (scoring + recalcWeight + normalise + running averages)
    -> ParticleFilter::updateResample (CDF, drand48 draws, selection, KLD stop, weights 1/M).
Workload at N=1: BASELINE.json configs[1] -- 2-D likelihood field, 100 000 particles x 1081
beams, 2000x2000 map, converged cloud; inputs are resident in HBM before the
timed region.  For N>1 the driver launches one rank per GPU through torch.distributed.run;
each rank holds a 100 000-particle shard of one filter (weak scaling).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline`.
When the run is the plain headline (no --config / --model / size flag) the same line also carries
  other_configs   N=1: short timed regions of BASELINE configs[2] (beam model), configs[4] (3-D) and
                  configs[0] (the reference's own CPU case), each with its own roofline + cpu_baseline
  strong_scaling  BASELINE configs[3] as worded: 1 000 000 particles TOTAL, 1 000 000 / N per GPU
  exchange_ms     N>1: the sharded step timed once per exchange (mailbox, then torch.distributed = RCCL)
so that one driver command times every BASELINE configuration.
"""
import argparse
import copy
import gc
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6290.0  # same guide: measured streaming copy

# oracle's single-thread pace (evaluations per second) used only to SIZE the CPU sample for a budget
CPU_PACE = {"lf": 3.5e7, "gompertz": 3.0e7, "beam": 5.0e6, "cloud3d": 1.5e7}


class Stage:
    """A named stage of the run with a bound: when it is exceeded, ONE stderr line names the stage and the process
    ends with a non-zero code -- a hung bring-up (rendez-vous, mailbox mapping, a collective that never completes)
    must not sit there until the driver's own limit."""
    current = "start"

    def __init__(self, name, limit_s):
        self.name, self.limit = name, float(limit_s)
        self.timer = None

    def _expired(self):
        try:
            sys.stderr.write("bench.py: stage '%s' exceeded %.0f s on rank %s -- giving up\n" % (
                self.name, self.limit, os.environ.get("RANK", "0")))
            sys.stderr.flush()
        finally:
            os._exit(124)

    def __enter__(self):
        Stage.current = self.name
        self.timer = threading.Timer(self.limit, self._expired)
        self.timer.daemon = True
        self.timer.start()
        return self

    def __exit__(self, et, ev, tb):
        self.timer.cancel()
        if et is not None and et is not SystemExit:
            sys.stderr.write("bench.py: stage '%s' failed on rank %s: %s: %s\n" % (
                self.name, os.environ.get("RANK", "0"), getattr(et, "__name__", et), ev))
            sys.stderr.flush()
        return False


def algorithmic_bytes(model, n, beams, mean_cells=None):
    """LF family 4 B/eval + 40 B/particle + 16 B/beam; beam model C x 1 B/eval;
    3-D 5 B/eval + 12 B/point + 40 B/particle."""
    if model == "cloud3d":
        return 5.0 * n * beams + 12.0 * beams + 40.0 * n
    per_eval = 4.0 if model != "beam" else float(mean_cells or 0.0)
    return per_eval * n * beams + 40.0 * n + 16.0 * beams


def build_workload(args, rank):
    from badger_amcl_amd import synth
    if args.model == "cloud3d":
        # BASELINE.json configs[4]: 3-D likelihood field, 64 x 1024-point cloud
        pi, dr, mn, mx = synth.box_room_lut()
        pts = synth.grid_cloud(64, 1024)
        n = args.particles
        samples = synth.converged_cloud(n, np.array([0.0, 0.0, 0.0]), seed=42 + rank, sigma=(0.15, 0.15, 0.05))
        return dict(lut3=(pi, dr, mn, mx), points=pts, samples=samples, n=n, beams=pts.shape[0],
                    tf_xyz=(0.3, 0.2, 0.6), tf_quat=(0.0, 0.0, 0.0, 1.0))
    size, beams, n = args.map_size, args.beams, args.particles
    cells, origin = synth.make_map(size)
    pose = synth.true_pose(size)
    ranges, angles = synth.cast_scan(cells, origin, 0.05, pose, beams, seed=5)
    if args.cloud == "converged":
        samples = synth.converged_cloud(n, pose, seed=42 + rank)
    else:
        samples = synth.spread_cloud(n, size, seed=43 + rank, margin=0.5)
    return dict(cells=cells, origin=origin, pose=pose, ranges=ranges, angles=angles, samples=samples,
                size=size, beams=beams, n=n)


LUT_NAMES = {
    "reference": "reference brushfire, host builder (OccupancyMap::updateDistancesLUT, occupancy_map.cpp:138-252)",
    "exact-edt": "exact capped Euclidean distance transform, device builder (updateDistancesLUTExact)",
}


def setup_engine(args, wl, device):
    import badger_amcl_amd as bpf
    from badger_amcl_amd import synth
    e = bpf.Engine(device)
    if args.model == "cloud3d":
        pi, dr, mn, mx = wl["lut3"]
        m = bpf.OctoMap(e, 0.05)
        m.setDistancesLUT(pi, dr, mn, mx, 0.3)
        sc = bpf.PointCloudScanner(e)
        sc.init(wl["beams"], m)
        sc.setPointCloudModel(0.5, 0.05, 0.1)
        sc.setMapFactors(*synth.MAP_FACTORS)
        sc.setPointCloudScannerToFootprintTF(wl["tf_xyz"], wl["tf_quat"])
        pf = bpf.ParticleFilter(e, 100, wl.get("n_global", wl["n"]), 0.0, 0.0, 85.0)
        pf.srand48(42)
        pf.initWithSamples(wl["samples"])
        pf.snapshot()
        return e, m, sc, pf, bpf.PointCloudData(wl["points"]), None
    m = bpf.OccupancyMap(e, 0.05)
    m.setCells(wl["cells"])
    m.setOrigin(wl["origin"])
    if wl.get("lut") is not None:
        m.setDistancesLUT(wl["lut"], 2.0)  # a LUT an earlier record of this run already built for the same map
    elif args.lut == "reference":
        m.updateDistancesLUT(2.0)
    else:
        m.updateDistancesLUTExact(2.0)
    sc = bpf.PlanarScanner(e)
    sc.init(wl["beams"], m)
    if args.model == "lf":
        p = synth.LF_DEFAULTS
        sc.setModelLikelihoodField(p["z_hit"], p["z_rand"], p["sigma_hit"], 2.0)
    elif args.model == "gompertz":
        p = synth.GOMPERTZ_LAUNCH
        sc.setModelLikelihoodFieldGompertz(p["z_hit"], p["z_rand"], p["sigma_hit"], 2.0, p["gompertz_a"],
                                           p["gompertz_b"], p["gompertz_c"], p["input_shift"], p["input_scale"],
                                           p["output_shift"])
    else:
        p = synth.BEAM_DEFAULTS
        sc.setModelBeam(p["z_hit"], p["z_short"], p["z_max"], p["z_rand"], p["sigma_hit"], p["lambda_short"])
    sc.setMapFactors(*synth.MAP_FACTORS)
    sc.setPlanarScannerPose(synth.SCANNER_POSE)
    # sharded: every engine carries the GLOBAL min / max sample counts (the KLD bound is global)
    pf = bpf.ParticleFilter(e, 100, wl.get("n_global", wl["n"]), 0.0, 0.0, 85.0)
    pf.setResampleModel(1 if args.resampler == "systematic" else 0)
    pf.srand48(42)
    pf.initWithSamples(wl["samples"])
    pf.snapshot()
    data = bpf.PlanarData(wl["ranges"], wl["angles"], 30.0)
    lut = m.getDistancesLUT()
    return e, m, sc, pf, data, lut


def baseline_metric():
    """BASELINE.json's metric string, verbatim."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "particle-beam evals/sec (sensor update+resample), 100k particles×1081 beams"


def cpu_sample_particles(model, n, beams, budget_s):
    """Particles of the CPU sample: the whole set when one step of it fits the budget, otherwise the prefix the
    oracle's pace gets through in about the budget (a BOUNDED sample of the same workload)."""
    fit = int(budget_s * CPU_PACE[model] / max(1, beams))
    return max(min(n, 100), min(n, fit))


def cpu_baseline(args, wl, lut, budget_s):
    """The oracle (a port of the reference's CPU path) timed on this box's host cores: 1 thread,
    same workload, as many whole steps as fit the budget (at least one)."""
    from badger_amcl_amd import synth
    from oracle import pyoracle as orc
    n_cpu = cpu_sample_particles(args.model, wl["n"], wl["beams"], budget_s)
    if args.model == "cloud3d":
        pi, dr, mn, mx = wl["lut3"]
        olut = orc.OctoMapLUT(mn, mx, 0.05, 0.3, pi, dr)
        op = orc.cloud(orc.CLOUD_MODEL, wl["beams"], wl["tf_xyz"], wl["tf_quat"], z_hit=0.5, z_rand=0.05,
                       sigma_hit=0.1)
        op.off_map_factor = synth.MAP_FACTORS[0]
        opf = orc.ParticleFilter(100, n_cpu, 0.0, 0.0, 85.0, seed=42)
        opf.set_samples(wl["samples"][:n_cpu], leaf_count=0)
        t0 = time.perf_counter()
        opf.update_sensor(lambda s, conv: orc.cloud_apply(op, olut, s, wl["points"]))
        out = opf.update_resample()
        dt = time.perf_counter() - t0
        return dict(value=float(n_cpu) * wl["beams"] / dt, unit="particle-beam evals/s", cores=1, kind="port",
                    sample="1 step of %d particles x %d points (3-D cloud model), %.1f s, oracle gcc -O2 single "
                           "thread; resampled to M=%d" % (n_cpu, wl["beams"], dt, out.sample_count)), None, opf
    omap = orc.OccupancyMap(wl["cells"], 0.05, wl["origin"], 2.0, lut)
    if args.model == "lf":
        p = orc.planar(orc.MODEL_LF, wl["beams"], scanner_pose=synth.SCANNER_POSE, **synth.LF_DEFAULTS)
    elif args.model == "gompertz":
        p = orc.planar(orc.MODEL_LF_GOMPERTZ, wl["beams"], scanner_pose=synth.SCANNER_POSE, **synth.GOMPERTZ_LAUNCH)
    else:
        p = orc.planar(orc.MODEL_BEAM, wl["beams"], scanner_pose=synth.SCANNER_POSE, **synth.BEAM_DEFAULTS)
    p.off_map_factor, p.non_free_space_factor, p.non_free_space_radius = synth.MAP_FACTORS
    steps, t_used, stats = 0, 0.0, {}
    last = None
    while steps == 0 or (t_used < budget_s and t_used / steps * (steps + 1) < budget_s * 1.5):
        opf = orc.ParticleFilter(100, n_cpu, 0.0, 0.0, 85.0, seed=42)
        opf.set_resample_model(1 if args.resampler == "systematic" else 0)
        opf.set_samples(wl["samples"][:n_cpu], leaf_count=0)
        t0 = time.perf_counter()
        opf.update_sensor(lambda s, conv: orc.planar_apply(p, omap, s, wl["ranges"], wl["angles"], 30.0, conv, stats))
        out = opf.update_resample()
        t_used += time.perf_counter() - t0
        steps += 1
        last = out
    evals = float(n_cpu) * wl["beams"] * steps
    mean_cells = (stats.get("cells", 0) / stats["evals"]) if stats.get("evals") else None
    return dict(value=evals / t_used, unit="particle-beam evals/s", cores=1, kind="port",
                sample="%d step(s) of %d particles x %d beams (%s, %s cloud, %s resampling), %.1f s, oracle "
                       "gcc -O2 single thread; resampled to M=%d" % (steps, n_cpu, wl["beams"], args.model, args.cloud,
                                                                     args.resampler, t_used, last.sample_count)), \
        mean_cells, opf


def cpu_baseline_all_cores(args, wl, lut, budget_s):
    """SURVEY.md section 8(d)'s generous figure: the same oracle with the scoring loop sharded over every host
    core of a one-GPU job's share (at most 16; contiguous particle ranges, one thread each; ctypes releases the GIL),
    then the serial normalisation and resampler.  One whole step of a budgeted prefix of the set."""
    from badger_amcl_amd import synth
    from oracle import pyoracle as orc
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 16))  # a one-GPU job's share of the host (the box runs one job per GPU)
    n = cpu_sample_particles(args.model, wl["n"], wl["beams"], budget_s * cores)
    if args.model == "cloud3d":
        pi, dr, mn, mx = wl["lut3"]
        olut = orc.OctoMapLUT(mn, mx, 0.05, 0.3, pi, dr)

        def score(view):
            op = orc.cloud(orc.CLOUD_MODEL, wl["beams"], wl["tf_xyz"], wl["tf_quat"], z_hit=0.5, z_rand=0.05,
                           sigma_hit=0.1)
            op.off_map_factor = synth.MAP_FACTORS[0]
            return orc.cloud_apply(op, olut, view, wl["points"])
    else:
        omap = orc.OccupancyMap(wl["cells"], 0.05, wl["origin"], 2.0, lut)
        kw = {"lf": synth.LF_DEFAULTS, "gompertz": synth.GOMPERTZ_LAUNCH, "beam": synth.BEAM_DEFAULTS}[args.model]
        mid = {"lf": orc.MODEL_LF, "gompertz": orc.MODEL_LF_GOMPERTZ, "beam": orc.MODEL_BEAM}[args.model]

        def score(view):
            p = orc.planar(mid, wl["beams"], scanner_pose=synth.SCANNER_POSE, **kw)
            p.off_map_factor, p.non_free_space_factor, p.non_free_space_radius = synth.MAP_FACTORS
            return orc.planar_apply(p, omap, view, wl["ranges"], wl["angles"], 30.0, 0, None)
    opf = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0, seed=42)
    opf.set_resample_model(1 if args.resampler == "systematic" else 0)
    opf.set_samples(wl["samples"][:n], leaf_count=0)
    bounds = [(n * t) // cores for t in range(cores + 1)]
    totals = [0.0] * cores

    def work(t, view):
        totals[t] = score(view)

    def sensor(samples, conv):
        th = [threading.Thread(target=work, args=(t, samples[bounds[t]:bounds[t + 1]])) for t in range(cores)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        total = 0.0
        for v in totals:
            total += v
        return total

    t0 = time.perf_counter()
    opf.update_sensor(sensor)
    out = opf.update_resample()
    dt = time.perf_counter() - t0
    return dict(value=float(n) * wl["beams"] / dt, unit="particle-beam evals/s", cores=cores, kind="port",
                sample="1 step of %d particles x %d beams (%s), scoring on %d threads (contiguous particle ranges), "
                       "serial normalise + resample, %.2f s; resampled to M=%d" % (n, wl["beams"], args.model, cores, dt,
                                                                                   out.sample_count))


def apply_config_preset(args):
    """--config k = BASELINE.json configs[k-1]; explicit flags given with it still win where they differ from the
    defaults."""
    if args.config is None:
        return
    if args.config == 1:
        args.model, args.beams, args.map_size = "lf", 181, 400
        args.particles = args.particles or 5000
    elif args.config == 2:
        args.model = "lf"
    elif args.config == 3:
        args.model = "beam"
    elif args.config == 4:
        args.model = "lf"
        args.particles = args.particles or 125000
    elif args.config == 5:
        args.model = "cloud3d"


def self_launch(n):
    """Start n ranks of this script under torch.distributed.run (one per GPU, rendez-vous on 127.0.0.1) from a parent
    that never touches the GPU; pass rank 0's JSON line through; return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in proc.stdout:
        txt = raw.decode(errors="replace")
        if txt.lstrip().startswith('{"metric"'):
            line = txt.strip()
        else:
            sys.stderr.write(txt)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        rc = 3
    return rc


STRONG_TOTAL = 1000000  # BASELINE.json configs[3]: 1M particles sharded 8 x MI355X


def workload_name(args, world):
    """config.workload: BASELINE.json's own wording where the run IS one of its configs."""
    key = (args.model, args.particles, args.beams, args.map_size)
    if getattr(args, "strong_total", None):
        return ("2D likelihood-field, 1M particles sharded %dxMI355X, RCCL weight/KLD all-reduce over xGMI (configs[3] as "
                "worded: %d particles TOTAL, %d per GPU, 1081 beams, 2000x2000 map)" % (
                    world, args.strong_total, args.strong_total // world))
    if key == ("lf", 100000, 1081, 2000):
        if world > 1:
            return ("2D likelihood-field, 100k particles per GPU, 1081 beams, 2000x2000 map: ONE filter of %d particles "
                    "sharded over %d GPUs (configs[3]'s layout at configs[1]'s per-GPU size)" % (100000 * world, world))
        return "2D likelihood-field, 100k particles, 1081 beams, 2000x2000 map"
    if key == ("lf", 5000, 181, 400):
        return "2D likelihood-field, 5000 particles, 181-beam scan, 400x400 static map (configs[0]: the reference's CPU case)"
    if key == ("beam", 100000, 1081, 2000):
        return "2D beam-model raycast, 100k particles, 1081 beams, 2000x2000 map" + (
            "" if world == 1 else " -- per GPU: ONE filter of %d particles sharded over %d GPUs" % (100000 * world, world))
    if key == ("lf", 125000, 1081, 2000):
        return ("2D likelihood-field, 125k particles per GPU (configs[3]: 1M particles sharded over 8 GPUs; this run: "
                "%d GPU%s, %d particles)" % (world, "" if world == 1 else "s", 125000 * world))
    if args.model == "cloud3d":
        return "3D (octomap) likelihood-field, %s particles, 64x1024-point cloud" % (
            "200k" if args.particles == 200000 else str(args.particles)) + (
            "" if world == 1 else " -- per GPU: ONE filter of %d particles sharded over %d GPUs" % (
                args.particles * world, world))
    return "2D %s, %d particles/GPU, %d beams, %dx%d map" % (args.model, args.particles, args.beams, args.map_size,
                                                           args.map_size)


def metric_name(args):
    """BASELINE.json's metric verbatim for the headline shape; the same quantity named for the shape that ran
    otherwise (a beam-model or 3-D line must not carry the headline's label)."""
    if (args.model, args.particles, args.beams) in (("lf", 100000, 1081), ("gompertz", 100000, 1081)):
        return baseline_metric()
    what = {"lf": "likelihood field", "gompertz": "likelihood field (Gompertz)", "beam": "beam-model raycast",
            "cloud3d": "3-D point-cloud model"}[args.model]
    beams = 65536 if args.model == "cloud3d" else args.beams
    return "particle-beam evals/sec (sensor update+resample), %s, %d particles/GPU x %d %s" % (
        what, args.particles, beams, "points" if args.model == "cloud3d" else "beams")


def pmc_evidence(model, cloud, k_ms, n, beams):
    """What the committed rocprofv3 PMC passes say binds the scoring kernel (profiles/pmc_traffic.json, written by
    tools/pmc_score.sh; static: not re-measured in this run, tagged with its source).  Without a record for this very
    workload nothing is claimed: `bound` is None (= unmeasured), not a guess."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    out = {"bound": None, "traffic": None, "traffic_source": None}
    try:
        recs = json.load(open(path))
    except Exception:
        return out
    rec = None
    for key in ("%s_%s_%d" % (model, cloud, n), "%s_%s" % (model, cloud), model):
        cand = recs.get(key)
        if cand and (cand.get("particles"), cand.get("beams")) == (n, beams):
            rec = cand
            break
    if rec is None:
        return out  # counters of another workload say nothing about this one
    out["traffic"] = rec.get("hbm_bytes_per_launch")
    out["traffic_source"] = "static: %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run)" % rec.get(
        "source", "profiles/pmc_traffic.json")
    if rec.get("hbm_bytes_per_launch") and k_ms > 0:
        out["hbm_measured_gbs"] = rec["hbm_bytes_per_launch"] / (k_ms * 1e-3) / 1e9
    if rec.get("issue_frac") is not None:
        out["issue_frac"] = rec["issue_frac"]  # SQ_ACTIVE_INST_VALU x 4 cycles / SIMDs / kernel cycles, that run
        out["bound"] = rec.get("bound", "valu_issue")
        if rec.get("bound_note"):
            out["bound_is"] = rec["bound_note"]
    if out.get("hbm_measured_gbs", 0.0) >= 0.5 * HBM_PEAK_GBS:
        # the L2s fetch at more than half the HBM peak (FETCH_SIZE counts every L2 fill, Infinity-Cache hits included):
        # the spread cloud, whose particles put the whole LUT through every XCD's 4 MB L2
        out["bound"] = "hbm"
        out["bound_is"] = ("L2 fills (2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included) above half the HBM "
                           "peak: the kernel waits for LUT lines, not for issue slots")
    return out


# ---------------------------------------------------------------------------------------------- one timed record
class Ctx:
    """What every record of a run shares: the ranks, the process group, the host-side meeting point."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.torch = None
        self.host_group = None  # gloo group for waits that must not sit in a GPU collective (rank 0's CPU leg)
        self.rehearsal = os.environ.get("BPF_BENCH_REHEARSAL") == "1"
        self.lut_cache = {}

    def host_barrier(self, limit_s):
        """All ranks meet on the HOST (gloo): the others are parked here while rank 0 times the CPU baseline."""
        if self.dist is None:
            return
        with Stage("host barrier (ranks parked while rank 0 times the CPU baseline)", limit_s):
            if self.host_group is not None:
                self.dist.barrier(group=self.host_group)
            else:
                self.dist.barrier()


def measure(args, ctx, steps, warmup, prewarm, cpu_budget, all_cores=False, host_path=False, profile_mode=1,
            exchange=None, compare_exchanges=False, prewarm_s=0.25, repeat=False):
    """One engine, one workload: W warm-up steps, K timed steps between fences, the scoring kernel's HIP events, the
    CPU baseline.  Returns the record (rank 0) or None (other ranks)."""
    torch, dist, world, rank = ctx.torch, ctx.dist, ctx.world, ctx.rank
    wl = build_workload(args, rank)
    n_global = wl["n"] * world
    if getattr(args, "strong_total", None):
        n_global = args.strong_total
    wl["n_global"] = n_global
    lut_key = (args.map_size, args.lut)
    if args.model != "cloud3d":
        wl["lut"] = ctx.lut_cache.get(lut_key)
    t_setup = time.perf_counter()
    with Stage("engine set-up (%s)" % args.model, 300):
        e, m, sc, pf, data, lut = setup_engine(args, wl, ctx.local_rank)
    t_setup = time.perf_counter() - t_setup
    if lut is not None:
        ctx.lut_cache[lut_key] = lut
    if os.environ.get("BPF_BENCH_FUSED") == "0":
        e.set_option(5, 0)  # BPF_OPT_FUSED_RESAMPLE off: the separate normalise / scan / draw / tail launches (A/B runs)

    odom = odata = None
    if args.motion != "none":
        import badger_amcl_amd as bpf
        odom = bpf.Odom(e)
        odom.setModel(["diff", "omni", "diff-corrected", "omni-corrected", "gaussian"].index(args.motion),
                      0.05, 0.05, 0.05, 0.05, 0.05)
        odata = bpf.OdomData((1.0, 2.0, 0.3), (0.02, 0.005, 0.01))

    sf = None
    selftest = None
    if dist is not None:
        from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
        with Stage("sharded filter bring-up (mailbox create / IPC map / connect round / self-test)", 180):
            backend = HipShardBackend(e, sc, pf, torch.device("cuda", ctx.local_rank))
            # BPF_SHARD_EXCHANGE=collective forces the RCCL all-gather / all-reduce; default: mailbox when every rank can
            sf = ShardedFilter(backend, dist, exchange=exchange or os.environ.get("BPF_SHARD_EXCHANGE", "auto"))
        selftest = sf.mailbox_verdict
        shard_counts = list(sf.counts)
        shard_leaf = sf.leaf_count  # global leaf count of the initial set (systematic resampler)

        def step():
            pf.restore()
            sf.restore(shard_counts, shard_leaf)
            if odom is not None:
                sf.update_action(odom, odata)
            sf.update_sensor(data)
            sf.update_resample()
    else:
        def step():
            pf.restore()
            if odom is not None:
                odom.updateAction(pf, odata)
            sc.updateSensor(pf, data)
            pf.updateResample()

    def fence():
        e.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        e.synchronize()
        torch.cuda.synchronize()

    def timed(k):
        """K steps between two fences; MAX over the ranks."""
        fence()
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # The interpreter's full garbage collection (45-60 ms) must not land inside the timed region and a pause of that
    # length right in front of it lets the GPU's clocks fall: main() collects and freezes before the first engine.
    # The event pool is created (16 384 hipEventCreate, tens of ms of host time) BEFORE the warm-up: a pause of that
    # length between the warm-up and the timed region lets the GPU's clocks fall, and a short timed region (the
    # driver's 20 steps) then reads 0.147 ms per step where 300 steps read 0.123.
    e.profile_enable(profile_mode)  # HIP events attached to the scoring dispatch only
    with Stage("warm-up (%s)" % args.model, 300):
        pw_done = 0
        if prewarm > 0:
            # every rank must run the same number of steps: the count comes from a first chunk timed with the
            # MAX over the ranks (the all-reduced value is the same bits everywhere)
            k0 = max(1, min(prewarm, 20))
            per_step = timed(k0) / k0
            pw_done = k0
            more = max(prewarm - k0, int(prewarm_s / max(per_step, 1e-6)) + 1 - k0)
            for _ in range(max(0, min(more, 20000))):
                step()
            pw_done += max(0, min(more, 20000))
        prewarm = pw_done
        for _ in range(warmup):
            step()
        fence()
    e.profile_reset()
    with Stage("timed region (%s)" % args.model, 600):
        dt = timed(steps)
    prof = e.profile_get()
    # sub-records: the same K steps once more (ms_per_step_repeat), so that a one-off stall of the shared host inside a
    # region of a few milliseconds shows as what it is; the record's figures come from the first region
    dt_repeat = None
    if repeat:
        e.profile_enable(0)
        with Stage("timed region, repeat (%s)" % args.model, 600):
            dt_repeat = timed(steps)
    # untimed extra pass with every kernel class bracketed, for the per-kernel breakdown
    e.profile_enable(2)
    e.profile_reset()
    n_all = max(1, min(steps, 10))
    for _ in range(n_all):
        step()
    fence()
    prof_all = e.profile_get()
    e.profile_enable(0)
    st = pf.getState() if dist is None else sf.state()
    ran_with_mailbox = bool(sf is not None and sf.mailbox)  # the exchange `value` was measured with

    exchange_ms = None
    if compare_exchanges and sf is not None:
        # the same sharded step once per exchange, in this one run: what ran above, then the other one
        first = "mailbox" if sf.mailbox else "collective"
        exchange_ms = {"mailbox": None, "collective": None}
        exchange_ms[first] = dt / steps * 1e3
        with Stage("second exchange (%s -> %s)" % (first, "collective" if sf.mailbox else "mailbox"), 300):
            if sf.mailbox:
                sf.use_collectives()
                other = "collective"
            else:
                other = "mailbox" if sf.try_mailbox() else None
            if other is not None:
                for _ in range(max(5, warmup)):
                    step()
                exchange_ms[other] = timed(steps) / steps * 1e3

    n_total = n_global if getattr(args, "strong_total", None) else wl["n"] * world
    evals_per_step = float(n_total) * wl["beams"]
    value = evals_per_step * steps / dt

    cpu, mean_cells, cpu_all, host = None, None, None, None
    if rank == 0:
        if cpu_budget > 0:
            with Stage("cpu baseline (%s)" % args.model, cpu_budget * 6 + 120):
                cpu, mean_cells, _ = cpu_baseline(args, wl, lut, cpu_budget)
                if all_cores:
                    cpu_all = cpu_baseline_all_cores(args, wl, lut, min(cpu_budget, 2.0))
        if host_path and dist is None and args.model != "cloud3d":
            host = host_buffer_path(wl, sc, pf, data)
    if dist is not None and cpu_budget > 0:
        ctx.host_barrier(cpu_budget * 6 + 180)
    if rank != 0:
        e.close()
        return None

    score = prof["score"]
    k_ms = score["ms"] / max(score["launches"], 1)
    if args.model == "beam":
        e.set_option(1, 1)
        e.cells_walked(reset=True)
        pf.restore()
        sc.updateSensor(pf, data)
        e.synchronize()
        mean_cells = e.cells_walked() / (float(wl["n"]) * wl["beams"])
        e.set_option(1, 0)
    abytes = algorithmic_bytes(args.model, wl["n"], wl["beams"], mean_cells)
    achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    ev = pmc_evidence(args.model, args.cloud, k_ms, wl["n"], wl["beams"])
    backend_name = None if dist is None else dist.get_backend()
    multi = multi_gpu_fields(backend_name, None if dist is None else dist.get_world_size(), selftest, exchange_ms)
    collective_is = multi.get("collective_is")
    line = {
        "metric": metric_name(args),
        "value": value, "unit": "particle-beam evals/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "prewarm": max(0, prewarm), "ms_per_step": dt / steps * 1e3,
        **({} if dt_repeat is None else {"ms_per_step_repeat": dt_repeat / steps * 1e3}),
        "higher_is_better": True,
        "scaling": "strong" if getattr(args, "strong_total", None) else "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload_name(args, world),
                   "cloud": args.cloud, "resampler": args.resampler, "motion": args.motion, "particles_per_gpu": wl["n"],
                   "particles_total": n_total, "resampled_to": int(st.sample_count),
                   "kld_leaf_count": int(st.leaf_count), "parallelism": "particle-shard x%d" % world,
                   "world_size": (1 if dist is None else dist.get_world_size()),
                   "distance_lut": (None if args.model == "cloud3d" else LUT_NAMES[args.lut]),
                   "collective_backend": backend_name,
                   "shard_exchange": (None if dist is None else (
                       "mailbox (peer stores into IPC-mapped device memory)" if ran_with_mailbox
                       else "collective (%s)" % collective_is))},
        # `achieved` is ALGORITHMIC GB/s (bytes the reference's formulation moves per launch / the
        # kernel's duration), `frac` its fraction of the HBM peak; `bound` is what the PMC passes show limits the
        # kernel (valu_issue / l1_access_rate: the LUT working set is L2-resident, see issue_frac / hbm_measured_gbs / bound_is); None = no PMC
        # pass of this workload is committed, so nothing is claimed
        "roofline": {"bound": ev["bound"], "kernel": e.score_kernel_name(), "achieved": achieved,
                     "achieved_is": "algorithmic GB/s", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": ev["traffic"],
                     "traffic_source": ev["traffic_source"], "kernel_ms": k_ms,
                     "algorithmic_bytes_per_launch": abytes,
                     "launches_timed": int(score["launches"]),
                     # SURVEY 8(d) asks for the fraction against the measured streaming-copy rate as well
                     # (6.29 TB/s, MI355X_MICROARCH.md), the "measured HBM roofline" of north_star
                     "peak_measured_copy": HBM_MEASURED_COPY_GBS,
                     "frac_of_measured_copy": achieved / HBM_MEASURED_COPY_GBS},
        "cpu_baseline": cpu,
        "cpu_baseline_all_cores": cpu_all,
        "host_buffer_path": host,
        "kernel_ms_per_step": {k: v["ms"] / n_all for k, v in prof_all.items() if v["launches"]},
        "setup_s": t_setup,
    }
    if ev["bound"] is None:
        line["roofline"]["bound_is"] = "unmeasured: no committed PMC pass for this workload (profiles/pmc_traffic.json)"
    elif "bound_is" in ev:
        line["roofline"]["bound_is"] = ev["bound_is"]
    for k in ("issue_frac", "hbm_measured_gbs"):
        if k in ev:
            line["roofline"][k] = ev[k]
    if args.model in ("lf", "gompertz") and k_ms > 0:
        # what the kernel really gathers per evaluation is a 1- or 2-byte level id (SURVEY 8(d) counts the reference's
        # 4-byte float): the same launch priced at the bytes it touches, for whoever wants that figure
        gb = e.lut_entry_bytes() if hasattr(e, "lut_entry_bytes") else 2
        touched = float(gb) * wl["n"] * wl["beams"] + 40.0 * wl["n"] + 16.0 * wl["beams"]
        line["roofline"]["gathered_bytes_per_eval"] = gb
        line["roofline"]["achieved_at_gathered_bytes_gbs"] = touched / (k_ms * 1e-3) / 1e9
    if mean_cells is not None and args.model == "beam":
        line["roofline"]["mean_cells_per_ray"] = mean_cells
    line.update(multi)
    e.close()
    return line


def multi_gpu_fields(backend_name, group_size, selftest, exchange_ms):
    """What an N > 1 line says about its exchanges: which collective library stood behind `collective`, how many ranks
    its communicator had, what the mailbox bring-up found, and the sharded step timed once per exchange."""
    if backend_name is None:
        return {}
    collective_is = "RCCL (torch.distributed nccl)" if backend_name == "nccl" else \
        "torch.distributed %s (rehearsal on one GPU: RCCL refuses several ranks per device)" % backend_name
    out = {"mailbox_selftest": selftest, "rccl_ranks": int(group_size) if backend_name == "nccl" else 0,
           "collective_is": collective_is}
    if exchange_ms is not None:
        out["exchange_ms"] = exchange_ms
        out["exchange_ms_is"] = ("ms per whole sharded step (restore + sensor update + resample) with that exchange, "
                                 "same engines, same run; `collective` = " + collective_is)
    return out


def host_buffer_path(wl, sc, pf, data):
    """Seam A with host buffers (PlanarScanner::applyModelToSampleSet on the caller's std::vector<PFSample>): the set
    goes up, is scored, the weights come back, inside every call -- the PCIe-inclusive figure, never `value`."""
    e = sc.e
    host_samples = wl["samples"].copy()   # stands for the reference's `samples` vector: allocated once, kept
    reps = 20

    def per_update():
        for _ in range(3):
            sc.applyModelToSampleSet(data, host_samples, 0)
        t0h = time.perf_counter()
        for _ in range(reps):
            sc.applyModelToSampleSet(data, host_samples, 0)
        return (time.perf_counter() - t0h) / reps

    def per_cycle():
        for _ in range(2):
            pf.initWithSamples(host_samples)
            sc.updateSensor(pf, data)
            pf.updateResample()
            pf.getCurrentSet(out=host_samples)
        t0h = time.perf_counter()
        for _ in range(10):
            host_samples[:] = wl["samples"]  # (untimed in spirit: the host's own motion update rewrites the set)
            t1 = time.perf_counter()
            pf.initWithSamples(host_samples)
            sc.updateSensor(pf, data)
            pf.updateResample()
            pf.getCurrentSet(out=host_samples)
            per_cycle.t += time.perf_counter() - t1
        return per_cycle.t / 10

    per_cycle.t = 0.0
    d_plain = per_update()          # pageable: through the runtime's bounce buffers
    c_plain = per_cycle()
    e.registerHostBuffer(host_samples)  # what an integration does once, next to the allocation of `samples`
    try:
        per_cycle.t = 0.0
        host_samples[:] = wl["samples"]
        dth = per_update()
        chunks, pinned = e.seam_last_plan()
        dte = per_cycle()
    finally:
        e.unregisterHostBuffer(host_samples)
    evals = float(wl["n"]) * wl["beams"]
    return {"ms_per_update": dth * 1e3, "evals_per_s": evals / dth,
            "what": ("applyModelToSampleSet on a host-resident set registered once (bpf_host_buffer_register): ONE scoring "
                     "launch reads the %.1f MB of records over PCIe itself as its waves reach them and writes them back "
                     "whole with the new weight; no copy, nothing left for the calling thread but to wait"
                     % (wl["n"] * 32 / 1e6)) if chunks < 0 else
                    ("applyModelToSampleSet on a host-resident set registered once: %.1f MB of records up in %d chunks, "
                     "chunk k scored while chunk k + 1 crosses PCIe, the weights stored into pinned host memory by the "
                     "scoring launches and written into the records chunk by chunk" % (wl["n"] * 32 / 1e6, chunks)),
            "pinned": bool(pinned), "chunks": chunks,
            "ms_per_update_unregistered": d_plain * 1e3,
            "unregistered_is": ("the same buffer as plain pageable memory: it goes up through the engine's own pinned bounce "
                                "buffer (BPF_OPT_HOST_DIRECT_PAGEABLE = 0), in two chunks for a set of this size"),
            # the whole cycle with the set crossing PCIe both ways (SURVEY 8(d)'s end-to-end figure): H2D of the set
            # (32 B/particle), sensor update, resample, D2H of the resampled set -- never `value` either
            "end_to_end_cycle_ms": dte * 1e3, "end_to_end_evals_per_s": evals / dte,
            "end_to_end_cycle_ms_unregistered": c_plain * 1e3}


SUB_KEYS = ("metric", "value", "unit", "steps", "warmup", "prewarm", "ms_per_step", "ms_per_step_repeat", "scaling",
            "n_gpus", "config",
            "roofline", "cpu_baseline", "kernel_ms_per_step", "exchange_ms", "mailbox_selftest", "rccl_ranks", "setup_s")


def sub_record(line):
    return None if line is None else {k: line[k] for k in SUB_KEYS if k in line}


def sub_args(args, **kw):
    a = copy.copy(args)
    a.config, a.model, a.particles, a.beams, a.map_size = None, "lf", None, 1081, 2000
    a.cloud, a.resampler, a.motion, a.strong_total = "converged", "multinomial", "none", None
    for k, v in kw.items():
        setattr(a, k, v)
    apply_config_preset(a)
    if a.particles is None:
        a.particles = 200000 if a.model == "cloud3d" else 100000
    return a


def is_plain_headline(args):
    return (args.config in (None, 2) and args.model == "lf" and args.particles in (None, 100000) and
            args.beams == 1081 and args.map_size == 2000 and args.cloud == "converged" and
            args.resampler == "multinomial" and args.motion == "none")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--model", default="lf", choices=["lf", "gompertz", "beam", "cloud3d"])
    ap.add_argument("--cloud", default="converged", choices=["converged", "spread"])
    ap.add_argument("--resampler", default="multinomial", choices=["multinomial", "systematic"])
    ap.add_argument("--motion", default="none",
                    choices=["none", "diff", "omni", "diff-corrected", "omni-corrected", "gaussian"],
                    help="also run Odom::updateAction on the device inside every step (default: the restored set "
                         "stands in for the motion update, as the metric is sensor update + resample)")
    ap.add_argument("--particles", type=int, default=None,
                    help="particles per GPU (default 100000; 200000 for cloud3d)")
    ap.add_argument("--beams", type=int, default=1081)
    ap.add_argument("--map-size", type=int, default=2000)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--prewarm", type=int, default=60,
                    help="untimed steps in front of the --warmup steps that bring the GPU's clocks up: at least this "
                         "many, and at least --prewarm-seconds of them (the clocks take ~40 ms of load to come up: with "
                         "60 steps = 8 ms a 20-step timed region reads 0.147 ms per step, with 300 steps 0.123); the count "
                         "that ran is reported as `prewarm` in the line; 0 = none")
    ap.add_argument("--prewarm-seconds", type=float, default=0.25)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 3, 4, 5],
                    help="BASELINE.json configs[k-1]: 1 = LF 5000 x 181 on a 400^2 map (the reference's own CPU case), "
                         "2 = LF 100k x 1081 (default), 3 = beam model 100k x 1081, 4 = LF 125k particles per GPU (1 M over "
                         "8 GPUs), 5 = 3-D 200k x 65 536 points")
    ap.add_argument("--lut", default="reference", choices=["reference", "exact-edt"],
                    help="2-D distance LUT: the reference's brushfire (host builder, what OccupancyMap::updateDistancesLUT "
                         "gives; default) or the exact EDT built on the device")
    ap.add_argument("--extras", default="auto", choices=["auto", "on", "off"],
                    help="other_configs / strong_scaling / exchange_ms sub-records in the same line (auto: when the run "
                         "is the plain headline)")
    ap.add_argument("--sub-cpu-budget", type=float, default=4.0, help="seconds of CPU baseline work per sub-record")
    ap.add_argument("--host-path", default="on", choices=["on", "off"],
                    help="also time Seam A with host buffers (host_buffer_path; N = 1 only)")
    args = ap.parse_args(argv)
    args.strong_total = None
    return args


def main():
    args = parse_args()
    extras = args.extras == "on" or (args.extras == "auto" and is_plain_headline(args))
    apply_config_preset(args)
    if args.particles is None:
        args.particles = 200000 if args.model == "cloud3d" else 100000

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process has made no GPU call (torch is not even
        # imported yet); it starts the N ranks as children and relays rank 0's JSON line.
        if os.environ.get("BPF_BENCH_REHEARSAL") == "1" and args.gpus > 4:
            # all ranks of a rehearsal share ONE GPU; the GPU boxes allow six processes on a card and the launcher
            # counts as well (a six-rank rehearsal was killed by the box's process guard)
            sys.stderr.write("bench.py: a rehearsal (all ranks on one GPU) takes at most 4 ranks\n")
            raise SystemExit(2)
        raise SystemExit(self_launch(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus and os.environ.get("BPF_FORCE_SHARDED") != "1":
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%s\n" % (args.gpus, os.environ.get("WORLD_SIZE")))
        raise SystemExit(2)

    # Native libraries (RCCL prints a version banner) write to fd 1; keep the real stdout for the
    # one JSON line and send everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    ctx = Ctx()
    world, rank = ctx.world, ctx.rank
    with Stage("import torch", 600):
        import torch
    ctx.torch = torch
    force_sharded = os.environ.get("BPF_FORCE_SHARDED") == "1"  # exercise the sharded path at world 1
    if world > 1 or force_sharded:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the host-side group never leaves it
        limit = datetime.timedelta(seconds=300)
        with Stage("process group rendez-vous (init_process_group)", 330):
            if ctx.rehearsal:
                # rehearsal of the N > 1 flow on a one-GPU box: every rank on cuda:0, gloo for the host-side exchanges
                # (RCCL refuses two ranks on one device); the timings mean nothing, the code path is the real one
                ctx.local_rank = 0
                dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=1800))
            else:
                torch.cuda.set_device(ctx.local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", ctx.local_rank), timeout=limit)
        ctx.dist = dist
        if dist.get_backend() != "gloo":
            with Stage("host-side group (gloo)", 120):
                try:
                    ctx.host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=1800))
                except Exception as err:  # noqa: BLE001 -- the ranks then park in the default group's barrier
                    sys.stderr.write("bench.py: no gloo side group (%s); parking in the default barrier\n" % (err,))
                    ctx.host_group = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    torch.cuda.set_device(ctx.local_rank)

    gc.collect()
    gc.freeze()  # see measure()

    line = measure(args, ctx, args.steps, args.warmup, args.prewarm, args.cpu_budget, all_cores=(world == 1),
                   host_path=(world == 1 and args.host_path == "on"),
                   compare_exchanges=(extras and ctx.dist is not None),
                   prewarm_s=args.prewarm_seconds)
    if extras:
        sub_budget = min(args.sub_cpu_budget, args.cpu_budget)
        if ctx.dist is None:
            # BASELINE configs[2], [4], [0] beside the headline: short timed regions of their own, same process
            others = []
            for cfg, k, w, pw, mode in ((3, max(10, min(args.steps, 40)), 3, 10, 3),
                                        (5, max(5, min(args.steps // 2, 12)), 2, 3, 3),
                                        (1, max(200, args.steps), 20, 100, 1)):
                others.append(sub_record(measure(sub_args(args, config=cfg), ctx, k, w, pw, sub_budget,
                                                 profile_mode=mode, repeat=True)))
            line["other_configs"] = others
        # BASELINE configs[3] as worded: 1 M particles TOTAL, 1 M / N per GPU (the N = 1 run is the curve's anchor)
        sa = sub_args(args, strong_total=STRONG_TOTAL)
        sa.particles = STRONG_TOTAL * (rank + 1) // world - STRONG_TOTAL * rank // world
        strong = measure(sa, ctx, max(20, min(args.steps, 60)), 5, 20, sub_budget,
                         profile_mode=(3 if world == 1 else 1), repeat=True)
        if rank == 0:
            line["strong_scaling"] = sub_record(strong)
    if rank == 0:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if ctx.dist is not None:
        with Stage("shutdown barrier", 120):
            ctx.dist.barrier()
            ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()
