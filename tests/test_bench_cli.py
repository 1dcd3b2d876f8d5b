"""bench.py's host logic without a GPU: the config presets name BASELINE.json's own workloads, the metric string is
BASELINE's only for the headline shape, the roofline accounting follows SURVEY 8(d), and the launcher refuses what it
cannot run (a WORLD_SIZE that contradicts --gpus, a machine without a GPU) instead of printing a line."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    a = argparse.Namespace(config=None, model="lf", particles=None, beams=1081, map_size=2000, cloud="converged")
    a.__dict__.update(kw)
    bench.apply_config_preset(a)
    if a.particles is None:
        a.particles = 200000 if a.model == "cloud3d" else 100000
    return a


def _baseline():
    return json.load(open(os.path.join(ROOT, "BASELINE.json")))


def test_presets_select_the_baseline_configs():
    cfgs = _baseline()["configs"]
    a1 = _args(config=1)
    assert (a1.model, a1.particles, a1.beams, a1.map_size) == ("lf", 5000, 181, 400)
    assert "5000 particles" in cfgs[0] and "181-beam" in cfgs[0] and "400" in cfgs[0]
    a2 = _args(config=2)
    assert (a2.model, a2.particles, a2.beams, a2.map_size) == ("lf", 100000, 1081, 2000)
    assert "100k particles" in cfgs[1] and "1081 beams" in cfgs[1]
    a3 = _args(config=3)
    assert (a3.model, a3.particles, a3.beams) == ("beam", 100000, 1081) and "beam-model" in cfgs[2]
    a4 = _args(config=4)
    assert (a4.model, a4.particles) == ("lf", 125000) and "1M particles sharded 8" in cfgs[3]
    a5 = _args(config=5)
    assert (a5.model, a5.particles) == ("cloud3d", 200000) and "200k particles" in cfgs[4]
    # an explicit particle count given with a preset wins
    assert _args(config=1, particles=777).particles == 777


def test_metric_is_baselines_string_only_for_the_headline_shape():
    base = _baseline()["metric"]
    assert bench.metric_name(_args(config=2)) == base
    assert bench.metric_name(_args()) == base
    for cfg in (1, 3, 4, 5):
        name = bench.metric_name(_args(config=cfg))
        assert name != base and name.startswith("particle-beam evals/sec (sensor update+resample)")
    assert "beam-model raycast" in bench.metric_name(_args(config=3))
    assert "65536 points" in bench.metric_name(_args(config=5))


def test_workload_names():
    assert bench.workload_name(_args(config=2), 1) == "2D likelihood-field, 100k particles, 1081 beams, 2000x2000 map"
    w8 = bench.workload_name(_args(config=2), 8)
    assert "800000 particles" in w8 and "8 GPUs" in w8 and "ONE filter" in w8
    assert "configs[0]" in bench.workload_name(_args(config=1), 1)
    w4 = bench.workload_name(_args(config=4), 8)
    assert "125k particles per GPU" in w4 and "8 GPUs, 1000000 particles" in w4
    assert bench.workload_name(_args(config=5), 1).startswith("3D (octomap) likelihood-field, 200k particles")


def test_algorithmic_bytes_follow_the_survey():
    # LF 4 B/eval + 40 B/particle + 16 B/beam -> 436.4 MB at 100 k x 1081
    assert bench.algorithmic_bytes("lf", 100000, 1081) == 4.0 * 100000 * 1081 + 40.0 * 100000 + 16.0 * 1081
    assert abs(bench.algorithmic_bytes("lf", 100000, 1081) - 436.4e6) < 0.1e6
    # beam model: C x 1 B/eval with the measured mean cells per ray
    assert bench.algorithmic_bytes("beam", 10, 20, mean_cells=76.2) == 76.2 * 200 + 400.0 + 320.0
    assert bench.algorithmic_bytes("cloud3d", 200000, 65536) == 5.0 * 200000 * 65536 + 12.0 * 65536 + 40.0 * 200000


def test_pmc_evidence_is_only_attached_to_the_workload_it_was_measured_on():
    ev = bench.pmc_evidence("lf", "converged", 0.0737, 100000, 1081)
    # the headline workload: the record carries what the end-of-round-3 measurements showed (L1 access rate; VALU issue
    # is 57 % of the cycles) and the sentence that says so
    assert ev["bound"] == "l1_access_rate" and "accesses per vector read" in ev["bound_is"]
    assert ev["traffic"] and "static" in ev["traffic_source"]
    assert 0.3 < ev["issue_frac"] < 1.1 and ev["hbm_measured_gbs"] > 0
    other = bench.pmc_evidence("lf", "converged", 0.0737, 7777, 181)
    assert other["traffic"] is None and "issue_frac" not in other
    # ... and without a record NOTHING is claimed about what binds the kernel (it used to say "hbm")
    assert other["bound"] is None
    assert bench.pmc_evidence("lf", "spread", 0.12, 123456, 1081)["bound"] is None
    # round 3: the spread cloud, 125 k, 1 M and cfg 1 have passes of their own, keyed by size where the key needs it
    # the spread cloud: 747 MB of L2 fills per launch (6.3 TB/s, "hbm") until it was scored in tile order; now below
    # the algorithmic bytes and issue-bound like the others
    sp = bench.pmc_evidence("lf", "spread", 0.0887, 100000, 1081)
    assert sp["bound"] == "valu_issue" and sp["traffic"] < 436.4e6
    assert bench.pmc_evidence("lf", "spread", 0.0887, 100000, 1081)["hbm_measured_gbs"] < 0.5 * bench.HBM_PEAK_GBS
    assert bench.pmc_evidence("lf", "converged", 0.089, 125000, 1081)["bound"] == "valu_issue"
    assert bench.pmc_evidence("lf", "converged", 0.66, 1000000, 1081)["bound"] == "valu_issue"
    assert bench.pmc_evidence("lf", "converged", 0.0095, 5000, 181)["bound"] == "latency"


def test_plain_headline_detection_and_sub_records():
    a = bench.parse_args([])
    assert bench.is_plain_headline(a)
    assert bench.is_plain_headline(bench.parse_args(["--gpus", "1", "--steps", "20", "--warmup", "5"]))
    for argv in (["--config", "3"], ["--model", "beam"], ["--particles", "5000"], ["--cloud", "spread"],
                 ["--resampler", "systematic"], ["--motion", "diff"], ["--beams", "181"]):
        assert not bench.is_plain_headline(bench.parse_args(argv)), argv
    cfgs = _baseline()["configs"]
    s3 = bench.sub_args(a, config=3)
    assert (s3.model, s3.particles, s3.beams, s3.map_size) == ("beam", 100000, 1081, 2000)
    assert bench.workload_name(s3, 1) == cfgs[2].replace("×", "x").split(", 1x")[0].replace("2000x2000", "2000x2000")
    s5 = bench.sub_args(a, config=5)
    assert (s5.model, s5.particles) == ("cloud3d", 200000)
    s1 = bench.sub_args(a, config=1)
    assert (s1.model, s1.particles, s1.beams, s1.map_size) == ("lf", 5000, 181, 400)
    # configs[3] as worded: 1 M particles TOTAL whatever N is
    for world in (1, 2, 4, 8):
        st = bench.sub_args(a, strong_total=bench.STRONG_TOTAL)
        per = [bench.STRONG_TOTAL * (r + 1) // world - bench.STRONG_TOTAL * r // world for r in range(world)]
        assert sum(per) == 1000000
        st.particles = per[0]
        name = bench.workload_name(st, world)
        assert "1M particles sharded %dxMI355X" % world in name and "1000000 particles TOTAL" in name
        assert "%d per GPU" % (1000000 // world) in name
    assert "1M particles sharded 8" in cfgs[3]
    # a sub-record keeps the graded fields
    line = {"metric": "m", "value": 1.0, "ms_per_step": 2.0, "roofline": {"frac": 0.5}, "cpu_baseline": {"value": 3},
            "config": {"workload": "w"}, "junk": 1}
    sub = bench.sub_record(line)
    assert set(sub) == {"metric", "value", "ms_per_step", "roofline", "cpu_baseline", "config"}


def test_cpu_sample_is_bounded_by_the_budget():
    # headline: one step of the whole set fits 12 s at the oracle's pace; a 4 s budget of the beam model does not
    assert bench.cpu_sample_particles("lf", 100000, 1081, 12.0) == 100000
    n3 = bench.cpu_sample_particles("beam", 100000, 1081, 4.0)
    assert 5000 < n3 < 40000
    assert bench.cpu_sample_particles("lf", 1000000, 1081, 4.0) < 200000
    assert bench.cpu_sample_particles("lf", 5000, 181, 4.0) == 5000
    assert 100 <= bench.cpu_sample_particles("cloud3d", 200000, 65536, 4.0) < 2000


def test_multi_gpu_fields_shape():
    """(a)-(c) of the N > 1 line: exchange_ms carries both exchanges, the self-test verdict and the communicator
    size are there, and a gloo rehearsal does not pass for RCCL."""
    assert bench.multi_gpu_fields(None, None, None, None) == {}
    ex = {"mailbox": 0.14, "collective": 0.17}
    f = bench.multi_gpu_fields("nccl", 8, "pass", ex)
    assert f["rccl_ranks"] == 8 and f["mailbox_selftest"] == "pass" and f["exchange_ms"] == ex
    assert "RCCL" in f["collective_is"] and "RCCL" in f["exchange_ms_is"]
    g = bench.multi_gpu_fields("gloo", 2, "fail: window self-test", {"mailbox": None, "collective": 0.2})
    assert g["rccl_ranks"] == 0 and "rehearsal" in g["collective_is"] and g["mailbox_selftest"].startswith("fail")
    for k in ("exchange_ms", "mailbox_selftest", "rccl_ranks"):
        assert k in bench.SUB_KEYS
    for k in ("roofline", "cpu_baseline", "config", "ms_per_step", "value"):
        assert k in bench.SUB_KEYS


def test_a_stage_that_exceeds_its_bound_names_itself_and_exits_nonzero():
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "with bench.Stage('mailbox connect round', 0.3):\n    time.sleep(20)\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode == 124 and r.stdout == b""
    assert b"stage 'mailbox connect round' exceeded" in r.stderr and r.stderr.count(b"\n") == 1
    # a stage that fails names itself too, and the error still goes up
    code = ("import sys; sys.path.insert(0, %r); import bench\n"
            "with bench.Stage('engine set-up', 5):\n    raise RuntimeError('boom')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    assert r.returncode != 0 and b"stage 'engine set-up' failed" in r.stderr and b"boom" in r.stderr


def _parked_rank(rank, world, port, q):
    import datetime
    import time
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    ctx = bench.Ctx()
    ctx.dist = dist
    ctx.host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))
    t0 = time.perf_counter()
    if rank == 0:
        time.sleep(1.5)  # rank 0's CPU baseline
    ctx.host_barrier(30)
    q.put((rank, time.perf_counter() - t0))
    dist.destroy_process_group()


def test_ranks_are_parked_on_the_host_while_rank_0_times_the_cpu_baseline():
    import multiprocessing as mp
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_parked_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got[1] >= 1.0  # rank 1 waited for rank 0's CPU leg


def _run(argv, env_extra, timeout=240):
    env = dict(os.environ)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_a_world_size_that_contradicts_gpus_is_refused():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout == b"" and b"WORLD_SIZE" in r.stderr


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a machine WITHOUT a GPU")
def test_without_a_gpu_there_is_no_line_and_no_cpu_fallback():
    env = {k: "" for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env = {k: v for k, v in os.environ.items() if k not in env}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--prewarm", "0",
                        "--cpu-budget", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert r.returncode != 0 and r.stdout == b""
    assert b"needs a GPU" in r.stderr or b"no CPU fallback" in r.stderr
