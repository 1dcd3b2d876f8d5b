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
    assert ev["bound"] == "valu_issue" and ev["traffic"] and "static" in ev["traffic_source"]
    assert 0.3 < ev["issue_frac"] < 1.1 and ev["hbm_measured_gbs"] > 0
    other = bench.pmc_evidence("lf", "converged", 0.0737, 5000, 181)
    assert other["traffic"] is None and "issue_frac" not in other


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
