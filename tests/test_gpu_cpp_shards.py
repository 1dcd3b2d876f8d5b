"""A sharded filter brought up and driven from plain C++ processes (tests/cpp/shard_two_procs.cpp): TCP rendez-vous
inside the library (bpf_shard_bootstrap), mailbox IPC handles over it, then bpf_shard_update_sensor_planar /
bpf_shard_update_resample -- no Python, torch.distributed or launcher anywhere in the data path.  Two ranks share the
one GPU of the box; a third process runs the same filter unsharded."""
import os
import re
import socket
import subprocess

import numpy as np
import pytest

from scenario import Scenario

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    exe = tmp_path / "shard_two_procs"
    libdir = os.path.join(ROOT, "badger_amcl_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shard_two_procs.cpp"), "-o", str(exe),
                           "-L", libdir, "-lbadger_pf_hip", "-Wl,-rpath," + libdir])
    return exe


def test_shard_driver_compiles_and_the_rccl_object_links():
    """CPU: the C++ driver builds against the C-ABI; libbadger_pf_rccl.so exists, is linked against librccl and
    exports the collective entry points the bootstrap resolves."""
    import tempfile
    import pathlib
    from badger_amcl_amd import build
    build.build()
    with tempfile.TemporaryDirectory() as d:
        _compile(pathlib.Path(d))
    so = build.OUT_RCCL
    assert os.path.exists(so)
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True, check=True).stdout
    assert "librccl" in needed
    syms = subprocess.run(["nm", "-D", so], capture_output=True, text=True, check=True).stdout
    for name in ("bpfc_unique_id", "bpfc_init", "bpfc_allgather_f64", "bpfc_allreduce_sum_i64",
                 "bpfc_allreduce_sum_i32", "bpfc_destroy"):
        assert name in syms


@pytest.mark.gpu
@pytest.mark.parametrize("world,flags", [(2, 2), (3, 2), (1, 1)])
def test_cpp_ranks_bootstrap_over_tcp_and_agree(tmp_path, orc, world, flags):
    """flags 2 = BPF_BOOTSTRAP_MAILBOX_ONLY (two and three ranks on the one GPU: RCCL refuses several ranks per device,
    the mailbox does not); flags 1 = BPF_BOOTSTRAP_FORCE_COLLECTIVE at world size 1: libbadger_pf_rccl.so is loaded, a
    real RCCL communicator is created and the totals / windows go through ncclAllGather / ncclAllReduce -- the code
    path a node takes when its ranks cannot map each other's memory, as far as one GPU can exercise it."""
    exe = _compile(tmp_path)
    sc = Scenario(orc, size=400, n=6000, beams=91, cloud="converged")
    paths = {}
    for name, arr in (("cells", sc.cells.astype(np.int32)), ("lut", sc.lut.astype(np.float32)),
                      ("samples", sc.samples), ("ranges", sc.ranges), ("angles", sc.angles)):
        paths[name] = str(tmp_path / (name + ".bin"))
        np.ascontiguousarray(arr).tofile(paths[name])
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    prefix = str(tmp_path / "out")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([str(exe), paths["cells"], paths["lut"], paths["samples"], paths["ranges"], paths["angles"],
                          "400", str(world), str(port), str(flags), prefix],
                         capture_output=True, text=True, env=env, timeout=240)
    assert res.returncode == 0, res.stdout + res.stderr
    rows = {}
    single = {}
    for line in res.stdout.splitlines():
        m = re.match(r"rank (\d+) cycle (\d+) mode (\d+) M (\d+) leaf (\d+) bins (\d+) windows (\d+) local (\d+) "
                     r"rng (\d+) miss (\d+)", line)
        if m:
            v = [int(x) for x in m.groups()]
            rows[(v[0], v[1])] = v[2:]
        m = re.match(r"single cycle (\d+) M (\d+) leaf (\d+) bins (\d+) rng (\d+)", line)
        if m:
            v = [int(x) for x in m.groups()]
            single[v[0]] = v[1:]
    assert len(rows) == 3 * world and len(single) == 3
    for cycle in range(3):
        ref = rows[(0, cycle)]
        mode, M, leaf, bins, windows, local0, rng, miss = ref
        assert mode == (1 if flags == 2 else 2) and miss == 0  # 1 = mailbox (between processes), 2 = RCCL
        locals_ = []
        for r in range(world):
            got = rows[(r, cycle)]
            assert (got[0], got[1], got[2], got[3], got[6]) == (mode, M, leaf, bins, rng)  # every rank agrees
            locals_.append(got[5])
            assert got[5] == (M * (r + 1)) // world - (M * r) // world
        assert sum(locals_) == M
        # the shards, put together in rank order, are the unsharded filter's set
        parts = [np.fromfile("%s.rank%d.cycle%d.bin" % (prefix, r, cycle), dtype=np.float64).reshape(-1, 4)
                 for r in range(world)]
        whole = np.concatenate(parts)
        one = np.fromfile("%s.single.cycle%d.bin" % (prefix, cycle), dtype=np.float64).reshape(-1, 4)
        assert single[cycle][0] == M and single[cycle][1] == leaf and single[cycle][3] == rng
        # (the shards' CDF slices are total_q / sum(totals): a draw within rounding of a slice edge may pick the
        # neighbouring particle -- none expected in a few thousand draws, one allowed)
        differing = np.flatnonzero(np.any(whole[:, :3] != one[:, :3], axis=1))
        assert differing.size <= 1
        assert np.all(whole[:, 3] == 1.0 / M)
