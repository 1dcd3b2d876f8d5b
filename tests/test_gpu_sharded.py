"""GPU test of the sharded path: two ranks share cuda:0, each drives the bpf_shard_* stage functions on its half of
the set; together they must reproduce the single-engine result, which the other gpu tests tie to the oracle.  The
exchanges run both ways: through the mailbox (peer stores into IPC-mapped device memory, here two processes on one
GPU; include/badger_pf.h bpf_shard_mailbox_*) and as collectives (gloo, staged through the host).  The remaining
host-side exchanges (set-up, beam-skip counts) always use gloo."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# update + resample cycles of the two-rank comparison (BPF_SHARD_CYCLES=150 turns it into a soak of the exchanges)
CYCLES = int(os.environ.get("BPF_SHARD_CYCLES", "2"))
ODOM = (2, 0.05, 0.04, 0.03, 0.02, 0.0)                         # diff-corrected
ODATA = ((1.0, 2.0, 0.3), (0.03, -0.01, 0.02), (0.03, 0.01, 0.02))  # pose, delta, absolute motion


def _scenario(cloud="converged"):
    from oracle import pyoracle as orc
    from scenario import Scenario
    return orc, Scenario(orc, size=400, n=6000, beams=181, cloud=cloud)


def _worker(rank, world, port, out_dir, cloud, device_min, resampler, exchange):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import badger_amcl_amd as bpf
    from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
    from scenario import Scenario
    orc, sc = _scenario(cloud)
    n = sc.samples.shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    e = bpf.Engine(0)
    shard = Scenario.__new__(Scenario)
    shard.__dict__.update(sc.__dict__)
    shard.samples = np.ascontiguousarray(sc.samples[lo:hi])
    m, scn, pf, data = shard.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    pf.setResampleModel(resampler)
    if exchange == "mailbox-staged":
        # the mailbox step without its single-launch forms (k_normalize_gathered_cdf, k_shard_stop_block): separate
        # normalise / CDF launches, the host's ordered replay of the window's keys, the tail launch
        e.set_option(5, 0)  # BPF_OPT_FUSED_RESAMPLE
        exchange = "mailbox"
    b = HipShardBackend(e, scn, pf, torch.device("cuda", 0))
    b.kld_device_min = device_min
    sf = ShardedFilter(b, dist, first_window=1024, exchange=exchange)
    assert sf.mailbox == (exchange == "mailbox")
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    recs = []
    for cycle in range(CYCLES):
        sf.update_action(od, bpf.OdomData(*ODATA))
        sf.update_sensor(data)
        w_after = pf.getCurrentSet().samples.copy()
        sf.update_resample()
        st = sf.state()
        recs.append(dict(w=w_after, samples=pf.getCurrentSet().samples.copy(), M=st.sample_count, leaf=st.leaf_count,
                         bins=st.bin_count, rng=pf.getRngState(), conv=st.converged, miss=st.cdf_miss,
                         w_slow=st.w_slow, windows=st.windows))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(recs, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()
    e.close()


@pytest.mark.parametrize("exchange", ["mailbox", "collective", "mailbox-staged"])
@pytest.mark.parametrize("cloud,device_min,resampler", [("converged", 8192, 0), ("spread", 512, 0),
                                                        ("converged", 8192, 1)])
def test_two_ranks_on_one_gpu_equal_single_engine(tmp_path, cloud, device_min, resampler, exchange):
    """converged: early KLD stop inside the first window (mailbox: k_shard_resample_block on every rank, the draws and
    their consumer in one launch; collective and mailbox-staged: the host's ordered replay).  spread with a low device threshold: no stop in the first window, so
    one window with the whole stream follows and the stop rule runs on the device."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), cloud, device_min, resampler, exchange), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True) for r in range(2)]

    import badger_amcl_amd as bpf
    orc, sc = _scenario(cloud)
    n = sc.samples.shape[0]
    e = bpf.Engine(0)
    m, scn, pf, data = sc.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    pf.setResampleModel(resampler)
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    for cycle in range(CYCLES):
        od.updateAction(pf, bpf.OdomData(*ODATA))
        scn.updateSensor(pf, data)
        w_ref = pf.getCurrentSet().samples[:, 3].copy()
        pf.updateResample()
        st = pf.getState()
        cur = pf.getCurrentSet()
        r0, r1 = recs[0][cycle], recs[1][cycle]
        w_sh = np.concatenate([r0["w"][:, 3], r1["w"][:, 3]])
        assert np.allclose(w_sh, w_ref, rtol=1e-12, atol=0)
        for r in (r0, r1):
            assert r["M"] == st.sample_count and r["leaf"] == st.leaf_count and r["bins"] == st.bin_count
            assert r["rng"] == pf.getRngState()
            assert r["conv"] == st.converged and not r["miss"]
        merged = np.concatenate([r0["samples"], r1["samples"]])
        assert np.array_equal(merged[:, :3], cur.samples[:, :3])
        assert np.all(merged[:, 3] == 1.0 / st.sample_count)
    if cloud == "spread" and resampler == 0:
        assert recs[0][0]["windows"] == 2 and recs[0][0]["M"] > 1024  # first window + the whole-stream window
    e.close()


def _cloud_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import badger_amcl_amd as bpf
    from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
    from oracle import pyoracle as orc
    from test_gpu_cloud import _setup
    lut, pts, s, tf_xyz, tf_quat, max_dist = _setup(orc, 3000, 8, 256, seed=6)
    n = s.shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    e = bpf.Engine(0)
    om = bpf.OctoMap(e, 0.05)
    om.setDistancesLUT(lut.pose_indices, lut.distance_ratios, lut.min_cells, lut.max_cells, max_dist)
    sc = bpf.PointCloudScanner(e)
    sc.init(128, om)
    sc.setPointCloudModel(0.5, 0.05, 0.1)
    sc.setMapFactors(0.95, 0.95, 0.3)
    sc.setPointCloudScannerToFootprintTF(tf_xyz, tf_quat)
    pf = bpf.ParticleFilter(e, 100, n, 0.0, 0.0, 85.0)
    pf.srand48(5)
    pf.initWithSamples(np.ascontiguousarray(s[lo:hi]))
    sf = ShardedFilter(HipShardBackend(e, sc, pf, torch.device("cuda", 0)), dist, first_window=1024)
    assert sf.mailbox  # "auto": both ranks can map each other's mailbox, so the exchanges go through it
    sf.update_sensor(bpf.PointCloudData(pts))
    w_after = pf.getCurrentSet().samples.copy()
    sf.update_resample()
    st = sf.state()
    rec = dict(w=w_after, samples=pf.getCurrentSet().samples.copy(), M=st.sample_count, leaf=st.leaf_count,
               rng=pf.getRngState())
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array([rec], dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()
    e.close()


def test_two_ranks_cloud3d_equal_single_engine(tmp_path):
    """The 3-D path shards like the planar one: scoring per shard (bpf_shard_score_cloud), then the same
    exchanges."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_cloud_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True)[0] for r in range(2)]
    import badger_amcl_amd as bpf
    from oracle import pyoracle as orc
    from test_gpu_cloud import _setup
    lut, pts, s, tf_xyz, tf_quat, max_dist = _setup(orc, 3000, 8, 256, seed=6)
    e = bpf.Engine(0)
    om = bpf.OctoMap(e, 0.05)
    om.setDistancesLUT(lut.pose_indices, lut.distance_ratios, lut.min_cells, lut.max_cells, max_dist)
    sc = bpf.PointCloudScanner(e)
    sc.init(128, om)
    sc.setPointCloudModel(0.5, 0.05, 0.1)
    sc.setMapFactors(0.95, 0.95, 0.3)
    sc.setPointCloudScannerToFootprintTF(tf_xyz, tf_quat)
    pf = bpf.ParticleFilter(e, 100, 3000, 0.0, 0.0, 85.0)
    pf.srand48(5)
    pf.initWithSamples(s)
    sc.updateSensor(pf, bpf.PointCloudData(pts))
    w_ref = pf.getCurrentSet().samples[:, 3].copy()
    pf.updateResample()
    st = pf.getState()
    w_sh = np.concatenate([recs[0]["w"][:, 3], recs[1]["w"][:, 3]])
    assert np.allclose(w_sh, w_ref, rtol=1e-12, atol=0)
    for r in recs:
        assert r["M"] == st.sample_count and r["leaf"] == st.leaf_count and r["rng"] == pf.getRngState()
    merged = np.concatenate([recs[0]["samples"], recs[1]["samples"]])
    assert np.array_equal(merged[:, :3], pf.getCurrentSet().samples[:, :3])
    e.close()


BEAMSKIP = dict(do_beamskip=1, beam_skip_distance=0.5, beam_skip_threshold=0.3, beam_skip_error_threshold=0.9)


def _beamskip_scenario():
    from oracle import pyoracle as orc
    from scenario import Scenario
    sc = Scenario(orc, size=200, n=3000, beams=60, cloud="converged", frac_max=0.0, frac_nan=0.0)
    # a tight cloud: the first resample reports "converged", which arms beam skipping for the second update
    from badger_amcl_amd import synth
    sc.samples = synth.converged_cloud(3000, sc.pose, seed=12, sigma=(0.05, 0.05, 0.02))
    return orc, sc


def _beamskip_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import badger_amcl_amd as bpf
    from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
    from scenario import Scenario
    orc, sc = _beamskip_scenario()
    n = sc.samples.shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    e = bpf.Engine(0)
    shard = Scenario.__new__(Scenario)
    shard.__dict__.update(sc.__dict__)
    shard.samples = np.ascontiguousarray(sc.samples[lo:hi])
    m, scn, pf, data = shard.gpu_objects(e, 60, "prob", min_samples=100, max_samples=n, seed=3, model_kw=BEAMSKIP)
    sf = ShardedFilter(HipShardBackend(e, scn, pf, torch.device("cuda", 0)), dist, first_window=1024)
    assert sf.mailbox  # "auto": both ranks can map each other's mailbox, so the exchanges go through it
    recs = []
    for cycle in range(2):
        sf.update_sensor(data)
        w_after = pf.getCurrentSet().samples.copy()
        sf.update_resample()
        st = sf.state()
        recs.append(dict(w=w_after, samples=pf.getCurrentSet().samples.copy(), M=st.sample_count, conv=st.converged,
                         rng=pf.getRngState()))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(recs, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()
    e.close()


def test_two_ranks_prob_model_with_beam_skipping(tmp_path):
    """The prob model's beam skipping needs per-beam agreement counts over the WHOLE set: one all-reduce of the
    int32 counts between its two passes (bpf_shard_beam_counts_dev / bpf_shard_score_planar_finish)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_beamskip_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True) for r in range(2)]
    import badger_amcl_amd as bpf
    orc, sc = _beamskip_scenario()
    n = sc.samples.shape[0]
    e = bpf.Engine(0)
    m, scn, pf, data = sc.gpu_objects(e, 60, "prob", min_samples=100, max_samples=n, seed=3, model_kw=BEAMSKIP)
    for cycle in range(2):
        if cycle == 1:
            assert pf.getState().converged == 1  # beam skipping is armed for this update
        scn.updateSensor(pf, data)
        w_ref = pf.getCurrentSet().samples[:, 3].copy()
        pf.updateResample()
        st = pf.getState()
        r0, r1 = recs[0][cycle], recs[1][cycle]
        w_sh = np.concatenate([r0["w"][:, 3], r1["w"][:, 3]])
        assert np.allclose(w_sh, w_ref, rtol=1e-12, atol=0)
        for r in (r0, r1):
            assert r["M"] == st.sample_count and r["rng"] == pf.getRngState() and r["conv"] == st.converged
        merged = np.concatenate([r0["samples"], r1["samples"]])
        assert np.array_equal(merged[:, :3], pf.getCurrentSet().samples[:, :3])
    e.close()


RECOVERY_ALPHA = (0.001, 0.1)  # the node's default decay rates (node.cpp:122-123)


def _recovery_scan(sc, cycle):
    return [sc.ranges, np.clip(sc.ranges * 0.6, 0.05, 29.0), np.full(sc.ranges.shape[0], 1.0)][cycle]


def _recovery_worker(rank, world, port, out_dir, resampler):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
    from scenario import Scenario
    orc, sc = _scenario("mixture")
    n = sc.samples.shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    e = bpf.Engine(0)
    shard = Scenario.__new__(Scenario)
    shard.__dict__.update(sc.__dict__)
    shard.samples = np.ascontiguousarray(sc.samples[lo:hi])
    m, scn, pf, data = shard.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21, alpha=RECOVERY_ALPHA)
    pf.setResampleModel(resampler)
    pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
    sf = ShardedFilter(HipShardBackend(e, scn, pf, torch.device("cuda", 0)), dist, first_window=1024)
    assert sf.mailbox  # "auto": both ranks can map each other's mailbox, so the exchanges go through it
    recs = []
    for cycle in range(3):
        sf.update_sensor(bpf.PlanarData(_recovery_scan(sc, cycle), sc.angles, sc.range_max))
        sf.update_resample()
        st = sf.state()
        recs.append(dict(samples=pf.getCurrentSet().samples.copy(), M=st.sample_count, leaf=st.leaf_count,
                         rng=pf.getRngState(), w_slow=st.w_slow))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(recs, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()
    e.close()


@pytest.mark.parametrize("resampler", [0, 1])
def test_two_ranks_recovery_random_poses(tmp_path, resampler):
    """w_diff > 0 over shards: every rank resolves the same draw chain, shard 0 contributes the random free-space
    poses, the averages are reset on every shard; equal to the single engine (itself checked against the oracle
    in test_gpu_parity.py::test_recovery_random_poses_match_oracle)."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_recovery_worker, args=(2, port, str(tmp_path), resampler), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True) for r in range(2)]
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    orc, sc = _scenario("mixture")
    n = sc.samples.shape[0]
    e = bpf.Engine(0)
    m, scn, pf, data = sc.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21, alpha=RECOVERY_ALPHA)
    pf.setResampleModel(resampler)
    pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
    w_diffs = []
    for cycle in range(3):
        scn.updateSensor(pf, bpf.PlanarData(_recovery_scan(sc, cycle), sc.angles, sc.range_max))
        pf.updateResample()
        st = pf.getState()
        w_diffs.append(st.w_diff)
        cur = pf.getCurrentSet().samples
        r0, r1 = recs[0][cycle], recs[1][cycle]
        for r in (r0, r1):
            assert r["M"] == st.sample_count and r["leaf"] == st.leaf_count and r["rng"] == pf.getRngState()
            assert r["w_slow"] == st.w_slow
        merged = np.concatenate([r0["samples"], r1["samples"]])
        assert np.array_equal(merged[:, :3], cur[:, :3])
    assert max(w_diffs) > 0.01
    e.close()


STRESS_CYCLES = 16


def _stress_worker(rank, world, port, out_dir):
    """Many cycles with the two ranks deliberately out of step on the host (sleeps of a different length per rank
    and cycle, before the sensor update and before the resample), so that one rank's kernels regularly sit in a
    mailbox wait while the other has not even issued its post, and the fast rank runs a whole exchange ahead."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import badger_amcl_amd as bpf
    from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
    from scenario import Scenario
    orc, sc = _scenario("converged")
    n = sc.samples.shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    e = bpf.Engine(0)
    shard = Scenario.__new__(Scenario)
    shard.__dict__.update(sc.__dict__)
    shard.samples = np.ascontiguousarray(sc.samples[lo:hi])
    m, scn, pf, data = shard.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    sf = ShardedFilter(HipShardBackend(e, scn, pf, torch.device("cuda", 0)), dist, first_window=1024,
                       exchange="mailbox")
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    rs = np.random.RandomState(100 + rank)
    recs = []
    for cycle in range(STRESS_CYCLES):
        sf.update_action(od, bpf.OdomData(*ODATA))
        time.sleep(float(rs.uniform(0.0, 0.004)) if (cycle + rank) % 3 else 0.0)
        sf.update_sensor(data)
        time.sleep(float(rs.uniform(0.0, 0.004)) if (cycle + rank) % 2 else 0.0)
        sf.update_resample()
        st = sf.state()
        recs.append(dict(samples=pf.getCurrentSet().samples.copy(), M=st.sample_count, leaf=st.leaf_count,
                         rng=pf.getRngState(), miss=st.cdf_miss))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(recs, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()
    e.close()


def test_mailbox_exchange_with_ranks_out_of_step(tmp_path):
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_stress_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True) for r in range(2)]
    import badger_amcl_amd as bpf
    orc, sc = _scenario("converged")
    n = sc.samples.shape[0]
    e = bpf.Engine(0)
    m, scn, pf, data = sc.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    for cycle in range(STRESS_CYCLES):
        od.updateAction(pf, bpf.OdomData(*ODATA))
        scn.updateSensor(pf, data)
        pf.updateResample()
        st = pf.getState()
        r0, r1 = recs[0][cycle], recs[1][cycle]
        for r in (r0, r1):
            assert r["M"] == st.sample_count and r["leaf"] == st.leaf_count and r["rng"] == pf.getRngState()
            assert not r["miss"]
        merged = np.concatenate([r0["samples"], r1["samples"]])
        assert np.array_equal(merged[:, :3], pf.getCurrentSet().samples[:, :3]), cycle
    e.close()


def _silent_peer(handle_path, stop_path):
    """Creates a mailbox as rank 1 of 2, publishes its handle and then never takes part in an exchange."""
    sys.path.insert(0, ROOT)
    import ctypes as C
    import time
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    buf = (C.c_ubyte * 64)()
    e.check(e.lib.bpf_shard_mailbox_create(e.h, 1, 2, 1024, buf))
    with open(handle_path + ".tmp", "wb") as f:
        f.write(bytes(buf))
    os.rename(handle_path + ".tmp", handle_path)
    for _ in range(200):  # keep the allocation alive until the test is done (at most 20 s)
        if os.path.exists(stop_path):
            break
        time.sleep(0.1)
    e.close()


def test_mailbox_connect_gives_up_when_a_peer_never_answers(tmp_path):
    """The waits of the mailbox exchange are bounded: a rank whose peer maps nothing and posts nothing gets
    BPF_ERR_EXCHANGE from the connect round after 5 s instead of a kernel that spins for ever."""
    import ctypes as C
    import time
    import torch.multiprocessing as mp
    import badger_amcl_amd as bpf
    handle_path, stop_path = str(tmp_path / "handle"), str(tmp_path / "stop")
    ctx = mp.get_context("spawn")
    peer = ctx.Process(target=_silent_peer, args=(handle_path, stop_path))
    peer.start()
    try:
        for _ in range(300):
            if os.path.exists(handle_path):
                break
            time.sleep(0.1)
        assert os.path.exists(handle_path)
        theirs = open(handle_path, "rb").read()
        e = bpf.Engine(0)
        mine = (C.c_ubyte * 64)()
        e.check(e.lib.bpf_shard_mailbox_create(e.h, 0, 2, 1024, mine))
        t0 = time.time()
        rc = e.lib.bpf_shard_mailbox_connect(e.h, C.c_char_p(bytes(mine) + theirs))
        waited = time.time() - t0
        assert rc == 9, (rc, e.lib.bpf_last_error_message(e.h))  # BPF_ERR_EXCHANGE
        assert 4.0 < waited < 9.0, waited
        # the engine is still usable and the mailbox can be dropped
        assert e.lib.bpf_shard_mailbox_destroy(e.h) == 0
        e.close()
    finally:
        open(stop_path, "w").close()
        peer.join(30)
        if peer.is_alive():
            peer.kill()


def test_three_ranks_mailbox_equal_single_engine(tmp_path):
    """Three processes on cuda:0 exchanging through the mailbox (an odd world size: uneven shards, three slots per
    exchange), two motion -> sensor -> resample cycles, against one engine on the whole set."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    W = 3
    mp.spawn(_worker, args=(W, port, str(tmp_path), "converged", 8192, 0, "mailbox"), nprocs=W, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True) for r in range(W)]
    import badger_amcl_amd as bpf
    orc, sc = _scenario("converged")
    n = sc.samples.shape[0]
    e = bpf.Engine(0)
    m, scn, pf, data = sc.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    for cycle in range(2):
        od.updateAction(pf, bpf.OdomData(*ODATA))
        scn.updateSensor(pf, data)
        w_ref = pf.getCurrentSet().samples[:, 3].copy()
        pf.updateResample()
        st = pf.getState()
        rr = [recs[k][cycle] for k in range(W)]
        assert np.allclose(np.concatenate([r["w"][:, 3] for r in rr]), w_ref, rtol=1e-12, atol=0)
        for r in rr:
            assert r["M"] == st.sample_count and r["leaf"] == st.leaf_count and r["bins"] == st.bin_count
            assert r["rng"] == pf.getRngState() and not r["miss"]
        merged = np.concatenate([r["samples"] for r in rr])
        assert np.array_equal(merged[:, :3], pf.getCurrentSet().samples[:, :3])
        M = st.sample_count
        for k in range(W):
            assert rr[k]["samples"].shape[0] == (M * (k + 1)) // W - (M * k) // W
    e.close()


def _stall_worker(rank, world, port, out_dir, stall_rank):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import badger_amcl_amd as bpf
    from badger_amcl_amd.sharded import HipShardBackend, ShardedFilter
    from scenario import Scenario
    orc, sc = _scenario("converged")
    n = sc.samples.shape[0]
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    e = bpf.Engine(0)
    shard = Scenario.__new__(Scenario)
    shard.__dict__.update(sc.__dict__)
    shard.samples = np.ascontiguousarray(sc.samples[lo:hi])
    m, scn, pf, data = shard.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    b = HipShardBackend(e, scn, pf, torch.device("cuda", 0))
    sf = ShardedFilter(b, dist, first_window=1024, exchange="mailbox", mailbox_timeout_ms=700)
    assert sf.mailbox
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    recs = []
    for cycle in range(3):
        if cycle == 1 and rank == stall_rank:
            time.sleep(2.5)  # a host stall well past the mailbox bound: the peer's wait for this rank's total runs out
        sf.update_action(od, bpf.OdomData(*ODATA))
        sf.update_sensor(data)
        sf.update_resample()
        st = sf.state()
        recs.append(dict(samples=pf.getCurrentSet().samples.copy(), M=st.sample_count, leaf=st.leaf_count,
                         bins=st.bin_count, rng=pf.getRngState(), recoveries=sf.recoveries, mailbox=sf.mailbox))
    np.save(os.path.join(out_dir, "stall%d.npy" % rank), np.array(recs, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()
    e.close()


@pytest.mark.timeout(240)
def test_a_rank_stalling_past_the_mailbox_bound_does_not_kill_the_filter(tmp_path):
    """a mailbox time-out used to be permanent and one-sided.  Now the consumer kernel of a failed wait
    leaves its data alone, every rank finds out within its own bound, they meet, finish the interrupted update over
    the collectives (normalising where that was still due, running the resample again), and set the mailbox up
    again.  Rank 1 sleeps 2.5 s before its second cycle with a 0.7 s bound; all three cycles must equal the
    undisturbed single-engine run."""
    import torch.multiprocessing as mp
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_stall_worker, args=(2, port, str(tmp_path), 1), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "stall%d.npy" % r), allow_pickle=True) for r in range(2)]
    import badger_amcl_amd as bpf
    orc, sc = _scenario("converged")
    n = sc.samples.shape[0]
    e = bpf.Engine(0)
    m, scn, pf, data = sc.gpu_objects(e, 181, "lf", min_samples=100, max_samples=n, seed=21)
    od = bpf.Odom(e)
    od.setModel(*ODOM)
    for cycle in range(3):
        od.updateAction(pf, bpf.OdomData(*ODATA))
        scn.updateSensor(pf, data)
        pf.updateResample()
        st = pf.getState()
        cur = pf.getCurrentSet()
        r0, r1 = recs[0][cycle], recs[1][cycle]
        for r in (r0, r1):
            assert (r["M"], r["leaf"], r["bins"], r["rng"]) == (st.sample_count, st.leaf_count, st.bin_count,
                                                                 pf.getRngState()), cycle
        merged = np.concatenate([r0["samples"], r1["samples"]])
        assert np.array_equal(merged[:, :3], cur.samples[:, :3]), cycle
    assert recs[0][2]["recoveries"] == 1 and recs[1][2]["recoveries"] == 1
    assert recs[0][0]["recoveries"] == 0
    assert recs[0][2]["mailbox"] and recs[1][2]["mailbox"]  # set up again after the step
    e.close()
