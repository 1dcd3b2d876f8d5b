"""world_size-2 gloo test of the sharded filter logic (no GPU): two ranks, each holding half of
the particle set, must reproduce what one process computes on the whole set."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


ODOM = (2, (0.05, 0.04, 0.03, 0.02, 0.0))                       # diff-corrected
ODATA = ((1.0, 2.0, 0.3), (0.03, -0.01, 0.02), (0.03, 0.01, 0.02))  # pose, delta, absolute motion


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scenario():
    from oracle import pyoracle as orc
    from scenario import Scenario
    return orc, Scenario(orc, size=200, n=1200, beams=61, cloud="mixture")


def _worker(rank, world, port, out_dir, cloud_split, resampler):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from badger_amcl_amd.sharded import ShardedFilter
    from shard_backends import OracleShardBackend
    orc, sc = _scenario()
    n = sc.samples.shape[0]
    lo, hi = cloud_split[rank], cloud_split[rank + 1]
    planar = sc.oracle_planar(61, "lf")
    b = OracleShardBackend(orc, sc.omap, planar, sc.samples[lo:hi], 100, n, seed=9)
    b._resample_model = resampler
    sf = ShardedFilter(b, dist, first_window=256)
    data = (sc.ranges, sc.angles, sc.range_max)
    records = []
    for cycle in range(2):
        sf.update_action(ODOM, ODATA)
        sf.update_sensor(data)
        w_after = b.samples.copy()
        sf.update_resample()
        st = sf.state()
        records.append(dict(w=w_after, samples=b.samples.copy(), M=st.sample_count, leaf=st.leaf_count,
                            bins=st.bin_count, rng=b.rng_state(), conv=st.converged, windows=st.windows,
                            w_slow=st.w_slow))
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), np.array(records, dtype=object), allow_pickle=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("split,resampler", [((0, 600, 1200), 0), ((0, 137, 1200), 0), ((0, 500, 1200), 1)])
def test_two_shards_equal_one_filter(tmp_path, split, resampler):
    sys.path.insert(0, HERE)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), split, resampler), nprocs=2, join=True)
    recs = [np.load(os.path.join(str(tmp_path), "rank%d.npy" % r), allow_pickle=True) for r in range(2)]

    orc, sc = _scenario()
    n = sc.samples.shape[0]
    opf = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0, seed=9)
    opf.set_samples(sc.samples)
    opf.set_resample_model(resampler)
    p = sc.oracle_planar(61, "lf")
    for cycle in range(2):
        opf.pf.rng = orc.odom_update_action(ODOM[0], ODOM[1], *ODATA, opf.samples[:opf.sample_count], opf.pf.rng)
        opf.update_sensor(lambda s, conv: sc.oracle_apply(p, s, conv))
        w_ref = opf.samples[:opf.sample_count, 3].copy()
        out = opf.update_resample()
        r0, r1 = recs[0][cycle], recs[1][cycle]
        # normalised weights: rank-ordered total vs the serial total differ by rounding only
        w_sh = np.concatenate([r0["w"][:, 3], r1["w"][:, 3]])
        assert np.allclose(w_sh, w_ref, rtol=1e-12, atol=0)
        for r in (r0, r1):
            assert r["M"] == out.sample_count
            assert r["leaf"] == out.leaf_count and r["bins"] == out.node_count
            assert r["rng"] == opf.pf.rng
            assert r["conv"] == out.converged
            assert abs(r["w_slow"] - opf.pf.w_slow) <= 1e-12 * opf.pf.w_slow
        M = out.sample_count
        merged = np.concatenate([r0["samples"], r1["samples"]])
        assert merged.shape[0] == M
        assert np.array_equal(merged[:, :3], opf.samples[:M, :3])
        assert np.all(merged[:, 3] == 1.0 / M)
        # shards are the even, index-ordered split
        assert r0["samples"].shape[0] == M // 2 and r1["samples"].shape[0] == M - M // 2
    if resampler == 0:
        assert recs[0][0]["windows"] >= 2  # first_window=256 forces the multi-window path
