"""SURVEY.md section 8(f) rows next-2 / next-3 on the GPU box: cluster statistics of the resident set
and the reference-order brushfire LUT builder, both through the C-ABI, against the oracle."""
import numpy as np
import pytest

from scenario import Scenario
from badger_amcl_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


def _oracle_stats(orc, samples, max_clusters):
    t = orc.KDTree()
    for k in range(samples.shape[0]):
        t.insert_pose(samples[k, :3], samples[k, 3])
    return t.cluster_stats(samples, max_clusters)


def _assert_stats_equal(pf, want):
    n, mean, cov = pf.computeClusterStats()
    assert n == want["n"]
    assert np.array_equal(mean, want["set_mean"])
    assert np.array_equal(cov, want["set_cov"], equal_nan=True)
    for k in range(n):
        w, m, cnt, c = pf.getClusterStats(k)
        assert cnt == want["count"][k]
        assert w == want["weight"][k]
        assert np.array_equal(m, want["mean"][k])
        assert np.array_equal(c, want["cov"][k], equal_nan=True)
    assert pf.getClusterStats(n) is None
    best_w, best_pose = pf.getMaxWeightPose()
    if n:
        k = int(np.argmax(want["weight"]))  # first maximum, like the strict '>' scan of node_2d.cpp:608
        assert best_w == want["weight"][k]
        assert np.array_equal(best_pose, want["mean"][k])


def test_cluster_stats_of_loaded_multimodal_set(engine, orc):
    """Three separated blobs + stragglers: several clusters, bit-exact counts / weights / means / covs."""
    sc_ = Scenario(orc, size=400, n=3000, beams=61, cloud="converged")
    pose = sc_.pose
    blobs = [synth.converged_cloud(1000, pose + off, seed=5 + i, sigma=(0.15, 0.15, 0.05))
             for i, off in enumerate([(0, 0, 0), (4.0, -2.0, 1.0), (-3.0, 3.5, -2.0)])]
    s = np.ascontiguousarray(np.concatenate(blobs))
    s[:, 3] = np.random.default_rng(9).uniform(0.5, 1.5, s.shape[0])
    s[:, 3] /= s[:, 3].sum()
    sc_.samples = s
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", min_samples=100, seed=3)
    want = _oracle_stats(orc, s, s.shape[0])
    assert want["n"] >= 3
    _assert_stats_equal(pf, want)
    # cached: a second query without a change of the set gives the same answer
    _assert_stats_equal(pf, want)


@pytest.mark.parametrize("resampler", [0, 1])
def test_cluster_stats_after_update_and_resample(engine, orc, resampler):
    """The statistics the node reads after updateResample (particle_filter.cpp:464-468): the engine's
    histogram tree of the resampled set is reused for the labelling."""
    sc_ = Scenario(orc, size=400, n=4000, beams=91, cloud="mixture")
    m, sc, pf, data = sc_.gpu_objects(engine, 91, "lf", min_samples=100, seed=13)
    pf.setResampleModel(resampler)
    sc.updateSensor(pf, data)
    # weighted, not yet resampled: tree from initWithSamples
    cur = pf.getCurrentSet().samples
    _assert_stats_equal(pf, _oracle_stats(orc, cur, 4000))
    pf.updateResample()
    cur = pf.getCurrentSet().samples
    _assert_stats_equal(pf, _oracle_stats(orc, cur, 4000))
    # restore() invalidates the tree; the statistics rebuild it from the set
    pf.snapshot()
    sc.updateSensor(pf, data)
    pf.updateResample()
    pf.restore()
    _assert_stats_equal(pf, _oracle_stats(orc, cur, 4000))


def test_reference_brushfire_lut_is_bit_identical(engine, orc):
    """bpf_map2d_build_distances_lut_reference == OccupancyMap::updateDistancesLUT (oracle restatement,
    itself pinned in test_oracle_pins.py), including the priority-queue tie order."""
    import badger_amcl_amd as bpf
    for size, max_dist in [(200, 2.0), (333, 0.7)]:
        cells, origin = synth.make_map(size)
        rng = np.random.default_rng(size)
        cells[rng.random(cells.shape) < 0.002] = 1   # scattered obstacles: many equidistant ties
        want = orc.OccupancyMap(cells, 0.05, origin).update_distances_lut(max_dist)
        m = bpf.OccupancyMap(engine, 0.05)
        m.setCells(cells)
        m.setOrigin(origin)
        m.updateDistancesLUTReference(max_dist)
        got = m.getDistancesLUT()
        assert np.array_equal(got.reshape(-1), np.asarray(want, dtype=np.float32).reshape(-1))


def test_octomap_lut_builder_matches_oracle(engine, orc):
    """bpf_map3d_build_distances_lut == OctoMap::updateDistancesLUT (oracle restatement): same column
    placement (octree leaf order = list order) and the same quantised distances, byte for byte."""
    import badger_amcl_amd as bpf
    occ = synth.box_room_voxels(lo=(-12, -9, -2), hi=(12, 9, 6))
    rng = np.random.default_rng(4)
    occ = occ[rng.permutation(occ.shape[0])]          # an arbitrary "leaf iteration" order
    extra = np.array([[100, 0, 0], [0, -50, 1]], dtype=np.int32)  # outside the cropped bounds: skipped
    occ = np.ascontiguousarray(np.concatenate([occ[:50], extra, occ[50:]]))
    mn, mx = (-14, -11, -3), (14, 11, 8)
    for res, max_dist in [(0.05, 0.3), (0.1, 0.45)]:
        want = orc.OctoMapLUT(mn, mx, res, max_dist)
        want.build(occ)
        om = bpf.OctoMap(engine, res)
        om.updateDistancesLUT(occ, mn, mx, max_dist)
        pi, dr = om.getDistancesLUT()
        assert np.array_equal(pi, want.pose_indices)
        assert np.array_equal(dr, want.distance_ratios)
