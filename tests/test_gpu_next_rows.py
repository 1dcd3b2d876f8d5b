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


def _close(a, b, rtol=1e-12, atol=1e-12):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


def _assert_stats_equal(pf, want, exact=True, set_atol=1e-12):
    """exact: the host evaluation (BPF_OPT_STATS_HOST), the reference's serial order bit for bit.  Otherwise the device
    evaluation: labels, counts and the choice of the heaviest cluster exact; sums differ from the reference's serial
    double chain by summation rounding only (budget 1e-12, relative for weights and means, absolute for the
    covariances, which are differences of numbers of order x^2)."""
    n, mean, cov = pf.computeClusterStats()
    assert n == want["n"]
    if exact:
        assert np.array_equal(mean, want["set_mean"])
        assert np.array_equal(cov, want["set_cov"], equal_nan=True)
    else:
        assert _close(mean, want["set_mean"], atol=set_atol) and _close(cov, want["set_cov"], atol=max(1e-10, set_atol))
    for k in range(n):
        w, m, cnt, c = pf.getClusterStats(k)
        assert cnt == want["count"][k]
        if exact:
            assert w == want["weight"][k]
            assert np.array_equal(m, want["mean"][k])
            assert np.array_equal(c, want["cov"][k], equal_nan=True)
        else:
            assert _close(w, want["weight"][k]) and _close(m, want["mean"][k])
            assert _close(c, want["cov"][k], atol=1e-10)
    assert pf.getClusterStats(n) is None
    best_w, best_pose = pf.getMaxWeightPose()
    if n:
        k = int(np.argmax(want["weight"]))  # first maximum, like the strict '>' scan of node_2d.cpp:608
        if exact:
            assert best_w == want["weight"][k]
            assert np.array_equal(best_pose, want["mean"][k])
        else:
            ws = np.sort(want["weight"])[::-1]
            if ws.size < 2 or ws[0] - ws[1] > 1e-12 * ws[0]:  # (a tie at rounding level may resolve either way)
                assert _close(best_w, want["weight"][k]) and _close(best_pose, want["mean"][k])


@pytest.fixture(params=["device", "host"])
def stats_mode(request, engine):
    import badger_amcl_amd.pf as hpf
    engine.set_option(hpf.OPT_STATS_HOST, 1 if request.param == "host" else 0)
    yield request.param == "host"
    engine.set_option(hpf.OPT_STATS_HOST, 0)


def test_cluster_stats_of_loaded_multimodal_set(engine, orc, stats_mode):
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
    _assert_stats_equal(pf, want, stats_mode)
    # cached: a second query without a change of the set gives the same answer
    _assert_stats_equal(pf, want, stats_mode)


@pytest.mark.parametrize("resampler", [0, 1])
def test_cluster_stats_after_update_and_resample(engine, orc, resampler, stats_mode):
    """The statistics the node reads after updateResample (particle_filter.cpp:464-468): the engine's
    histogram tree of the resampled set is reused for the labelling."""
    sc_ = Scenario(orc, size=400, n=4000, beams=91, cloud="mixture")
    m, sc, pf, data = sc_.gpu_objects(engine, 91, "lf", min_samples=100, seed=13)
    pf.setResampleModel(resampler)
    sc.updateSensor(pf, data)
    # weighted, not yet resampled: tree from initWithSamples
    cur = pf.getCurrentSet().samples
    _assert_stats_equal(pf, _oracle_stats(orc, cur, 4000), stats_mode)
    pf.updateResample()
    cur = pf.getCurrentSet().samples
    _assert_stats_equal(pf, _oracle_stats(orc, cur, 4000), stats_mode)
    # restore() invalidates the tree; the statistics rebuild it from the set
    pf.snapshot()
    sc.updateSensor(pf, data)
    pf.updateResample()
    pf.restore()
    _assert_stats_equal(pf, _oracle_stats(orc, cur, 4000), stats_mode)


def test_cluster_stats_on_device_full_size_spread_set(engine, orc):
    """100 000 spread particles after scoring and resampling (no early stop: M = 100 000, thousands
    of clusters): the device evaluation against the oracle's serial one -- cluster count, every cluster's label (via
    its count), weights / means / covariances within the rounding budget, the heaviest cluster -- without a copy of
    the set to the host (bpf_pf_get_max_weight_pose reads one result block).  Two runs give the same bits."""
    sc_ = Scenario(orc, size=2000, n=100000, beams=181, cloud="spread")
    m, sc, pf, data = sc_.gpu_objects(engine, 181, "lf", min_samples=100, seed=21)
    sc.updateSensor(pf, data)
    cur = pf.getCurrentSet().samples
    want = _oracle_stats(orc, cur, 100000)
    assert want["n"] > 500
    # set-level sums run over all 100 000 samples: the reference's serial double chain itself carries up to n eps of
    # rounding there (1e-8 absolute on a covariance of 800 m^2; the device sum is exact, checked against math.fsum
    # below), and the circular mean of uniformly spread headings divides it by a resultant of ~1e-3 -- hence the
    # wider budget against the ORACLE for the set's own mean / covariance
    _assert_stats_equal(pf, want, exact=False, set_atol=1e-7)
    # ... and the device's set statistics ARE the exactly summed ones: against math.fsum of the same double terms
    import math
    w, x, y = cur[:, 3], cur[:, 0], cur[:, 1]
    W = math.fsum(w)
    mx, my = math.fsum(w * x) / W, math.fsum(w * y) / W
    exact_cov = [math.fsum(w * x * x) / W - mx * mx, math.fsum(w * x * y) / W - mx * my,
                 math.fsum(w * y * x) / W - my * mx, math.fsum(w * y * y) / W - my * my]
    _, mean_d, cov_d = pf.computeClusterStats()
    assert abs(mean_d[0] - mx) <= 1e-13 * abs(mx) and abs(mean_d[1] - my) <= 1e-13 * abs(my)
    assert np.allclose(cov_d[:4], exact_cov, rtol=0, atol=1e-11)
    first = pf.getMaxWeightPose(), pf.computeClusterStats()
    pf.updateResample()
    assert pf.getState().sample_count == 100000
    cur = pf.getCurrentSet().samples
    want = _oracle_stats(orc, cur, 100000)
    _assert_stats_equal(pf, want, exact=False, set_atol=1e-7)
    # order-independent accumulation: the same bits when evaluated again from scratch
    pf.snapshot()
    pf.restore()
    again_w, again_pose = pf.getMaxWeightPose()
    n2, mean2, cov2 = pf.computeClusterStats()
    n1, mean1, cov1 = pf.computeClusterStats()
    assert n1 == n2 and np.array_equal(mean1, mean2) and np.array_equal(cov1, cov2)
    assert first[1][0] > 0 and again_w > 0


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


def test_reference_named_calls_give_the_reference_lut_by_default(engine, orc):
    """OccupancyMap.updateDistancesLUT and the implicit build inside setModelLikelihoodField (the reference calls
    map_->updateDistancesLUT there, planar_scanner.cpp:74) leave the brushfire's values behind, bit for bit; the
    exact EDT is only what its own name (updateDistancesLUTExact / BPF_OPT_LUT_EXACT_EDT) asks for."""
    import badger_amcl_amd as bpf
    cells, origin = synth.make_map(240)
    rng = np.random.default_rng(77)
    cells[rng.random(cells.shape) < 0.002] = 1
    want = np.asarray(orc.OccupancyMap(cells, 0.05, origin).update_distances_lut(1.3), dtype=np.float32).reshape(-1)

    def fresh():
        m = bpf.OccupancyMap(engine, 0.05)
        m.setCells(cells)
        m.setOrigin(origin)
        return m

    m = fresh()
    m.updateDistancesLUT(1.3)
    assert np.array_equal(m.getDistancesLUT().reshape(-1), want)
    # implicit: no LUT yet for this max distance when the model is set
    m = fresh()
    sc = bpf.PlanarScanner(engine)
    sc.init(30, m)
    sc.setModelLikelihoodField(0.95, 0.05, 0.2, 1.3)
    assert np.array_equal(m.getDistancesLUT().reshape(-1), want)
    # the explicitly named fast builders: exact EDT, never above the brushfire, different somewhere on this map
    m = fresh()
    m.updateDistancesLUTExact(1.3)
    edt = m.getDistancesLUT().reshape(-1)
    assert np.all(edt <= want) and np.mean(edt == want) > 0.97 and not np.array_equal(edt, want)
    engine.set_option(bpf.pf.OPT_LUT_EXACT_EDT, 1)
    try:
        m = fresh()
        sc.init(30, m)
        sc.setModelLikelihoodField(0.95, 0.05, 0.2, 1.3)
        assert np.array_equal(m.getDistancesLUT().reshape(-1), edt)
    finally:
        engine.set_option(bpf.pf.OPT_LUT_EXACT_EDT, 0)


@pytest.mark.parametrize("host", [0, 1])
def test_octomap_lut_builder_matches_oracle(engine, orc, host):
    """bpf_map3d_build_distances_lut == OctoMap::updateDistancesLUT (oracle restatement): same column
    placement (octree leaf order = list order) and the same quantised distances, byte for byte -- on the device
    (the FIFO brushfire replayed generation by generation, kernels_lut3d.hpp) and on the host."""
    import ctypes as C
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    occ = synth.box_room_voxels(lo=(-12, -9, -2), hi=(12, 9, 6))
    rng = np.random.default_rng(4)
    occ = occ[rng.permutation(occ.shape[0])]          # an arbitrary "leaf iteration" order
    extra = np.array([[100, 0, 0], [0, -50, 1]], dtype=np.int32)  # outside the cropped bounds: skipped
    occ = np.ascontiguousarray(np.concatenate([occ[:50], extra, occ[50:], occ[:7]]))  # and a few voxels twice
    mn, mx = (-14, -11, -3), (14, 11, 8)
    engine.set_option(hpf.OPT_LUT_HOST, host)
    try:
        for res, max_dist in [(0.05, 0.3), (0.1, 0.45), (0.05, 0.52)]:
            want = orc.OctoMapLUT(mn, mx, res, max_dist)
            want.build(occ)
            om = bpf.OctoMap(engine, res)
            om.updateDistancesLUT(occ, mn, mx, max_dist)
            pi, dr = om.getDistancesLUT()
            assert np.array_equal(pi, want.pose_indices)
            assert np.array_equal(dr, want.distance_ratios)
            g = C.c_int()
            engine.check(engine.lib.bpf_map3d_builder_generations(engine.h, C.byref(g)))
            assert (g.value == 0) == (host == 1) and (host == 1 or g.value >= 4)
    finally:
        engine.set_option(hpf.OPT_LUT_HOST, 0)


def test_octomap_lut_builder_scattered_obstacles_and_an_empty_map(engine, orc):
    """Random scattered voxels (many sources competing for every cell, cells improved several times within one
    generation, columns first touched in every generation) in a 60 x 50 x 20 volume, max_dist of 8 cells; and a map
    without any obstacle (the LUT is the shared all-255 column alone)."""
    import badger_amcl_amd as bpf
    rng = np.random.default_rng(11)
    mn, mx = (-30, -25, -4), (29, 24, 15)
    occ = np.stack([rng.integers(mn[d], mx[d] + 1, 900) for d in range(3)], axis=1).astype(np.int32)
    for res, max_dist in [(0.05, 0.4), (0.2, 1.0)]:
        want = orc.OctoMapLUT(mn, mx, res, max_dist)
        want.build(occ)
        om = bpf.OctoMap(engine, res)
        om.updateDistancesLUT(occ, mn, mx, max_dist)
        pi, dr = om.getDistancesLUT()
        assert np.array_equal(pi, want.pose_indices) and np.array_equal(dr, want.distance_ratios)
    om = bpf.OctoMap(engine, 0.05)
    om.updateDistancesLUT(np.zeros((0, 3), dtype=np.int32), mn, mx, 0.3)
    pi, dr = om.getDistancesLUT()
    assert not pi.any() and dr.size == mx[2] - mn[2] + 1 and np.all(dr == 255)


def test_messages_in_poses_out_against_the_oracle(engine, orc):
    """SURVEY 8(f) next-4, end to end on the GPU: an OccupancyGrid and a LaserScan as the node receives them go
    through the product's wire shaping (bpf_wire_*) into the engine -- map from the converted cells, LUT built in the
    reference's order, sensor update, resample -- and the resampled set comes out as a PoseArray; the same messages
    through the oracle's restatements of Node2D::convertMap / getAngleStats / updateLatestScanData
    (node_2d.cpp:265-295,497-560), its brushfire, scoring and resampler, and Node::publishParticleCloud
    (node.cpp:335-357).  Cells, bearings, ranges exact; weights 1e-9; resampled poses and the PoseArray exact."""
    import math
    import badger_amcl_amd as bpf
    from badger_amcl_amd import wire
    # the messages: a 200 x 200 grid at 0.1 m scaled up by 2 -> the usual 400 x 400 at 0.05 m; a 181-beam scan from an
    # upside-down scanner yawed by 0.2 rad, with short readings and a sensor range limit below the message's
    cells400, origin = synth.make_map(400)
    coarse = cells400[::2, ::2]
    data = np.where(coarse == -1, 0, np.where(coarse == 1, 100, -1)).astype(np.int8).reshape(-1)
    msg_origin = (-3.0, 1.5)
    cells, org, res = wire.occupancy_grid_to_cells(data, 200, 200, 0.1, msg_origin[0], msg_origin[1], 2)
    ocells, oorg, ores = orc.wire_convert_map(data, 200, 200, 0.1, msg_origin[0], msg_origin[1], 2)
    assert np.array_equal(cells, ocells) and (org[0], org[1], res) == (oorg[0], oorg[1], ores)
    q_mount = (math.cos(0.1), math.sin(0.1), 0.0, 0.0)  # roll pi (upside down) composed with yaw 0.2
    a0, da = wire.scan_angle_stats(-1.5, 3.0 / 180, q_mount)
    assert (a0, da) == orc.wire_scan_angle_stats(-1.5, 3.0 / 180, q_mount)
    pose = np.array([msg_origin[0] + 10.1, msg_origin[1] + 10.0, 0.3])
    true_ranges, _ = synth.cast_scan(cells, org, res, pose, 181, seed=4)
    scan = true_ranges[::-1].astype(np.float32).copy()  # the mirrored sweep of an upside-down scanner
    scan[::23] = np.float32(0.03)
    ranges, angles, rmax = wire.laserscan_to_planar(scan, np.float32(0.05), np.float32(40.0), a0, da, 0.1, 25.0)
    oranges, oangles, ormax = orc.wire_laserscan_to_planar(scan, np.float32(0.05), np.float32(40.0), a0, da, 0.1, 25.0)
    assert np.array_equal(ranges, oranges) and np.array_equal(angles, oangles) and rmax == ormax == 25.0
    # the engine on the product's shaping
    n = 3000
    samples = synth.converged_cloud(n, pose, seed=12)
    samples[:, 3] *= np.random.default_rng(13).uniform(0.5, 1.5, n)
    m = bpf.OccupancyMap(engine, res)
    m.setCells(cells)
    m.setOrigin(org)
    m.updateDistancesLUTReference(2.0)
    sc = bpf.PlanarScanner(engine)
    sc.init(181, m)
    sc.setModelLikelihoodField(0.95, 0.05, 0.2, 2.0)
    sc.setMapFactors(*synth.MAP_FACTORS)
    sc.setPlanarScannerPose((0.1, 0.0, 0.0))
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
    pf.srand48(3)
    pf.initWithSamples(samples)
    assert sc.updateSensor(pf, bpf.PlanarData(ranges, angles, rmax))
    got_w = pf.getCurrentSet().samples[:, 3].copy()
    pf.updateResample()
    got = pf.getCurrentSet().samples
    got_msg = wire.samples_to_pose_array(got)
    # the oracle on its own shaping
    omap = orc.OccupancyMap(ocells, ores, oorg)
    omap.update_distances_lut(2.0)
    assert np.array_equal(m.getDistancesLUT().reshape(-1), omap.lut.reshape(-1))
    p = orc.planar(orc.MODEL_LF, 181, scanner_pose=(0.1, 0.0, 0.0), off_map_factor=synth.MAP_FACTORS[0],
                   non_free_space_factor=synth.MAP_FACTORS[1], non_free_space_radius=synth.MAP_FACTORS[2],
                   **synth.LF_DEFAULTS)
    opf = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0, seed=3)
    opf.set_samples(samples)
    opf.update_sensor(lambda s, conv: orc.planar_apply(p, omap, s, oranges, oangles, ormax, conv))
    rel = np.abs(got_w - opf.samples[:n, 3]) / opf.samples[:n, 3]
    assert (rel > 1e-9).sum() <= 1
    out = opf.update_resample()
    M = out.sample_count
    assert pf.getState().sample_count == M
    assert np.array_equal(got[:, :3], opf.samples[:M, :3])
    want_msg = orc.wire_pose_array(opf.samples[:M])
    # positions exact; the quaternion is sin / cos of half the yaw from the host's libm, where a compiler may call
    # sincos for the pair (glibc's sincos and its sin / cos differ by an ulp on some CPUs)
    assert np.array_equal(got_msg[:, :5], want_msg[:, :5])
    assert np.allclose(got_msg[:, 5:], want_msg[:, 5:], rtol=0, atol=3e-16)


def test_cluster_stats_small_sets_one_block_and_beyond_its_limits(engine, orc):
    """Sets of at most 4096 samples take the single-block kernel (k_stats_block); one with more than 64 clusters
    (3 000 particles spread over the map: hundreds of one-bin clusters) is handed on to the general device path.  Both
    against the oracle, and a one-sample set."""
    sc_ = Scenario(orc, size=400, n=3000, beams=61, cloud="spread")
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", min_samples=100, seed=3)
    sc.updateSensor(pf, data)
    cur = pf.getCurrentSet().samples
    want = _oracle_stats(orc, cur, 3000)
    assert want["n"] > 64
    _assert_stats_equal(pf, want, exact=False)
    one = np.array([[3.0, 4.0, 0.5, 1.0]])
    pf2 = __import__("badger_amcl_amd").ParticleFilter(engine, 1, 10, 0.0, 0.0, 85.0)
    pf2.initWithSamples(one)
    _assert_stats_equal(pf2, _oracle_stats(orc, one, 10), exact=False)


def _tree_counts(engine, orc, samples):
    import badger_amcl_amd as bpf
    n = samples.shape[0]
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
    pf.initWithSamples(samples)
    st = pf.getState()
    return st.leaf_count, st.bin_count, engine.kld_last_form()


@pytest.mark.parametrize("kind", ["spread", "half", "line_then_blob", "small"])
def test_histogram_tree_in_lds_pieces_equals_the_level_loop_and_the_oracle(engine, orc, kind):
    """The device-side histogram tree of a long key stream (the set's PFKDTree, pf_kdtree.cpp:49-150) grown in
    LDS-sized pieces (kernels_kld2.hpp) against the level-per-launch form and the oracle's tree: same leaf and bin
    counts.  `line_then_blob` puts every later key below ONE node of the top tree (a bucket no block can hold): the
    pieces decline and the level loop takes over."""
    import badger_amcl_amd as bpf
    rng = np.random.default_rng(11)
    n = 60000
    s = np.zeros((n, 4))
    s[:, 3] = 1.0 / n
    if kind == "spread":
        s[:, 0] = rng.uniform(0, 100, n); s[:, 1] = rng.uniform(0, 100, n); s[:, 2] = rng.uniform(-3.1, 3.1, n)
    elif kind == "half":
        h = n // 2
        s[:h, 0] = rng.normal(50, 0.2, h); s[:h, 1] = rng.normal(50, 0.2, h); s[:h, 2] = rng.normal(0.3, 0.05, h)
        s[h:, 0] = rng.uniform(0, 100, n - h); s[h:, 1] = rng.uniform(0, 100, n - h)
        s[h:, 2] = rng.uniform(-3.1, 3.1, n - h)
        s[:] = s[rng.permutation(n)]
    elif kind == "line_then_blob":
        k = 2300  # more distinct bins than the top tree holds, all on a line far from the rest
        s[:k, 0] = -500.0 - 0.5 * np.arange(k); s[:k, 1] = -500.0; s[:k, 2] = 0.0
        s[k:, 0] = rng.uniform(0, 100, n - k); s[k:, 1] = rng.uniform(0, 100, n - k)
        s[k:, 2] = rng.uniform(-3.1, 3.1, n - k)
    else:
        n = 9000  # fewer tree keys than one block's table: the top tree is the whole tree
        s = s[:n]
        s[:, 0] = rng.uniform(0, 20, n); s[:, 1] = rng.uniform(0, 20, n); s[:, 2] = rng.uniform(-1, 1, n)
        s[:, 3] = 1.0 / n
    otree = orc.ParticleFilter(100, s.shape[0], 0.0, 0.0, 85.0, seed=1)
    otree.set_samples(s)
    want_leaf = otree.leaf_count
    got = {}
    for local in (1, 0):
        engine.set_option(bpf.pf.OPT_KLD_LOCAL, local)
        try:
            got[local] = _tree_counts(engine, orc, s)
        finally:
            engine.set_option(bpf.pf.OPT_KLD_LOCAL, 1)
    assert got[1][:2] == got[0][:2] and got[1][0] == want_leaf
    assert got[0][2] == 1
    assert got[1][2] == (1 if kind == "line_then_blob" else 2)


def test_histogram_tree_tables_shared_with_cluster_statistics_are_clean_again(engine, orc):
    """The pieces form leaves the hash tables cleared BEHIND its result for the next build; the cluster statistics use
    the same tables in between (abi_statistics.inl).  Three builds of different sets with statistics between them: every
    leaf count equals the oracle's tree."""
    import badger_amcl_amd as bpf
    rng = np.random.default_rng(23)
    n = 40000
    for k in range(3):
        s = np.zeros((n, 4))
        s[:, 0] = rng.uniform(0, 60 + 20 * k, n); s[:, 1] = rng.uniform(0, 80, n); s[:, 2] = rng.uniform(-3.1, 3.1, n)
        s[:, 3] = 1.0 / n
        otree = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0, seed=1)
        otree.set_samples(s)
        pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
        pf.initWithSamples(s)
        st = pf.getState()
        assert engine.kld_last_form() == 2
        assert st.leaf_count == otree.leaf_count
        want = _oracle_stats(orc, s, n)
        # (runs on the tables the tree has just left cleared; the set spans 80 m: its covariance sums carry x^2 ~ 6e3)
        _assert_stats_equal(pf, want, exact=False, set_atol=1e-8)
