"""BASELINE.json full sizes (100 k particles, 1081 beams, 2000^2 map) through size-independent properties, and
the edge cases the path has: the oracle takes minutes at these sizes, so the checks are
product-vs-product identities (two independent product paths must agree exactly), conservation laws and
oracle spot checks on slices."""
import numpy as np
import pytest

from badger_amcl_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def world(engine):
    import badger_amcl_amd as bpf
    size, beams = 2000, 1081
    cells, origin = synth.make_map(size)
    pose = synth.true_pose(size)
    ranges, angles = synth.cast_scan(cells, origin, 0.05, pose, beams, seed=5)
    m = bpf.OccupancyMap(engine, 0.05)
    m.setCells(cells)
    m.setOrigin(origin)
    m.updateDistancesLUT(2.0)
    sc = bpf.PlanarScanner(engine)
    sc.init(beams, m)
    sc.setModelLikelihoodField(0.95, 0.05, 0.2, 2.0)
    sc.setMapFactors(*synth.MAP_FACTORS)
    sc.setPlanarScannerPose(synth.SCANNER_POSE)
    return dict(size=size, pose=pose, sc=sc, data=bpf.PlanarData(ranges, angles, 30.0), map=m)


def test_spread_cloud_device_tree_equals_host_replay(engine, world, orc):
    """100 k spread particles never stop early: the KLD stop rule runs as the device-side level-synchronous tree.
    Forcing the host's ordered replay instead must give the identical set, counts and RNG state."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    n = 100000
    s = synth.spread_cloud(n, world["size"], seed=43)
    res = {}
    for mode, dmin in [("device", 8192), ("host", 0)]:
        engine.set_option(hpf.OPT_KLD_DEVICE_MIN, dmin)
        try:
            pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
            pf.srand48(11)
            pf.initWithSamples(s, leaf_count=1)
            world["sc"].updateSensor(pf, world["data"])
            pf.updateResample()
            st = pf.getState()
            res[mode] = (st.sample_count, st.leaf_count, st.bin_count, pf.getRngState(), pf.getCurrentSet().samples,
                         st.kld_on_device)
        finally:
            engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 8192)
    d, h = res["device"], res["host"]
    assert d[5] == 1 and h[5] == 0
    assert d[:4] == h[:4]
    assert np.array_equal(d[4], h[4])
    # the oracle's tree on the resampled poses gives the same leaf / bin counts
    t = orc.KDTree()
    for r in d[4][:, :3]:
        t.insert_pose(r, 1.0)
    assert (t.leaf_count(), t.node_count()) == (d[1], d[2])


def test_motion_update_full_size_properties(engine, orc):
    """100 k particles, every model: weights untouched; the drand48 state equals the oracle's after the same
    update (the oracle's serial loop takes ~15 ms here); zero noise parameters move every particle by exactly the
    odometry delta (to rounding); a slice of poses equals the oracle's to 1e-12."""
    import badger_amcl_amd as bpf
    n = 100000
    s = synth.converged_cloud(n, np.array([50.0, 50.0, 0.3]), seed=3)
    s[:, 3] = np.random.default_rng(4).uniform(0.1, 1.0, n)
    od = bpf.Odom(engine)
    pose, delta, absm = (3.0, -1.0, 0.7), (0.21, -0.08, 0.12), (0.25, 0.09, 0.15)
    for model in range(5):
        pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
        pf.setRngState(0x1234567 + model)
        pf.initWithSamples(s, leaf_count=1)
        od.setModel(model, 0.2, 0.15, 0.25, 0.1, 0.3)
        od.updateAction(pf, bpf.OdomData(pose, delta, absm))
        got = pf.getCurrentSet().samples
        want = s.copy()
        st = orc.odom_update_action(model, (0.2, 0.15, 0.25, 0.1, 0.3), pose, delta, absm, want, 0x1234567 + model)
        assert pf.getRngState() == st
        assert np.array_equal(got[:, 3], s[:, 3])
        assert np.abs(got[:, :3] - want[:, :3]).max() <= 1e-12
    # zero alphas: no noise at all, every Gaussian is exactly 0 * x
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
    pf.initWithSamples(s, leaf_count=1)
    od.setModel(bpf.pf.ODOM_MODEL_OMNI_CORRECTED, 0.0, 0.0, 0.0, 0.0, 0.0)
    od.updateAction(pf, bpf.OdomData(pose, delta, absm))
    got = pf.getCurrentSet().samples
    assert np.allclose(got[:, 2], s[:, 2] + delta[2], rtol=0, atol=1e-15)
    moved = np.hypot(got[:, 0] - s[:, 0], got[:, 1] - s[:, 1])
    assert np.allclose(moved, np.hypot(delta[0], delta[1]), rtol=0, atol=1e-13)


def test_edge_cases_of_the_scan(engine, world, orc):
    """All beams at / beyond range_max (skipped, planar_scanner.cpp:279-282): p = 1 for every particle; all NaN:
    likewise; one valid beam; range_count just below / above max_beams (step clamps to 1)."""
    import badger_amcl_amd as bpf
    sc, m = world["sc"], world["map"]
    n = 1000
    s = synth.converged_cloud(n, world["pose"], seed=9)
    angles = np.linspace(-2.0, 2.0, 1081)

    def weights(ranges, ang=angles):
        got = s.copy()
        total = sc.applyModelToSampleSet(bpf.PlanarData(ranges, ang, 30.0), got, 0)
        return got[:, 3], total

    # every reading at max range: nothing is scored, p = 1 (times the map factors of recalcWeight)
    w_max, t_max = weights(np.full(1081, 30.0))
    w_nan, t_nan = weights(np.full(1081, np.nan))
    assert np.array_equal(w_max, w_nan) and t_max == t_nan
    assert np.all(w_max <= s[:, 3]) and np.all(w_max >= s[:, 3] * 0.95 * 0.95 * (1 - 1e-15))
    # a single valid beam changes only that term
    r = np.full(1081, 30.0)
    r[540] = 4.0
    w_one, _ = weights(r)
    assert np.all(w_one >= w_max * (1 - 1e-15)) and np.all(w_one <= w_max * 2.0 * (1 + 1e-15))
    # fewer readings than max_beams: step = max((R - 1) / (max_beams - 1), 1) = 1, every reading used
    few = np.linspace(3.0, 9.0, 17)
    w_few, t_few = weights(few, np.linspace(-1.0, 1.0, 17))
    want = s.copy()
    omap = orc.OccupancyMap(np.asarray(m.cells), 0.05, m.origin, max_dist=2.0, lut=m.getDistancesLUT())
    p = orc.planar(orc.MODEL_LF, 1081, scanner_pose=synth.SCANNER_POSE, off_map_factor=synth.MAP_FACTORS[0],
                   non_free_space_factor=synth.MAP_FACTORS[1], non_free_space_radius=synth.MAP_FACTORS[2],
                   **synth.LF_DEFAULTS)
    want_total = orc.planar_apply(p, omap, want, few, np.linspace(-1.0, 1.0, 17), 30.0, 0)
    assert np.allclose(w_few, want[:, 3], rtol=1e-9, atol=0) and abs(t_few - want_total) <= 1e-9 * want_total


def test_edge_cases_of_the_set(engine, world):
    """One particle; every particle off the map (weights scale by off_map_factor only); all-zero weights
    (uniform reset, particle_filter.cpp:258-266); resampling a set of identical poses (one bin: the KLD limit is
    max_samples for a single leaf, so the set is refilled to max_samples)."""
    import badger_amcl_amd as bpf
    sc, data = world["sc"], world["data"]
    pf = bpf.ParticleFilter(engine, 10, 500, 0.0, 0.0, 85.0)
    one = np.array([[world["pose"][0], world["pose"][1], world["pose"][2], 1.0]])
    pf.initWithSamples(one)
    sc.updateSensor(pf, data)
    assert pf.getCurrentSet().samples[0, 3] == 1.0  # normalised: a single particle carries all the weight
    pf.updateResample()
    st = pf.getState()
    # one occupied bin: leaf count 1 -> resampleLimit returns max_samples (particle_filter.cpp:477-478)
    assert st.sample_count == 500 and st.leaf_count == 1 and st.bin_count == 1
    cur = pf.getCurrentSet().samples
    assert np.all(cur[:, :3] == one[0, :3]) and np.all(cur[:, 3] == 1.0 / 500)
    # all off the map
    off = synth.converged_cloud(300, np.array([-500.0, -500.0, 0.0]), seed=2)
    got = off.copy()
    total = sc.applyModelToSampleSet(data, got, 0)
    assert total > 0 and np.allclose(got[:, 3] / off[:, 3], got[0, 3] / off[0, 3], rtol=1e-15)
    # zero weights
    z = synth.converged_cloud(64, world["pose"], seed=3)
    z[:, 3] = 0.0
    pf.initWithSamples(z)
    sc.updateSensor(pf, data)
    assert np.all(pf.getCurrentSet().samples[:, 3] == 1.0 / 64)


def test_engines_release_their_memory(orc):
    """Create, use (every buffer family: motion, device tree, recovery chain, statistics) and destroy engines in a
    loop: device memory does not creep."""
    import ctypes as C
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    from badger_amcl_amd import _lib
    from scenario import Scenario
    sc_ = Scenario(orc, size=200, n=20000, beams=61, cloud="spread")

    def free_bytes():
        f, t = C.c_size_t(), C.c_size_t()
        assert _lib.load().bpf_device_memory_info(0, C.byref(f), C.byref(t)) == 0
        return f.value

    def cycle():
        e = bpf.Engine(0)
        e.set_option(hpf.OPT_KLD_DEVICE_MIN, 1)
        m, sc, pf, data = sc_.gpu_objects(e, 61, "lf", min_samples=100, seed=1, alpha=(0.001, 0.1))
        pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
        od = bpf.Odom(e)
        od.setModel(0, 0.1, 0.1, 0.1, 0.1)
        for ranges in (sc_.ranges, np.full(61, 1.0)):
            od.updateAction(pf, bpf.OdomData((0, 0, 0), (0.01, 0, 0.01)))
            sc.updateSensor(pf, bpf.PlanarData(ranges, sc_.angles, sc_.range_max))
            pf.updateResample()
            pf.computeClusterStats()
        e.close()

    cycle()
    free0 = free_bytes()
    for _ in range(4):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 32 << 20, (free0, free1)


def test_graded_and_equal_shares_give_the_same_weights(engine, world):
    """BPF_OPT_GRADED_SHARES only changes which wave scores which particle: per-particle weights are bit-identical,
    the total differs at most by its summation order (100 k x 1081, the size where the graded partition is on)."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    n = 100000
    s = synth.converged_cloud(n, world["pose"], seed=21)
    s[:, 3] = np.random.default_rng(5).uniform(0.5, 1.5, n) / n
    out = {}
    for mode in (1, 0):
        engine.set_option(hpf.OPT_GRADED_SHARES, mode)
        try:
            got = s.copy()
            total = world["sc"].applyModelToSampleSet(world["data"], got, 0)
            out[mode] = (got[:, 3].copy(), total)
        finally:
            engine.set_option(hpf.OPT_GRADED_SHARES, 1)
    assert np.array_equal(out[1][0], out[0][0])
    assert abs(out[1][1] - out[0][1]) <= 1e-12 * out[0][1]


def test_graded_shares_3d(engine, orc):
    """The same for the 3-D kernel at a size where its graded partition is on (12 288 particles x 65 536 points),
    plus an oracle spot check on a slice of the particles."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    pi, dr, mn, mx = synth.box_room_lut()
    pts = synth.grid_cloud(64, 1024)
    n = 12288
    rng = np.random.default_rng(8)
    s = np.zeros((n, 4))
    s[:, 0] = 0.3 + rng.normal(0, 0.1, n)
    s[:, 1] = 0.2 + rng.normal(0, 0.1, n)
    s[:, 2] = rng.normal(0, 0.05, n)
    s[:, 3] = rng.uniform(0.5, 1.5, n) / n
    om = bpf.OctoMap(engine, 0.05)
    om.setDistancesLUT(pi, dr, mn, mx, 0.3)
    sc = bpf.PointCloudScanner(engine)
    sc.init(65536, om)
    sc.setPointCloudModel(0.5, 0.05, 0.1)
    sc.setMapFactors(0.95, 0.95, 0.3)
    tf_xyz, tf_quat = (0.0, 0.0, 0.6), (0.0, 0.0, 0.0, 1.0)
    sc.setPointCloudScannerToFootprintTF(tf_xyz, tf_quat)
    data = bpf.PointCloudData(pts)
    out = {}
    for mode in (1, 0):
        engine.set_option(hpf.OPT_GRADED_SHARES, mode)
        try:
            got = s.copy()
            total = sc.applyModelToSampleSet(data, got)
            out[mode] = (got[:, 3].copy(), total)
        finally:
            engine.set_option(hpf.OPT_GRADED_SHARES, 1)
    assert np.array_equal(out[1][0], out[0][0])
    assert abs(out[1][1] - out[0][1]) <= 1e-12 * out[0][1]
    # oracle on 24 of the particles (65 536 points each)
    idx = np.linspace(0, n - 1, 24).astype(int)
    olut = orc.OctoMapLUT(mn, mx, 0.05, 0.3, pi, dr)
    op = orc.cloud(orc.CLOUD_MODEL, 65536, tf_xyz, tf_quat, z_hit=0.5, z_rand=0.05, sigma_hit=0.1)
    op.off_map_factor = 0.95
    want = np.ascontiguousarray(s[idx])
    orc.cloud_apply(op, olut, want, pts)
    rel = np.abs(out[1][0][idx] - want[:, 3]) / want[:, 3]
    assert (rel > 1e-9).sum() <= 1  # one point within rounding of a voxel face may resolve differently


def test_one_million_particles_on_one_gpu(engine, world, orc):
    """BASELINE config 4's particle count (1 M) on a single GPU: the CDF takes its two-launch form (more than 256
    tiles of 2 048 weights), the scoring kernel's ranges are ten times longer.  Size-independent checks: a particle's
    weight does not depend on how the set is cut (the same million scored in ten host-buffer calls of 100 k), the
    normalised weights sum to one, and the resample equals the oracle's on the weights the GPU produced (exact:
    count, leaf count, RNG state, poses)."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    n = 1000000
    samples = synth.converged_cloud(n, world["pose"], seed=77)
    samples[:, 3] = np.random.default_rng(5).uniform(0.5, 1.5, n) / n
    sc, data = world["sc"], world["data"]
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
    pf.srand48(9)
    pf.initWithSamples(samples)
    engine.set_option(hpf.OPT_CDF_SERIAL, 0)
    assert sc.updateSensor(pf, data)
    st0 = pf.getState()
    before = pf.getCurrentSet().samples
    assert abs(before[:, 3].sum() - 1.0) < 1e-11
    # the same particles in ten separate calls (un-normalised weights): equal up to the normalisation
    raw = np.empty(n)
    for k in range(10):
        part = np.ascontiguousarray(samples[k * 100000:(k + 1) * 100000])
        sc.applyModelToSampleSet(data, part, 0)
        raw[k * 100000:(k + 1) * 100000] = part[:, 3]
    assert np.allclose(raw / raw.sum(), before[:, 3], rtol=1e-12, atol=0)
    assert np.array_equal(raw / st0.total, before[:, 3]) or rel_max(raw / st0.total, before[:, 3]) < 1e-15
    pf.updateResample()
    st1 = pf.getState()
    after = pf.getCurrentSet().samples
    opf = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0, seed=9)
    opf.set_samples(before, leaf_count=st0.leaf_count)
    out = opf.update_resample()
    assert out.status == 0
    # the parallel CDF differs from the oracle's serial one by rounding: a draw within that distance of a CDF step may
    # pick the neighbouring particle (none expected in ~3 000 draws; allow one and require everything else exact)
    assert st1.sample_count == out.sample_count and st1.leaf_count == out.leaf_count
    assert pf.getRngState() == opf.pf.rng
    diff = np.flatnonzero(np.any(after[:, :3] != opf.samples[:out.sample_count, :3], axis=1))
    assert diff.size <= 1, diff[:5]
    assert np.all(after[:, 3] == 1.0 / out.sample_count)


def rel_max(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def test_tile_sorted_scoring_of_a_spread_cloud_gives_the_same_weights(engine, world):
    """A cloud the previous resample found spread is scored in map-tile order with each XCD taking a contiguous eighth
    of it (HOST_MODE 3 of k_score_field).  Nothing in a particle's own arithmetic depends on the order: the weights
    equal the index-order weights up to the rounding of the one division by the total (whose summation shape differs),
    and the resample that follows picks the same particles."""
    import badger_amcl_amd as bpf
    n = 100000
    samples = synth.spread_cloud(n, world["size"], seed=77, margin=0.5)
    sc, data = world["sc"], world["data"]
    out = {}
    for mode in (0, 1):
        engine.set_option(bpf.pf.OPT_TILE_SORT, mode)
        try:
            pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)  # a new filter: no resample has said "spread" yet
            pf.srand48(5)
            pf.initWithSamples(samples)
            sc.updateSensor(pf, data)
            assert engine.score_last_form() == 0   # nothing says "spread" yet
            pf.updateResample()                     # runs to the end: no KLD stop for a spread cloud
            assert pf.getState().sample_count == n
            pf.srand48(5)
            pf.initWithSamples(samples)
            sc.updateSensor(pf, data)
            assert engine.score_last_form() == (3 if mode else 0)
            w = pf.getCurrentSet().samples[:, 3].copy()
            st = pf.getState()
            pf.updateResample()
            out[mode] = (w, st.total, pf.getCurrentSet().samples.copy(), pf.getState())
        finally:
            engine.set_option(bpf.pf.OPT_TILE_SORT, 1)
    w0, t0, set0, st0 = out[0]
    w1, t1, set1, st1 = out[1]
    assert abs(t1 - t0) <= 1e-13 * abs(t0)
    assert np.max(np.abs(w1 - w0) / w0) <= 1e-14
    assert abs(w1.sum() - 1.0) < 1e-12
    assert (st1.sample_count, st1.leaf_count) == (st0.sample_count, st0.leaf_count)
    assert np.array_equal(set1[:, :3], set0[:, :3])
