"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs.  Tolerances: integer / index / pose-copy results bit-exact; weights
1e-9 relative (north_star allows 1e-6; observed ~1e-13), stated per test."""
import numpy as np
import pytest

from scenario import Scenario, rel_err

pytestmark = pytest.mark.gpu

W_TOL = 1e-9


@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


def _knife_edge_budget(n_evals):
    # an end point within ~1e-12 cells of a cell border may land in the neighbour cell
    # (DESIGN.md "Knife edges"): expected count n_evals * 2e-12, allow 1 + that.
    return 1 + int(n_evals * 2e-12)


@pytest.mark.parametrize("model", ["lf", "gompertz", "prob", "beam"])
@pytest.mark.parametrize("cloud", ["converged", "mixture"])
def test_apply_model_to_sample_set_matches_oracle(engine, orc, model, cloud):
    """Seam A with host buffers (PlanarScanner::applyModelToSampleSet)."""
    sc_ = Scenario(orc, size=200, n=257, beams=61, cloud=cloud)
    max_beams = 61 if model != "beam" else 31
    m, sc, pf, data = sc_.gpu_objects(engine, max_beams, model)
    got = sc_.samples.copy()
    total = sc.applyModelToSampleSet(data, got, 0)
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(max_beams, model), want)
    assert np.array_equal(got[:, :3], want[:, :3])
    bad = rel_err(got[:, 3], want[:, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(257 * 61), (model, cloud, np.flatnonzero(bad)[:10])
    assert abs(total - want_total) <= 1e-9 * abs(want_total)


@pytest.mark.parametrize("cloud,n", [("converged", 5000), ("mixture", 1000), ("converged", 63)])
def test_beam_model_longer_scan_and_more_particles(engine, orc, cloud, n):
    """The beam model on a 181-beam scan decimated to 91 rays, sets large enough to give every wave several trips."""
    sc_ = Scenario(orc, size=200, n=n, beams=181, cloud=cloud, frac_nan=0.0)  # the beam model does not skip NaN ranges
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(91, "beam"), want)
    assert want_total == want_total
    m, sc, pf, data = sc_.gpu_objects(engine, 91, "beam")
    w = sc_.samples.copy()
    total = sc.applyModelToSampleSet(data, w, 0)
    bad = rel_err(w[:, 3], want[:, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(n * 91), np.flatnonzero(bad)[:10]
    assert abs(total - want_total) <= 1e-9 * abs(want_total)


def test_lf_decimation_and_single_particle(engine, orc):
    """max_beams < range_count (step > 1), N = 1, zero scanner offset."""
    sc_ = Scenario(orc, size=200, n=1, beams=181, cloud="converged", scanner_pose=(0.0, 0.0, 0.0))
    m, sc, pf, data = sc_.gpu_objects(engine, 30, "lf")
    got = sc_.samples.copy()
    total = sc.applyModelToSampleSet(data, got, 0)
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(30, "lf"), want)
    assert rel_err(got[:, 3], want[:, 3]).max() <= W_TOL
    assert abs(total - want_total) <= 1e-9 * abs(want_total)


def test_max_beams_below_two_is_a_no_op(engine, orc):
    """planar_scanner.cpp:128-129,144-145: returns false / 0.0 and leaves the weights alone."""
    sc_ = Scenario(orc, size=200, n=33, beams=61)
    m, sc, pf, data = sc_.gpu_objects(engine, 1, "lf")
    got = sc_.samples.copy()
    assert sc.applyModelToSampleSet(data, got, 0) == 0.0
    assert np.array_equal(got, sc_.samples)
    assert sc.updateSensor(pf, data) is False


@pytest.mark.parametrize("model", ["lf", "gompertz"])
def test_update_sensor_normalises_like_reference(engine, orc, model):
    """ParticleFilter::updateSensor on the resident set: normalised weights, w_slow / w_fast."""
    sc_ = Scenario(orc, size=400, n=5000, beams=181, cloud="mixture")
    m, sc, pf, data = sc_.gpu_objects(engine, 181, model, alpha=(0.001, 0.1))
    assert sc.updateSensor(pf, data) is True
    assert sc.updateSensor(pf, data) is True  # second update exercises the running averages
    cur = pf.getCurrentSet()
    st = pf.getState()
    opf = orc.ParticleFilter(100, 5000, 0.001, 0.1)
    opf.set_samples(sc_.samples, leaf_count=0)
    p = sc_.oracle_planar(181, model)
    for _ in range(2):
        total = opf.update_sensor(lambda s, conv: sc_.oracle_apply(p, s, conv))
    bad = rel_err(cur.samples[:, 3], opf.samples[:5000, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(2 * 5000 * 181)
    assert abs(cur.samples[:, 3].sum() - 1.0) < 1e-12
    assert abs(st.total - total) <= 1e-9 * total
    assert abs(st.w_slow - opf.pf.w_slow) <= 1e-9 * opf.pf.w_slow
    assert abs(st.w_fast - opf.pf.w_fast) <= 1e-9 * opf.pf.w_fast


def test_zero_total_resets_to_uniform(engine, orc):
    """particle_filter.cpp:258-266"""
    sc_ = Scenario(orc, size=200, n=100, beams=61)
    sc_.samples[:, 3] = 0.0
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf")
    sc.updateSensor(pf, data)
    cur = pf.getCurrentSet()
    assert np.all(cur.samples[:, 3] == 1.0 / 100)


@pytest.mark.timeout(120)
def test_infinite_weights_do_not_stall_the_normalise_launch(engine, orc):
    """inf / inf: the tile sums of the normalise launch are NaN.  Its blocks hand those sums to each other through
    slots whose empty state is a NaN bit pattern as well (kernels_fused.hpp), so this is the input that must not leave
    a block waiting: the call returns, the weights are NaN (as in the reference, which divides the same way), and the
    next update of a sane set is exact again (the slots of the other launch parity were put back)."""
    sc_ = Scenario(orc, size=200, n=6000, beams=61)  # three look-back tiles
    good = sc_.samples.copy()
    sc_.samples[:, 3] = np.inf
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf")
    sc.updateSensor(pf, data)
    assert np.all(np.isnan(pf.getCurrentSet().samples[:, 3]))
    want = good.copy()
    total = sc_.oracle_apply(sc_.oracle_planar(61, "lf"), want)
    for _ in range(2):  # both launch parities
        pf.initWithSamples(good)
        sc.updateSensor(pf, data)
        got = pf.getCurrentSet().samples
        assert rel_err(got[:, 3], want[:, 3] / total).max() <= W_TOL


def _oracle_resample_from(orc, samples, w_slow, w_fast, leaf_count, seed, model, min_s, max_s, pop=None):
    opf = orc.ParticleFilter(min_s, max_s, 0.0, 0.0, 85.0, seed=seed)
    opf.set_samples(samples, leaf_count=leaf_count)
    opf.pf.w_slow, opf.pf.w_fast = w_slow, w_fast
    opf.set_resample_model(model)
    if pop:
        opf.set_population_size_parameters(*pop)
    out = opf.update_resample()
    return opf, out


@pytest.mark.parametrize("cloud,n", [("converged", 5000), ("spread", 3000)])
@pytest.mark.parametrize("resampler", [0, 1])
@pytest.mark.parametrize("serial_cdf", [0, 1])
def test_update_resample_matches_oracle(engine, orc, cloud, n, resampler, serial_cdf):
    """Seam B.  The oracle resamples the weights the GPU produced, so the only arithmetic
    difference left is the CDF summation order (none with the serial option): sample count,
    source indices (via bit-equal poses), weights 1/M, leaf count and RNG state must agree exactly."""
    import badger_amcl_amd.pf as hpf
    sc_ = Scenario(orc, size=400, n=n, beams=91, cloud=cloud)
    engine.set_option(hpf.OPT_CDF_SERIAL, serial_cdf)
    try:
        m, sc, pf, data = sc_.gpu_objects(engine, 91, "lf", min_samples=100, seed=7)
        pf.setResampleModel(resampler)
        sc.updateSensor(pf, data)
        before = pf.getCurrentSet()
        st0 = pf.getState()
        pf.updateResample()
        after = pf.getCurrentSet()
        st1 = pf.getState()
    finally:
        engine.set_option(hpf.OPT_CDF_SERIAL, 0)
    opf, out = _oracle_resample_from(orc, before.samples, st0.w_slow, st0.w_fast, st0.leaf_count, 7, resampler,
                                     100, n)
    assert out.status == 0
    assert st1.sample_count == out.sample_count
    assert st1.leaf_count == out.leaf_count
    assert st1.bin_count == out.node_count
    M = out.sample_count
    assert np.array_equal(after.samples[:, :3], opf.samples[:M, :3])
    assert np.all(after.samples[:, 3] == 1.0 / M)
    assert pf.getRngState() == opf.pf.rng
    assert st1.converged == out.converged
    assert abs(st1.percent_converged - out.percent_converged) < 1e-4
    assert st1.last_status == 0


@pytest.mark.parametrize("cloud,n,pop", [("spread", 3000, None), ("converged", 3000, None), ("mixture", 20000, None),
                                         ("spread", 30000, (0.0025, 0.9975)), ("mixture", 9000, (0.05, 0.99))])
@pytest.mark.parametrize("persistent", [0, 1])
def test_kld_stop_rule_on_device_matches_oracle(engine, orc, cloud, n, pop, persistent):
    """The level-synchronous device build of the histogram tree (long draw streams) against the oracle's serial
    insertion: stop count, leaf count, bin count, poses and RNG state exact.  BPF_OPT_KLD_DEVICE_MIN = 1 sends
    even these small sets through it; serial CDF so that no draw sits on a summation-order knife edge.  Both forms:
    one launch with grid barriers between the levels (k_kld_tree_persistent), and one launch pair per level."""
    import badger_amcl_amd.pf as hpf
    sc_ = Scenario(orc, size=400, n=n, beams=61, cloud=cloud)
    engine.set_option(hpf.OPT_CDF_SERIAL, 1)
    engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 1)
    engine.set_option(hpf.OPT_KLD_PERSISTENT, persistent)
    try:
        m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", min_samples=100, seed=5)
        if pop:
            pf.setPopulationSizeParameters(*pop)
        results = []
        for cycle in range(2):  # the second cycle starts from the window hint the first one left
            sc.updateSensor(pf, data)
            before = pf.getCurrentSet()
            st0 = pf.getState()
            rng0 = pf.getRngState()
            pf.updateResample()
            after = pf.getCurrentSet()
            st1 = pf.getState()
            opf = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0)
            opf.pf.rng = rng0
            opf.set_samples(before.samples, leaf_count=st0.leaf_count)
            opf.pf.w_slow, opf.pf.w_fast = st0.w_slow, st0.w_fast
            if pop:
                opf.set_population_size_parameters(*pop)
            out = opf.update_resample()
            assert out.status == 0
            if n <= 4096 or out.sample_count > 4096:  # otherwise the host's first window already found the stop
                assert st1.kld_on_device == 1
            assert st1.sample_count == out.sample_count, cycle
            assert st1.leaf_count == out.leaf_count and st1.bin_count == out.node_count
            M = out.sample_count
            assert np.array_equal(after.samples[:, :3], opf.samples[:M, :3])
            assert pf.getRngState() == opf.pf.rng
            results.append(M)
            # continue from the whole cloud again so that the second cycle is a long stream too
            pf.initWithSamples(sc_.samples)
        # cluster statistics after a device-side stop: the host tree is rebuilt on demand
        sc.updateSensor(pf, data)
        pf.updateResample()
        cur = pf.getCurrentSet().samples
        t = orc.KDTree()
        for k in range(cur.shape[0]):
            t.insert_pose(cur[k, :3], cur[k, 3])
        want = t.cluster_stats(cur, n)
        cnt, mean, cov = pf.computeClusterStats()
        assert cnt == want["n"] and np.allclose(mean, want["set_mean"], rtol=1e-12, atol=1e-12)
    finally:
        engine.set_option(hpf.OPT_CDF_SERIAL, 0)
        engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 8192)
        engine.set_option(hpf.OPT_KLD_PERSISTENT, 0)


def test_stop_beyond_the_first_window_takes_sized_follow_up_windows(engine, orc):
    """First a resample with a loose KLD bound (pop_err 0.05: a set of ~130, so the next resample starts with the
    minimum window of 1024 draws), then the default bound on the full cloud again, whose stop lies near 1900 draws,
    with the device tree switched off: the host replay goes on through a follow-up window sized from the bound for
    the leaves seen so far.  Exact against the oracle."""
    import badger_amcl_amd.pf as hpf
    sc_ = Scenario(orc, size=400, n=20000, beams=61, cloud="converged")
    engine.set_option(hpf.OPT_CDF_SERIAL, 1)
    engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 0)  # 0 = never
    try:
        m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", min_samples=100, seed=5)
        pf.setPopulationSizeParameters(0.05, 0.99)
        sc.updateSensor(pf, data)
        pf.updateResample()
        assert pf.getState().sample_count < 800
        pf.setPopulationSizeParameters(0.01, 3.0)
        pf.initWithSamples(sc_.samples)
        sc.updateSensor(pf, data)
        before, st0, rng0 = pf.getCurrentSet(), pf.getState(), pf.getRngState()
        pf.updateResample()
        after, st1 = pf.getCurrentSet(), pf.getState()
        opf = orc.ParticleFilter(100, 20000, 0.0, 0.0, 85.0)
        opf.pf.rng = rng0
        opf.set_samples(before.samples, leaf_count=st0.leaf_count)
        out = opf.update_resample()
        assert out.status == 0 and out.sample_count > 1024
        assert st1.resample_windows == 2 and st1.kld_on_device == 0
        assert st1.sample_count == out.sample_count and st1.leaf_count == out.leaf_count
        assert np.array_equal(after.samples[:, :3], opf.samples[:out.sample_count, :3])
        assert pf.getRngState() == opf.pf.rng
    finally:
        engine.set_option(hpf.OPT_CDF_SERIAL, 0)
        engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 8192)


def test_resample_kld_parameters_and_second_cycle(engine, orc):
    """Launch-file KLD parameters (kld_err .0025 passed as pop_err, kld_z .9975 as pop_z) and two
    full update+resample cycles, the second starting from the resampled set."""
    sc_ = Scenario(orc, size=400, n=4000, beams=91, cloud="converged")
    m, sc, pf, data = sc_.gpu_objects(engine, 91, "gompertz", min_samples=500, seed=11)
    pf.setPopulationSizeParameters(0.0025, 0.9975)
    opf = orc.ParticleFilter(500, 4000, 0.0, 0.0, 85.0, seed=11)
    opf.set_population_size_parameters(0.0025, 0.9975)
    opf.set_samples(sc_.samples)
    p = sc_.oracle_planar(91, "gompertz")
    for cycle in range(2):
        sc.updateSensor(pf, data)
        pf.updateResample()
        opf.update_sensor(lambda s, conv: sc_.oracle_apply(p, s, conv))
        out = opf.update_resample()
        st = pf.getState()
        cur = pf.getCurrentSet()
        assert st.sample_count == out.sample_count, cycle
        assert st.leaf_count == out.leaf_count
        assert np.array_equal(cur.samples[:, :3], opf.samples[:out.sample_count, :3])
        assert pf.getRngState() == opf.pf.rng


@pytest.mark.parametrize("resampler", [0, 1])
@pytest.mark.parametrize("n,device_kld", [(2500, False), (6000, True)])
def test_recovery_random_poses_match_oracle(engine, orc, resampler, n, device_kld):
    """w_diff > 0 (augmented-MCL recovery, particle_filter.cpp:295-324,383-388) with random_pose_fn_ =
    Node::randomFreeSpacePose: which draws become random poses, the poses themselves (cell of
    Node2D::updateFreeSpaceIndices + heading, bit for bit), the interleaved consumption of the drand48 stream, the
    grown systematic count and the reset of w_slow / w_fast, against the oracle over three cycles."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    sc_ = Scenario(orc, size=200, n=n, beams=61, cloud="mixture")
    engine.set_option(hpf.OPT_CDF_SERIAL, 1)
    engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 1 if device_kld else 8192)
    try:
        m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", min_samples=100, seed=31, alpha=(0.001, 0.1))
        pf.setResampleModel(resampler)
        pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
        opf = orc.ParticleFilter(100, n, 0.001, 0.1, 85.0, seed=31)
        opf.set_resample_model(resampler)
        opf.set_samples(sc_.samples)
        n_free = opf.set_random_pose_source(sc_.omap, sc_.map_factors[2])
        assert n_free > 1000
        p = sc_.oracle_planar(61, "lf")
        # scan 1 fits the map, scans 2 and 3 are progressively worse: w_fast falls below w_slow
        scans = [sc_.ranges, np.clip(sc_.ranges * 0.6, 0.05, 29.0), np.full(61, 1.0)]
        w_diffs = []
        for cycle, ranges in enumerate(scans):
            sc.updateSensor(pf, bpf.PlanarData(ranges, sc_.angles, sc_.range_max))
            # the oracle scores the device's current set, so that only the resampling is under test
            cur = pf.getCurrentSet()
            st0 = pf.getState()
            opf.set_samples(cur.samples, leaf_count=st0.leaf_count)
            opf.pf.w_slow, opf.pf.w_fast = st0.w_slow, st0.w_fast
            opf.pf.rng = pf.getRngState()
            pf.updateResample()
            out = opf.update_resample()
            st1 = pf.getState()
            w_diffs.append(out.w_diff)
            assert out.status == 0, cycle
            assert abs(st1.w_diff - out.w_diff) <= 1e-12
            assert st1.sample_count == out.sample_count, (cycle, out.w_diff)
            assert st1.leaf_count == out.leaf_count and st1.bin_count == out.node_count
            M = out.sample_count
            after = pf.getCurrentSet().samples
            assert np.array_equal(after[:, :3], opf.samples[:M, :3])
            assert np.all(after[:, 3] == 1.0 / M)
            assert pf.getRngState() == opf.pf.rng
            if out.w_diff > 0:
                assert st1.w_slow == 0.0 and st1.w_fast == 0.0  # particle_filter.cpp:453-455
                n_random = int((opf.last_idx < 0).sum())
                assert n_random > 0
        assert max(w_diffs) > 0.01  # the scenario really exercised the recovery branch
    finally:
        engine.set_option(hpf.OPT_CDF_SERIAL, 0)
        engine.set_option(hpf.OPT_KLD_DEVICE_MIN, 8192)


def test_w_diff_positive_without_a_generator_is_refused_loudly(engine, orc):
    """Without a random pose generator the recovery branch cannot run: the engine reports it instead of
    silently doing something else."""
    import badger_amcl_amd as bpf
    sc_ = Scenario(orc, size=200, n=500, beams=61)
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", alpha=(0.001, 0.5))
    sc.updateSensor(pf, data)
    # a much worse second scan drops w_fast below w_slow
    bad = bpf.PlanarData(np.full(61, 1.0), sc_.angles, sc_.range_max)
    sc.updateSensor(pf, bad)
    st = pf.getState()
    assert st.w_fast < st.w_slow  # the precondition of the branch under test, not a filter on it
    with pytest.raises(bpf.BpfError) as ei:
        pf.updateResample()
    assert ei.value.code == 4


def test_prob_beamskip_matches_oracle(engine, orc):
    """calcLikelihoodFieldModelProb with beam skipping active (converged set)."""
    sc_ = Scenario(orc, size=200, n=300, beams=60, cloud="converged", frac_max=0.0, frac_nan=0.0)
    kw = dict(do_beamskip=1, beam_skip_distance=0.5, beam_skip_threshold=0.3, beam_skip_error_threshold=0.9)
    m, sc, pf, data = sc_.gpu_objects(engine, 60, "prob", model_kw=kw)
    got = sc_.samples.copy()
    total = sc.applyModelToSampleSet(data, got, 1)
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(60, "prob", kw), want, 1)
    assert rel_err(got[:, 3], want[:, 3]).max() <= W_TOL
    assert abs(total - want_total) <= 1e-9 * abs(want_total)
    # with invalid beams present and the error threshold tripped every weight becomes zero
    sc2 = Scenario(orc, size=200, n=100, beams=60, cloud="spread", frac_max=0.2, frac_nan=0.0)
    kw2 = dict(kw, beam_skip_threshold=0.99, beam_skip_error_threshold=0.1)
    m, sc, pf, data = sc2.gpu_objects(engine, 60, "prob", model_kw=kw2)
    got = sc2.samples.copy()
    total = sc.applyModelToSampleSet(data, got, 1)
    want = sc2.samples.copy()
    want_total = sc2.oracle_apply(sc2.oracle_planar(60, "prob", kw2), want, 1)
    assert want_total == 0.0 and total == 0.0
    assert np.array_equal(got[:, 3], want[:, 3])


def test_calc_range_matches_the_cell_by_cell_walk(engine, orc):
    """OccupancyMap::calcRange through the C-ABI (jumping walk over the chessboard-distance grid) against the
    oracle's cell-by-cell Bresenham on 20 000 random rays: exact, including rays that run the whole 30 m."""
    import math
    sc_ = Scenario(orc, size=400, n=4, beams=11)
    m, sc, pf, data = sc_.gpu_objects(engine, 11, "beam")
    rng = np.random.default_rng(12)
    n = 20000
    ext = 400 * 0.05
    ox, oy = rng.uniform(-0.5, ext + 0.5, n), rng.uniform(-0.5, ext + 0.5, n)
    oa = rng.uniform(-math.pi, math.pi, n)
    mr = rng.choice([0.3, 2.0, 8.0, 30.0], n)
    got = m.calcRange(ox, oy, oa, mr)
    want = np.array([sc_.omap.calc_range(float(a), float(b), float(c), float(d)) for a, b, c, d in zip(ox, oy, oa, mr)])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert (want < mr).mean() > 0.3 and (want == mr).mean() > 0.05


def test_beam_model_step_zero_is_refused(engine, orc):
    import badger_amcl_amd as bpf
    sc_ = Scenario(orc, size=200, n=10, beams=20)
    m, sc, pf, data = sc_.gpu_objects(engine, 60, "beam")
    with pytest.raises(bpf.BpfError) as ei:
        sc.applyModelToSampleSet(data, sc_.samples.copy(), 0)
    assert ei.value.code == 7


def test_device_distance_lut_is_exact_edt(engine, orc):
    """bpf_map2d_build_distances_lut: exact capped EDT on the reference's value lattice; the
    reference's brushfire may only be >= it, and equal almost everywhere."""
    import badger_amcl_amd as bpf
    sc_ = Scenario(orc, size=200, n=10, beams=20, max_dist=0.5)
    m = bpf.OccupancyMap(engine, sc_.res)
    m.setCells(sc_.cells)
    m.setOrigin(sc_.origin)
    m.updateDistancesLUTExact(0.5)
    got = m.getDistancesLUT()
    occ = np.argwhere(sc_.cells == 1)
    ys, xs = np.mgrid[0:200, 0:200]
    best = np.full((200, 200), 10**9, dtype=np.int64)
    for oy, ox in occ:
        y0, y1 = max(oy - 11, 0), min(oy + 12, 200)
        x0, x1 = max(ox - 11, 0), min(ox + 12, 200)
        d2 = (ys[y0:y1, x0:x1] - oy) ** 2 + (xs[y0:y1, x0:x1] - ox) ** 2
        best[y0:y1, x0:x1] = np.minimum(best[y0:y1, x0:x1], d2)
    radius = int(np.floor(0.5 / 0.05))
    want = np.where(best <= radius * radius, (np.sqrt(best.astype(np.float64)) * 0.05), 0.5).astype(np.float32)
    assert np.array_equal(got, want)
    assert np.all(sc_.lut >= got)
    assert np.mean(sc_.lut == got) > 0.99


def test_full_size_properties(engine, orc):
    """BASELINE config 2 size (100k x 1081, 2000^2 map): size-independent properties."""
    import badger_amcl_amd as bpf
    from badger_amcl_amd import synth
    size, n, beams = 2000, 100000, 1081
    cells, origin = synth.make_map(size)
    pose = synth.true_pose(size)
    ranges, angles = synth.cast_scan(cells, origin, 0.05, pose, beams, seed=5)
    samples = synth.converged_cloud(n, pose)
    m = bpf.OccupancyMap(engine, 0.05)
    m.setCells(cells)
    m.setOrigin(origin)
    m.updateDistancesLUT(2.0)
    sc = bpf.PlanarScanner(engine)
    sc.init(beams, m)
    sc.setModelLikelihoodField(0.95, 0.05, 0.2, 2.0)
    sc.setMapFactors(*synth.MAP_FACTORS)
    sc.setPlanarScannerPose(synth.SCANNER_POSE)
    data = bpf.PlanarData(ranges, angles, 30.0)
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)

    def run(scale, seed):
        s = samples.copy()
        s[:, 3] *= scale
        pf.initWithSamples(s)
        pf.srand48(seed)
        sc.updateSensor(pf, data)
        w = pf.getCurrentSet().samples[:, 3].copy()
        st0 = pf.getState()
        pf.updateResample()
        return w, st0, pf.getCurrentSet(), pf.getState()

    w1, s1, set1, st1 = run(1.0, 42)
    w2, s2, set2, st2 = run(2.0, 42)
    # normalised weights sum to one and do not depend on the scale of the prior weights
    assert abs(w1.sum() - 1.0) < 1e-12
    assert np.allclose(w1, w2, rtol=1e-13, atol=0)
    assert abs(s2.total - 2.0 * s1.total) <= 1e-12 * s2.total
    # p = 1 + sum pz^3 with pz <= 1: every weight ratio lies in [1, 1 + beams]
    ratio = w1 * s1.total / (1.0 / n)
    assert ratio.min() >= 0.95 * 0.95 and ratio.max() <= 1.0 + beams
    # same seed, same weights -> identical resample; outputs are copies of input poses
    assert st1.sample_count == st2.sample_count and st1.leaf_count == st2.leaf_count
    assert np.array_equal(set1.samples, set2.samples)
    M = st1.sample_count
    assert 100 <= M <= n and np.all(set1.samples[:, 3] == 1.0 / M)
    src = {tuple(r) for r in samples[:, :3]}
    assert all(tuple(r) in src for r in set1.samples[:200, :3])
    # the oracle's kd-tree on the resampled poses reproduces the engine's leaf / bin counts
    t = orc.KDTree()
    for r in set1.samples[:, :3]:
        t.insert_pose(r, 1.0)
    assert (t.leaf_count(), t.node_count()) == (st1.leaf_count, st1.bin_count)
    # and the KLD stop rule holds exactly at M and at no earlier draw
    opf = orc.ParticleFilter(100, n)
    assert M > opf.resample_limit(st1.leaf_count) or M == n


def test_lds_window_scoring_path_matches_oracle(engine, orc):
    """BPF_OPT_WINDOW_PATH (opt-in; a measured negative result on wide clouds): the LDS-window
    kernels must give the same weights as the oracle when the device-side switch takes them."""
    import badger_amcl_amd.pf as hpf
    from badger_amcl_amd import synth
    # 1081 beams: a chunk of 64 consecutive beams is a 16-degree arc; range_max 8 m keeps the arcs short
    sc_ = Scenario(orc, size=400, n=16384, beams=1081, cloud="converged", frac_nan=0.0, range_max=8.0)
    # a tight cloud, so that every chunk's 3-sigma footprint fits a window and the device-side switch takes them
    sc_.samples = synth.converged_cloud(16384, sc_.pose, seed=77, sigma=(0.05, 0.05, 0.01))
    sc_.samples[:, 3] *= np.random.default_rng(78).uniform(0.5, 1.5, 16384)
    engine.set_option(hpf.OPT_WINDOW_PATH, 1)
    try:
        m, sc, pf, data = sc_.gpu_objects(engine, 1081, "lf")
        sc.updateSensor(pf, data)
        got = pf.getCurrentSet().samples
        plan = engine.window_plan()
    finally:
        engine.set_option(hpf.OPT_WINDOW_PATH, 0)
    opf = orc.ParticleFilter(100, 16384, 0.0, 0.0, 85.0)
    opf.set_samples(sc_.samples, leaf_count=1)
    p = sc_.oracle_planar(1081, "lf")
    opf.update_sensor(lambda s, conv: sc_.oracle_apply(p, s, conv))
    bad = rel_err(got[:, 3], opf.samples[:16384, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(16384 * 1081)
    assert plan["chunks_total"] == 17 and plan["used_window"], plan  # the device-side switch took the windows


@pytest.mark.parametrize("resampler", [0, 1])
@pytest.mark.parametrize("cloud,n,min_s,pop", [("converged", 20000, 100, None), ("converged", 3000, 100, None),
                                                ("mixture", 2500, 500, (0.05, 0.99)), ("spread", 1500, 10, None),
                                                ("converged", 100000, 100, None)])
def test_one_block_resample_equals_the_general_path_and_the_oracle(engine, orc, cloud, n, min_s, pop, resampler):
    """BPF_OPT_FUSED_RESAMPLE (default on): normalise + CDF in one launch and, when the candidate stream fits 4096
    draws, the whole resample in one single-block launch with the histogram tree grown in LDS.  Three cycles (the
    second and third start from the window hint the previous one left); against the separate launches with the
    host's ordered replay (must be bit-identical: set, counts, RNG state, convergence) and against the oracle."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    size = 2000 if n == 100000 else 400
    beams = 181 if n == 100000 else 61
    sc_ = Scenario(orc, size=size, n=n, beams=beams, cloud=cloud)
    runs = {}
    for mode in (1, 0):
        engine.set_option(hpf.OPT_FUSED_RESAMPLE, mode)
        try:
            m, sc, pf, data = sc_.gpu_objects(engine, beams, "lf", min_samples=min_s, seed=19)
            pf.setResampleModel(resampler)
            if pop:
                pf.setPopulationSizeParameters(*pop)
            log = []
            for cycle in range(3):
                sc.updateSensor(pf, data)
                before, st0, rng0 = pf.getCurrentSet().samples, pf.getState(), pf.getRngState()
                pf.updateResample()
                st1 = pf.getState()
                log.append((before, st0, rng0, pf.getCurrentSet().samples, st1, pf.getRngState()))
            runs[mode] = log
        finally:
            engine.set_option(hpf.OPT_FUSED_RESAMPLE, 1)
    used_block = 0
    for cycle in range(3):
        b1, s01, r01, a1, s1, r1 = runs[1][cycle]
        b0, s00, r00, a0, s0, r0 = runs[0][cycle]
        assert np.array_equal(b1, b0) and s01.total == s00.total and s01.w_slow == s00.w_slow, cycle
        assert np.array_equal(a1, a0) and r1 == r0, cycle
        assert (s1.sample_count, s1.leaf_count, s1.bin_count, s1.converged) == \
               (s0.sample_count, s0.leaf_count, s0.bin_count, s0.converged), cycle
        assert s1.percent_converged == s0.percent_converged
        assert s0.kld_on_device in (0, 1)
        used_block += s1.kld_on_device == 2
        opf = orc.ParticleFilter(min_s, n, 0.0, 0.0, 85.0)
        opf.pf.rng = r01
        opf.set_resample_model(resampler)
        opf.pf.w_slow, opf.pf.w_fast = s01.w_slow, s01.w_fast  # 0 / 0 would make w_diff NaN
        if pop:
            opf.set_population_size_parameters(*pop)
        opf.set_samples(b1, leaf_count=s01.leaf_count)
        out = opf.update_resample()
        assert out.status == 0
        M = out.sample_count
        assert (s1.sample_count, s1.leaf_count, s1.bin_count) == (M, out.leaf_count, out.node_count), cycle
        assert np.array_equal(a1[:, :3], opf.samples[:M, :3]) and np.all(a1[:, 3] == 1.0 / M)
        assert r1 == opf.pf.rng and s1.converged == out.converged
    # the one-block kernel really ran (a first cycle beyond its window may not use it; the systematic resampler's
    # count follows the previous set's leaf count and can stay above 4096 for the first cycles of a large set)
    assert used_block >= (2 if resampler == 0 else (1 if n <= 4096 else 0))


def test_one_block_resample_declines_keys_beyond_its_packing(engine, orc):
    """Poses 5e6 m from the origin: the histogram key floor(x / 0.5) does not fit 24 bits, the one-block kernel
    reports it and the general path (host replay, 32-bit keys) takes over with the same result as the oracle."""
    sc_ = Scenario(orc, size=200, n=2000, beams=61, cloud="converged")
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", min_samples=100, seed=23)
    far = sc_.samples.copy()
    far[:, 0] += 5.0e6
    far[:, 3] /= far[:, 3].sum()
    pf.initWithSamples(far)
    rng0 = pf.getRngState()
    pf.updateResample()
    st = pf.getState()
    opf = orc.ParticleFilter(100, 2000, 0.0, 0.0, 85.0)
    opf.pf.rng = rng0
    opf.set_samples(far)
    out = opf.update_resample()
    assert st.kld_on_device == 0
    assert (st.sample_count, st.leaf_count) == (out.sample_count, out.leaf_count)
    assert np.array_equal(pf.getCurrentSet().samples[:, :3], opf.samples[:out.sample_count, :3])


@pytest.mark.parametrize("model", ["lf", "gompertz", "prob", "beam"])
@pytest.mark.parametrize("kind", ["all_max", "all_nan", "one_valid"])
def test_degenerate_scans_match_oracle(engine, orc, model, kind):
    """Scans with nothing (or almost nothing) to score: every range at range_max, every range NaN, a single valid
    return -- the skip rules of planar_scanner.cpp:265-282 and the models' own treatment of max-range readings."""
    if model == "beam" and kind != "all_max":
        pytest.skip("the beam model does not skip NaN ranges (planar_scanner.cpp:168-234): NaN in, NaN out")
    sc_ = Scenario(orc, size=200, n=129, beams=61, cloud="mixture")
    r = sc_.ranges
    keep = float(r[17]) if np.isfinite(r[17]) and r[17] < sc_.range_max else 3.0
    r[:] = sc_.range_max if kind == "all_max" else np.nan
    if kind == "one_valid":
        r[17] = keep
    max_beams = 61 if model != "beam" else 31
    m, sc, pf, data = sc_.gpu_objects(engine, max_beams, model)
    got = sc_.samples.copy()
    total = sc.applyModelToSampleSet(data, got, 0)
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(max_beams, model), want)
    assert np.array_equal(got[:, :3], want[:, :3])
    assert rel_err(got[:, 3], want[:, 3]).max() <= W_TOL
    assert abs(total - want_total) <= 1e-9 * abs(want_total)


@pytest.mark.parametrize("resampler", [0, 1])
@pytest.mark.parametrize("kind", ["one_heavy", "last_heavy", "two_equal", "tiny_set"])
def test_resample_of_degenerate_weight_vectors_matches_oracle(engine, orc, resampler, kind):
    """Weight vectors at the edges of the CDF search (particle_filter.cpp:356-420, :443-503): all the mass on the first
    or on the last sample (zeros everywhere else), two equal masses at the ends, and a set of three samples."""
    import badger_amcl_amd as bpf
    rng = np.random.default_rng(5)
    n = 3 if kind == "tiny_set" else 2000
    s = np.zeros((n, 4))
    s[:, 0] = rng.uniform(10, 12, n); s[:, 1] = rng.uniform(20, 22, n); s[:, 2] = rng.uniform(-0.2, 0.2, n)
    if kind == "one_heavy":
        s[0, 3] = 1.0
    elif kind == "last_heavy":
        s[-1, 3] = 1.0
    elif kind == "two_equal":
        s[0, 3] = s[-1, 3] = 0.5
    else:
        s[:, 3] = [0.25, 0.5, 0.25]
    min_s = 2 if kind == "tiny_set" else 100
    pf = bpf.ParticleFilter(engine, min_s, n, 0.0, 0.0, 85.0)
    pf.srand48(11)
    pf.setResampleModel(resampler)
    pf.initWithSamples(s)
    st0 = pf.getState()
    pf.updateResample()
    after = pf.getCurrentSet()
    st1 = pf.getState()
    # No sensor update has run: w_slow = w_fast = 0 and the reference forms w_diff = 1 - 0 / 0 = NaN
    # (particle_filter.cpp:436-438).  Its multinomial loop treats that as "no random poses" (every comparison with NaN
    # is false); its systematic one converts NaN to int (:305, undefined behaviour).  The engine takes NaN as 0 in
    # both; the oracle is given equal averages, which is the same thing without the undefined step
    # (the node itself never resamples before it has scored).
    assert st0.w_slow == 0.0 and st0.w_fast == 0.0
    opf, out = _oracle_resample_from(orc, s, 1.0, 1.0, st0.leaf_count, 11, resampler, min_s, n)
    assert out.status == 0 and st1.last_status == 0
    M = out.sample_count
    assert st1.sample_count == M and st1.leaf_count == out.leaf_count and st1.bin_count == out.node_count
    assert np.array_equal(after.samples[:, :3], opf.samples[:M, :3])
    assert np.all(after.samples[:, 3] == 1.0 / M)
    assert pf.getRngState() == opf.pf.rng
