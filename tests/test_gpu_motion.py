"""Motion update on the device through the C-ABI against the oracle.

Exact: the drand48 state after the update (every rejected attempt and r == 0 re-draw accounted for)
and the untouched weights.  Poses: the device's log / sin / cos differ from glibc's by an ulp or two,
so x / y / theta are compared with an absolute tolerance of 1e-12 (values are O(10); observed ~1e-15)."""
import numpy as np
import pytest

from badger_amcl_amd import synth

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-12
A, C_, MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1
ALPHA = (0.2, 0.15, 0.25, 0.1, 0.3)
ODATA = dict(pose=(3.0, -1.0, 0.7), delta=(0.21, -0.08, 0.12), absm=(0.25, 0.09, 0.15))


@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


def _filter(engine, samples, rng_state, max_samples=None):
    import badger_amcl_amd as bpf
    pf = bpf.ParticleFilter(engine, 10, max_samples or samples.shape[0], 0.0, 0.0, 85.0)
    pf.setRngState(rng_state)
    pf.initWithSamples(samples, leaf_count=1)
    return pf


def _cloud(n, seed=0):
    s = synth.spread_cloud(n, 400, seed=seed)
    s[:, 3] = np.random.default_rng(seed).uniform(0.1, 1.0, n)
    return s


@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("n", [1, 777, 20000])
def test_update_action_matches_oracle(engine, orc, model, n):
    import badger_amcl_amd as bpf
    s = _cloud(n, seed=model + n)
    rng0 = 0x5A5A1234330E ^ (model * 0x1111)
    pf = _filter(engine, s, rng0)
    od = bpf.Odom(engine)
    od.setModel(model, *ALPHA)
    od.updateAction(pf, bpf.OdomData(ODATA["pose"], ODATA["delta"], ODATA["absm"]))
    got = pf.getCurrentSet().samples
    want = s.copy()
    st = orc.odom_update_action(model, ALPHA, ODATA["pose"], ODATA["delta"], ODATA["absm"], want, rng0)
    assert pf.getRngState() == st
    assert np.array_equal(got[:, 3], want[:, 3])
    assert np.abs(got[:, :3] - want[:, :3]).max() <= POSE_TOL
    assert pf.getState().sample_count == n


def test_two_updates_then_sensor_and_resample(engine, orc):
    """The whole predict -> score -> resample cycle stays on the device; the stream position the
    resampler starts from is the one the motion update left."""
    import badger_amcl_amd as bpf
    from scenario import Scenario
    sc_ = Scenario(orc, size=400, n=5000, beams=91, cloud="converged")
    m, sc, pf, data = sc_.gpu_objects(engine, 91, "lf", min_samples=100, seed=21)
    od = bpf.Odom(engine)
    od.setModel(bpf.pf.ODOM_MODEL_DIFF_CORRECTED, 0.05, 0.05, 0.05, 0.05)
    odata = bpf.OdomData((1.0, 2.0, 0.3), (0.02, 0.01, 0.01))
    opf = orc.ParticleFilter(100, 5000, 0.0, 0.0, 85.0, seed=21)
    opf.set_samples(sc_.samples)
    p = sc_.oracle_planar(91, "lf")
    for _ in range(2):
        od.updateAction(pf, odata)
        cur = opf.samples[:opf.sample_count]
        opf.pf.rng = orc.odom_update_action(2, (0.05, 0.05, 0.05, 0.05, 0.0), odata.pose, odata.delta,
                                            odata.absolute_motion, cur, opf.pf.rng)
        assert pf.getRngState() == opf.pf.rng
    assert np.abs(pf.getCurrentSet().samples[:, :3] - opf.samples[:5000, :3]).max() <= POSE_TOL
    # continue from the DEVICE poses on both sides, so the comparison below is exact again
    dev = pf.getCurrentSet().samples
    opf.set_samples(dev)
    sc.updateSensor(pf, data)
    pf.updateResample()
    opf.update_sensor(lambda s, conv: sc_.oracle_apply(p, s, conv))
    out = opf.update_resample()
    assert pf.getState().sample_count == out.sample_count
    assert pf.getRngState() == opf.pf.rng
    assert np.array_equal(pf.getCurrentSet().samples[:, :3], opf.samples[:out.sample_count, :3])


@pytest.mark.parametrize("steps_before_zero", [1, 2, 3, 4, 1001, 2500])
def test_exact_zero_in_the_stream_is_redrawn(engine, orc, steps_before_zero):
    """pdf_gaussian.cpp:83-92: `do r = drand48(); while (r == 0.0)`.  The generator passes through
    state 0 once per period; start the stream `steps_before_zero` steps ahead of it."""
    import badger_amcl_amd as bpf
    inv_a = pow(A, -1, 1 << 48)
    st = 0
    for _ in range(steps_before_zero):
        st = ((st - C_) * inv_a) & MASK
    n = 400  # ~3060 uniforms: every tested position lies inside the update
    s = _cloud(n, seed=5)
    pf = _filter(engine, s, st)
    od = bpf.Odom(engine)
    od.setModel(3, *ALPHA)
    od.updateAction(pf, bpf.OdomData(ODATA["pose"], ODATA["delta"], ODATA["absm"]))
    want = s.copy()
    st_after = orc.odom_update_action(3, ALPHA, ODATA["pose"], ODATA["delta"], ODATA["absm"], want, st)
    assert pf.getRngState() == st_after
    assert np.abs(pf.getCurrentSet().samples[:, :3] - want[:, :3]).max() <= POSE_TOL


def test_sharded_update_equals_whole(engine, orc):
    """Two engines, each holding a contiguous half: same Gaussians per global index, same final state."""
    import badger_amcl_amd as bpf
    n = 9001
    s = _cloud(n, seed=8)
    rng0 = 0xDEADBEEF330E
    pf = _filter(engine, s, rng0)
    od = bpf.Odom(engine)
    od.setModel(0, *ALPHA)
    data = bpf.OdomData(ODATA["pose"], ODATA["delta"], ODATA["absm"])
    od.updateAction(pf, data)
    whole = pf.getCurrentSet().samples
    cut = 4000
    parts = []
    for lo, hi in [(0, cut), (cut, n)]:
        e2 = bpf.Engine(0)
        try:
            pf2 = _filter(e2, s[lo:hi], rng0, max_samples=n)
            od2 = bpf.Odom(e2)
            od2.setModel(0, *ALPHA)
            od2.updateActionShard(data, lo, n)
            parts.append(pf2.getCurrentSet().samples)
            assert pf2.getRngState() == pf.getRngState()
        finally:
            e2.close()
    assert np.array_equal(np.concatenate(parts), whole)


@pytest.mark.parametrize("n", [500, 20000])
def test_init_with_gaussian_matches_oracle(engine, orc, n):
    """ParticleFilter::initWithGaussian given PDFGaussian's decomposition: drand48 state and leaf / bin counts exact,
    poses within POSE_TOL (device log), weights 1/max_samples, averages zeroed, not converged."""
    import badger_amcl_amd as bpf
    mean = (12.5, -3.25, 0.8)
    ang = 0.4
    cr = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    cd = (0.5, 0.2, 0.1)
    pf = bpf.ParticleFilter(engine, 10, n, 0.001, 0.1, 85.0)
    pf.setRngState(0xABCDEF12330E)
    pf.initWithGaussian(mean, cr, cd)
    opf = orc.ParticleFilter(10, n, 0.001, 0.1, 85.0)
    opf.pf.rng = 0xABCDEF12330E
    opf.init_with_gaussian(mean, cr, cd)
    st = pf.getState()
    got = pf.getCurrentSet().samples
    assert st.sample_count == n and pf.getRngState() == opf.pf.rng
    assert np.abs(got[:, :3] - opf.samples[:, :3]).max() <= POSE_TOL
    assert np.all(got[:, 3] == 1.0 / n)
    assert st.w_slow == 0.0 and st.w_fast == 0.0 and st.converged == 0
    # the tree is built from the DEVICE poses: compare with the oracle's tree on those (a pose within 1e-15 of
    # a bin edge could differ from the oracle's own sample)
    t = orc.KDTree()
    for r in got[:, :3]:
        t.insert_pose(r, 1.0)
    assert (st.leaf_count, st.bin_count) == (t.leaf_count(), t.node_count())


@pytest.mark.parametrize("n", [700, 30000])
def test_init_with_random_free_space_poses_matches_oracle(engine, orc, n):
    """ParticleFilter::initWithPoseFn(Node::randomFreeSpacePose) -- global localisation: everything exact."""
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    from scenario import Scenario
    sc_ = Scenario(orc, size=200, n=64, beams=61)
    m, sc, pf0, data = sc_.gpu_objects(engine, 61, "lf")
    pf = bpf.ParticleFilter(engine, 10, n, 0.0, 0.0, 85.0)
    pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
    pf.srand48(99)
    pf.initWithRandomPoses()
    opf = orc.ParticleFilter(10, n, 0.0, 0.0, 85.0, seed=99)
    assert opf.set_random_pose_source(sc_.omap, sc_.map_factors[2]) > 1000
    opf.init_with_free_space_poses()
    st = pf.getState()
    got = pf.getCurrentSet().samples
    assert np.array_equal(got, opf.samples)
    assert pf.getRngState() == opf.pf.rng
    assert (st.sample_count, st.leaf_count, st.bin_count) == (n, opf.leaf_count, opf.node_count)
    # and the filter works from there
    sc.updateSensor(pf, data)
    pf.updateResample()
    assert 10 <= pf.getState().sample_count <= n
