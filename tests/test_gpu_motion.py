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
