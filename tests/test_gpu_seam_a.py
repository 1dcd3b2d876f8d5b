"""Seam A with the particle set in HOST memory (PlanarScanner::applyModelToSampleSet on the reference's own
std::vector<PFSample>, planar_scanner.cpp:141-164; the set is allocated once, particle_filter.cpp:62-89): the pipelined
form (chunks up on a copy stream, each scored while the next crosses PCIe, the weights stored into pinned host memory
by the scoring launches themselves and written into the records by the calling thread chunk by chunk) against the
oracle and against the plain upload / score / download sequence, with registered, unregistered, unaligned and moving
buffers; the lazily built histogram tree of an adopted set against the eager one."""
import gc

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


@pytest.fixture(scope="module")
def big(engine, orc):
    """70 000 particles x 91 beams: several trips per wave of the scoring kernel, small enough for the oracle."""
    sc_ = Scenario(orc, size=300, n=70000, beams=91, cloud="mixture")
    m, sc, pf, data = sc_.gpu_objects(engine, 91, "lf")
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(91, "lf"), want)
    return dict(sc_=sc_, sc=sc, pf=pf, data=data, want=want, want_total=want_total, keep=(m,))


def _check(got, total, big):
    assert np.array_equal(got[:, :3], big["want"][:, :3])
    bad = rel_err(got[:, 3], big["want"][:, 3]) > W_TOL
    assert bad.sum() <= 1 + int(got.shape[0] * 91 * 2e-12), np.flatnonzero(bad)[:10]
    assert abs(total - big["want_total"]) <= 1e-9 * abs(big["want_total"])


BIG_CHUNKS = 2  # sets of 40 000 particles or more go through in two chunks ...
IN_PLACE = -1   # ... unless the buffer is registered (and 16-byte aligned): one launch reads and writes the records in place


def test_pipelined_form_equals_the_plain_sequence_and_the_oracle(engine, big):
    import badger_amcl_amd as bpf
    sc, data, sc_ = big["sc"], big["data"], big["sc_"]
    plain = sc_.samples.copy()
    engine.set_option(bpf.pf.OPT_SEAM_CHUNKS, 1)
    try:
        t_plain = sc.applyModelToSampleSet(data, plain, 0)
        assert engine.seam_last_plan() == (0, False)  # upload / score / download
    finally:
        engine.set_option(bpf.pf.OPT_SEAM_CHUNKS, 0)
    _check(plain, t_plain, big)
    reg = sc_.samples.copy()
    engine.registerHostBuffer(reg)
    try:
        for chunks in (0, 2, 3, 4, 7):
            for buf, pinned in ((sc_.samples.copy(), False), (reg, True)):
                engine.set_option(bpf.pf.OPT_SEAM_CHUNKS, chunks)
                try:
                    buf[:] = sc_.samples
                    total = sc.applyModelToSampleSet(data, buf, 0)
                    want_plan = IN_PLACE if (pinned and chunks == 0) else (chunks or BIG_CHUNKS)
                    assert engine.seam_last_plan() == (want_plan, pinned)
                finally:
                    engine.set_option(bpf.pf.OPT_SEAM_CHUNKS, 0)
                # same kernels on the same particles: the weights are the same bits; the total is the sum of the
                # chunks' totals, the plain sequence's one fixed-shape sum
                diff = np.flatnonzero((buf != plain).any(axis=1))
                assert diff.size == 0 and abs(total - t_plain) <= 1e-13 * t_plain, (
                    "chunks %d pinned %s: %d rows differ, first %s last %s; row %s vs %s; total %r vs %r" % (
                        chunks, pinned, diff.size, diff[:4], diff[-4:], buf[diff[0]] if diff.size else None,
                        plain[diff[0]] if diff.size else None, total, t_plain))
    finally:
        engine.unregisterHostBuffer(reg)


def test_registered_buffer(engine, big):
    sc, data, sc_ = big["sc"], big["data"], big["sc_"]
    buf = sc_.samples.copy()
    engine.registerHostBuffer(buf)
    try:
        assert engine.isHostBufferRegistered(buf) and engine.isHostBufferRegistered(buf[100:5000])
        for _ in range(3):
            buf[:] = sc_.samples
            total = sc.applyModelToSampleSet(data, buf, 0)
            assert engine.seam_last_plan() == (IN_PLACE, True)
            _check(buf, total, big)
        # the live prefix of a registered buffer (sample_count < max_samples) is read as pinned memory too
        buf[:] = sc_.samples
        sc.applyModelToSampleSet(data, buf[:40000], 0)
        assert engine.seam_last_plan() == (IN_PLACE, True)
        assert np.array_equal(buf[40000:], sc_.samples[40000:])
        assert rel_err(buf[:40000, 3], big["want"][:40000, 3]).max() <= W_TOL
        # the filter's own host-buffer calls take the same registration
        pf = big["pf"]
        pf.initWithSamples(buf)
        out = pf.getCurrentSet()
        assert np.array_equal(out.samples, buf)
    finally:
        engine.unregisterHostBuffer(buf)
    assert not engine.isHostBufferRegistered(buf)
    with pytest.raises(Exception):
        engine.unregisterHostBuffer(buf)


def test_unregistered_unaligned_buffer(engine, big):
    sc, data, sc_ = big["sc"], big["data"], big["sc_"]
    n = sc_.samples.shape[0]
    raw = np.empty(n * 4 + 3, dtype=np.float64)
    for shift in (1, 3):  # 8-byte aligned only: neither 16 nor 32, and not on a page boundary
        got = raw[shift:shift + 4 * n].reshape(n, 4)
        assert got.ctypes.data % 16 != 0 or got.ctypes.data % 32 != 0
        got[:] = sc_.samples
        total = sc.applyModelToSampleSet(data, got, 0)
        assert engine.seam_last_plan() == (BIG_CHUNKS, False)
        _check(got, total, big)
    # ... and the same unaligned range, registered: pinned for the copy engine, but not read in place (the in-place
    # launch moves the records as 16-byte pairs)
    got = raw[1:1 + 4 * n].reshape(n, 4)
    got[:] = sc_.samples
    engine.registerHostBuffer(got)
    try:
        total = sc.applyModelToSampleSet(data, got, 0)
        assert engine.seam_last_plan() == ((IN_PLACE if got.ctypes.data % 16 == 0 else BIG_CHUNKS), True)
        _check(got, total, big)
    finally:
        engine.unregisterHostBuffer(got)


def test_a_buffer_that_moves_between_calls(engine, big):
    """Registered buffer A, then the same set at another address (not registered), then A is unregistered and freed
    and a new buffer -- quite possibly at A's old address -- takes its place: every call reads the memory it is
    given, never a stale mapping."""
    sc, data, sc_ = big["sc"], big["data"], big["sc_"]
    a = sc_.samples.copy()
    engine.registerHostBuffer(a)
    total = sc.applyModelToSampleSet(data, a, 0)
    assert engine.seam_last_plan() == (IN_PLACE, True)
    _check(a, total, big)
    addr_a = a.ctypes.data
    b = sc_.samples.copy()
    assert b.ctypes.data != addr_a
    total = sc.applyModelToSampleSet(data, b, 0)
    assert engine.seam_last_plan() == (BIG_CHUNKS, False)
    _check(b, total, big)
    engine.unregisterHostBuffer(a)
    del a
    gc.collect()
    for k in range(12):
        c = sc_.samples.copy()   # the allocator may hand A's pages out again
        c[:, 3] *= (k + 2.0)     # different contents each time: a stale mapping would show
        total = sc.applyModelToSampleSet(data, c, 0)
        assert engine.seam_last_plan() == (BIG_CHUNKS, False)
        assert rel_err(c[:, 3], big["want"][:, 3] * (k + 2.0)).max() <= W_TOL
        assert abs(total - big["want_total"] * (k + 2.0)) <= 1e-9 * abs(total)
        del c


def test_auto_registration_is_opt_in(engine, big):
    import badger_amcl_amd as bpf
    sc, data, sc_ = big["sc"], big["data"], big["sc_"]
    buf = sc_.samples.copy()
    sc.applyModelToSampleSet(data, buf, 0)
    assert not engine.isHostBufferRegistered(buf)  # never on its own
    engine.set_option(bpf.pf.OPT_HOST_AUTO_REGISTER, 1)
    try:
        buf[:] = sc_.samples
        total = sc.applyModelToSampleSet(data, buf, 0)
        assert engine.seam_last_plan() == (IN_PLACE, True) and engine.isHostBufferRegistered(buf)
        _check(buf, total, big)
        buf[:] = sc_.samples
        total = sc.applyModelToSampleSet(data, buf, 0)
        _check(buf, total, big)
    finally:
        engine.set_option(bpf.pf.OPT_HOST_AUTO_REGISTER, 0)
        engine.unregisterHostBuffer(buf)


@pytest.mark.parametrize("model", ["gompertz", "prob"])
def test_pipelined_form_other_field_models(engine, orc, model):
    sc_ = Scenario(orc, size=200, n=40000, beams=61, cloud="converged")
    m, sc, pf, data = sc_.gpu_objects(engine, 61, model)
    want = sc_.samples.copy()
    want_total = sc_.oracle_apply(sc_.oracle_planar(61, model), want)
    got = sc_.samples.copy()
    engine.registerHostBuffer(got)
    try:
        total = sc.applyModelToSampleSet(data, got, 0)
        assert engine.seam_last_plan() == (IN_PLACE, True)
    finally:
        engine.unregisterHostBuffer(got)
    bad = rel_err(got[:, 3], want[:, 3]) > W_TOL
    assert bad.sum() <= 1
    assert abs(total - want_total) <= 1e-9 * abs(want_total)


def test_beam_model_and_beam_skipping_keep_the_plain_sequence(engine, orc):
    sc_ = Scenario(orc, size=200, n=40000, beams=61, cloud="converged", frac_nan=0.0)
    m, sc, pf, data = sc_.gpu_objects(engine, 31, "beam")
    got = sc_.samples.copy()
    engine.registerHostBuffer(got)
    try:
        sc.applyModelToSampleSet(data, got, 0)
        assert engine.seam_last_plan() == (0, True)
    finally:
        engine.unregisterHostBuffer(got)
    want = sc_.samples.copy()
    sc_.oracle_apply(sc_.oracle_planar(31, "beam"), want)
    assert (rel_err(got[:, 3], want[:, 3]) > W_TOL).sum() <= 1


def test_the_tree_of_an_adopted_set_is_built_when_it_is_first_needed(engine, orc):
    """bpf_pf_set_samples defers the histogram tree (leaf count) of the adopted set; whoever needs it gets the value
    an eager build gives: get_state, the systematic resampler, and -- since the reference builds the tree when the set
    is created -- a motion update must not change it."""
    import badger_amcl_amd as bpf
    sc_ = Scenario(orc, size=200, n=20000, beams=61, cloud="spread")
    m, sc, pf, data = sc_.gpu_objects(engine, 61, "lf", max_samples=20000)
    otree = orc.ParticleFilter(100, 20000, 0.0, 0.0, 85.0, seed=42)
    otree.set_samples(sc_.samples)   # builds the oracle's kd-tree of the set
    want_leaf = otree.leaf_count
    pf.initWithSamples(sc_.samples)
    assert pf.getState().leaf_count == want_leaf
    # motion first: the leaf count stays that of the poses the set was created with
    pf.initWithSamples(sc_.samples)
    odom = bpf.Odom(engine)
    odom.setModel(0, 0.05, 0.05, 0.05, 0.05, 0.05)
    odom.updateAction(pf, bpf.OdomData((1.0, 2.0, 0.3), (0.2, 0.05, 0.1)))
    assert pf.getState().leaf_count == want_leaf
    # systematic resampling sizes the new set from it
    for resampler in (1, 0):
        pf.setResampleModel(resampler)
        pf.srand48(7)
        pf.initWithSamples(sc_.samples)
        sc.updateSensor(pf, data)
        pf.updateResample()
        st = pf.getState()
        opf = orc.ParticleFilter(100, 20000, 0.0, 0.0, 85.0, seed=7)
        opf.set_resample_model(resampler)
        opf.set_samples(sc_.samples)
        p = sc_.oracle_planar(61, "lf")
        opf.update_sensor(lambda s, c: sc_.oracle_apply(p, s, c))
        out = opf.update_resample()
        assert (st.sample_count, st.leaf_count) == (out.sample_count, out.leaf_count)
    pf.setResampleModel(0)


def test_pageable_buffers_that_come_and_go(engine):
    """Sets of more than a megabyte in pageable memory that is freed and allocated again between the calls -- the
    allocator hands the same addresses out again, and a cache of on-the-fly pins keyed by address (the HIP runtime
    keeps one for pageable copies) would read the pages of a buffer that is gone: every set that goes in comes back
    out, bit for bit (pageable memory goes through the engine's own bounce buffer, host_common.inl)."""
    import badger_amcl_amd as bpf
    rng = np.random.default_rng(3)
    n = 60000
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
    for k in range(25):
        s = rng.uniform(-50.0, 50.0, (n, 4))
        s[:, 3] = rng.uniform(0.1, 1.0, n)
        pf.initWithSamples(s, leaf_count=1)
        want = s.copy()
        del s
        junk = np.full((n, 4), float(k))  # quite possibly where s was
        got = pf.getCurrentSet().samples
        assert np.array_equal(got, want), k
        del junk, got
