"""Every single-GPU BASELINE.json configuration AT ITS OWN SIZE against the oracle (VERDICT r01 "next" 1):

* cfg 2 -- 2-D likelihood field, 100 000 particles x 1081 beams, 2000^2 map: ALL 100 000 weights after
  `updateSensor` against `orc.planar_apply` + `orc_pf_normalize` (~4 s of oracle), then `updateResample` exact
  (count, poses, leaf / bin count, RNG state, convergence flag), converged AND spread cloud.
* cfg 3 -- beam model on the same map with range_max 30 m (601-cell rays): the whole set on the GPU, the oracle on
  a 2 048-particle slice x 1081 rays of it plus a 512-particle slice of a spread cloud (off-map and in-wall
  starts), and `OccupancyMap::calcRange` exact on 24 000 rays of that map.
* cfg 5 -- 3-D, 200 000 particles x 65 536 points: the whole set on the GPU, the oracle on a slice of it.

Reference: planar_scanner.cpp:168-323, occupancy_map.cpp:257-364, point_cloud_scanner.cpp:132-229,
particle_filter.cpp:223-267,356-471.  Weights within 1e-9 relative (north_star: 1e-6), everything else exact.
Knife-edge budget: an end point within ~1e-12 cells of a cell border may land in the
neighbouring cell, expected 2e-12 per evaluation -> at 1.08e8 evaluations the tests allow ONE weight beyond 1e-9.
"""
import math

import numpy as np
import pytest

from badger_amcl_amd import synth
from scenario import Scenario, rel_err

pytestmark = pytest.mark.gpu

W_TOL = 1e-9
N2, BEAMS, SIZE = 100000, 1081, 2000


@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


def _knife_edge_budget(n_evals):
    return 1 + int(n_evals * 2e-12)


@pytest.mark.parametrize("cloud", ["converged", "spread"])
@pytest.mark.parametrize("resampler", [0, 1])
def test_cfg2_all_weights_and_resample_match_oracle(engine, orc, cloud, resampler):
    """BASELINE config 2 at full size.  The scan carries 2 % max-range and 1 % NaN readings (skipped,
    planar_scanner.cpp:279-282); the spread cloud has particles off the map and inside walls (recalcWeight's
    branches, :642-682) and never stops early (100 000 draws: the device-side KLD tree)."""
    if resampler == 1 and cloud == "spread":
        pytest.skip("systematic x spread adds nothing over systematic x converged + multinomial x spread")
    sc_ = Scenario(orc, size=SIZE, n=N2, beams=BEAMS, cloud=cloud, frac_max=0.02, frac_nan=0.01,
                   scanner_pose=synth.SCANNER_POSE)
    m, sc, pf, data = sc_.gpu_objects(engine, BEAMS, "lf", min_samples=100, seed=7)
    pf.setResampleModel(resampler)
    assert sc.updateSensor(pf, data) is True
    before = pf.getCurrentSet().samples
    st0 = pf.getState()
    rng0 = pf.getRngState()

    opf = orc.ParticleFilter(100, N2, 0.0, 0.0, 85.0, seed=7)
    opf.set_resample_model(resampler)
    opf.set_samples(sc_.samples, leaf_count=st0.leaf_count)
    p = sc_.oracle_planar(BEAMS, "lf")
    stats = {}
    total = opf.update_sensor(lambda s, conv: sc_.oracle_apply(p, s, conv, stats))
    n_valid = int(np.sum(np.isfinite(sc_.ranges) & (sc_.ranges < sc_.range_max)))
    assert stats["evals"] == N2 * n_valid and n_valid > 1000
    assert np.array_equal(before[:, :3], sc_.samples[:, :3])
    bad = rel_err(before[:, 3], opf.samples[:N2, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(N2 * BEAMS), np.flatnonzero(bad)[:10]
    assert abs(st0.total - total) <= 1e-9 * total
    assert abs(st0.w_slow - opf.pf.w_slow) <= 1e-9 * opf.pf.w_slow
    assert abs(before[:, 3].sum() - 1.0) < 1e-11

    # resample: the oracle draws from the weights the GPU produced, so only the CDF's summation order differs
    # (a draw within ~1e-13 of a CDF step could pick the neighbour: none expected, none allowed here)
    pf.updateResample()
    st1 = pf.getState()
    after = pf.getCurrentSet().samples
    opf2 = orc.ParticleFilter(100, N2, 0.0, 0.0, 85.0)
    opf2.pf.rng = rng0
    opf2.set_resample_model(resampler)
    opf2.set_samples(before, leaf_count=st0.leaf_count)
    opf2.pf.w_slow, opf2.pf.w_fast = st0.w_slow, st0.w_fast
    out = opf2.update_resample()
    assert out.status == 0
    M = out.sample_count
    assert st1.sample_count == M and st1.leaf_count == out.leaf_count and st1.bin_count == out.node_count
    assert np.array_equal(after[:, :3], opf2.samples[:M, :3])
    assert np.all(after[:, 3] == 1.0 / M)
    assert pf.getRngState() == opf2.pf.rng
    assert st1.converged == out.converged and abs(st1.percent_converged - out.percent_converged) < 1e-4
    if cloud == "spread" and resampler == 0:
        assert M == N2 and st1.kld_on_device == 1
    if cloud == "converged":
        assert M < 20000
    # the oracle's own chain (its weights, its CDF): same count and leaf count, poses equal but for a draw that sat
    # on a rounding-level difference between the two CDFs (expected 1e-9 per draw)
    out_own = opf.update_resample()
    assert out_own.sample_count == M and out_own.leaf_count == st1.leaf_count
    differing = np.flatnonzero(np.any(after[:, :3] != opf.samples[:M, :3], axis=1))
    assert differing.size <= 1, differing[:5]


@pytest.mark.parametrize("model", ["gompertz", "prob"])
def test_cfg2_size_other_field_models_match_oracle(engine, orc, model):
    """The production model (Gompertz, badger_amcl_2d.launch:69) and the probabilistic one at the same size."""
    sc_ = Scenario(orc, size=SIZE, n=N2, beams=BEAMS, cloud="mixture", frac_max=0.02, frac_nan=0.01,
                   scanner_pose=synth.SCANNER_POSE)
    m, sc, pf, data = sc_.gpu_objects(engine, BEAMS, model, min_samples=100, seed=3)
    assert sc.updateSensor(pf, data) is True
    got = pf.getCurrentSet().samples
    opf = orc.ParticleFilter(100, N2, 0.0, 0.0, 85.0)
    opf.set_samples(sc_.samples, leaf_count=1)
    p = sc_.oracle_planar(BEAMS, model)
    opf.update_sensor(lambda s, conv: sc_.oracle_apply(p, s, conv))
    bad = rel_err(got[:, 3], opf.samples[:N2, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(N2 * BEAMS), np.flatnonzero(bad)[:10]


@pytest.fixture(scope="module")
def cfg3(orc):
    # the beam model does not skip NaN readings (planar_scanner.cpp:193-226), so none are injected; max-range
    # readings take the z_max branch
    return Scenario(orc, size=SIZE, n=N2, beams=BEAMS, cloud="converged", frac_max=0.02, frac_nan=0.0,
                    scanner_pose=synth.SCANNER_POSE, range_max=30.0)


def test_cfg3_beam_model_full_size_against_oracle_slices(engine, orc, cfg3):
    """BASELINE config 3: 100 000 x 1081 rays of up to 601 cells on the 2000^2 map.  The whole set runs on the
    GPU (resident path and host-buffer path); the oracle scores every 49th particle (2 048 of them, ~0.8 s) and
    512 particles of a spread cloud."""
    sc_ = cfg3
    m, sc, pf, data = sc_.gpu_objects(engine, BEAMS, "beam", min_samples=100, seed=7)
    raw = sc_.samples.copy()
    total = sc.applyModelToSampleSet(data, raw, 0)
    assert total > 0 and np.isfinite(total)
    p = sc_.oracle_planar(BEAMS, "beam")
    idx = np.arange(0, N2, 49)[:2048]
    want = np.ascontiguousarray(sc_.samples[idx])
    stats = {}
    sc_.oracle_apply(p, want, 0, stats)
    assert stats["evals"] == idx.size * BEAMS
    assert stats["cells"] / stats["evals"] > 30  # the rays really are long walks
    bad = rel_err(raw[idx, 3], want[:, 3]) > W_TOL
    assert bad.sum() <= _knife_edge_budget(idx.size * BEAMS), idx[np.flatnonzero(bad)[:10]]
    # resident path: the same weights, normalised by the same total
    assert sc.updateSensor(pf, data) is True
    cur = pf.getCurrentSet().samples
    st = pf.getState()
    assert abs(st.total - total) <= 1e-12 * total
    assert rel_err(cur[:, 3], raw[:, 3] / total).max() < 1e-13
    # a spread slice: rays that start off the map or inside a wall return 0 at once, others run 30 m
    sp = synth.spread_cloud(512, SIZE, seed=91)
    sp[:, 3] = np.random.default_rng(92).uniform(0.5, 1.5, 512) / 512
    got = sp.copy()
    sc.applyModelToSampleSet(data, got, 0)
    want = sp.copy()
    sc_.oracle_apply(p, want, 0)
    bad = rel_err(got[:, 3], want[:, 3]) > W_TOL
    assert bad.sum() <= 1, np.flatnonzero(bad)[:10]
    # resample from the beam model's weights: exact against the oracle
    rng0 = pf.getRngState()
    pf.updateResample()
    st1 = pf.getState()
    opf = orc.ParticleFilter(100, N2, 0.0, 0.0, 85.0)
    opf.pf.rng = rng0
    opf.set_samples(cur, leaf_count=st.leaf_count)
    opf.pf.w_slow, opf.pf.w_fast = st.w_slow, st.w_fast
    out = opf.update_resample()
    assert out.status == 0 and st1.sample_count == out.sample_count and st1.leaf_count == out.leaf_count
    assert np.array_equal(pf.getCurrentSet().samples[:, :3], opf.samples[:out.sample_count, :3])
    assert pf.getRngState() == opf.pf.rng


def test_cfg3_calc_range_exact_on_the_2000_map(engine, orc, cfg3):
    """occupancy_map.cpp:257-364 on the headline map: 24 000 rays, a third of them with the full 30 m / 601 cells,
    starts on and off the map, all octants plus the exact axis and diagonal directions."""
    sc_ = cfg3
    m, sc, pf, data = sc_.gpu_objects(engine, BEAMS, "beam")
    rng = np.random.default_rng(12)
    n = 24000
    ext = SIZE * 0.05
    ox, oy = rng.uniform(-1.0, ext + 1.0, n), rng.uniform(-1.0, ext + 1.0, n)
    oa = rng.uniform(-math.pi, math.pi, n)
    oa[:64] = np.repeat(np.arange(-4, 4) * (math.pi / 4), 8)
    mr = rng.choice([0.3, 8.0, 30.0], n)
    mr[:64] = 30.0
    got = m.calcRange(ox, oy, oa, mr)
    want = np.array([sc_.omap.calc_range(float(a), float(b), float(c), float(d))
                     for a, b, c, d in zip(ox, oy, oa, mr)])
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
    assert (want < mr).mean() > 0.3 and (want == mr).mean() > 0.02
    assert (want[mr == 30.0] > 3.0).mean() > 0.1


def test_calc_range_beyond_the_24_bit_capacity_is_refused(engine, orc):
    """the jump walk forms j * 2*dmin with 24-bit multiplies; rays of 32 760 cells or more are refused
    (BPF_ERR_CAPACITY) rather than walked wrongly."""
    import badger_amcl_amd as bpf
    sc_ = Scenario(orc, size=200, n=4, beams=11)
    m, sc, pf, data = sc_.gpu_objects(engine, 11, "beam")
    one = np.array([5.0])
    ok = m.calcRange(one, one, np.array([0.1]), np.array([32000 * 0.05]))
    assert ok.shape == (1,)
    with pytest.raises(bpf.BpfError):
        m.calcRange(one, one, np.array([0.1]), np.array([32768 * 0.05]))
    with pytest.raises(bpf.BpfError):
        big = bpf.PlanarData(np.full(11, 3.0), sc_.angles, 32768 * 0.05)
        sc.applyModelToSampleSet(big, sc_.samples.copy(), 0)


def test_cfg5_cloud3d_full_size_against_oracle_slice(engine, orc):
    """BASELINE config 5: 200 000 particles x 65 536 points (1.3e10 evaluations).  Whole set on the GPU through the
    resident path; the oracle scores 64 particles spread over the set (each against all 65 536 points), including
    some pushed off the map; then the resample is checked exactly against the oracle on the GPU's weights."""
    import badger_amcl_amd as bpf
    pi, dr, mn, mx = synth.box_room_lut()
    pts = synth.grid_cloud(64, 1024)
    assert pts.shape[0] == 65536
    n = 200000
    rng = np.random.default_rng(8)
    s = np.zeros((n, 4))
    s[:, 0] = 0.3 + rng.normal(0, 0.1, n)
    s[:, 1] = 0.2 + rng.normal(0, 0.1, n)
    s[:, 2] = rng.normal(0, 0.05, n)
    s[::1000, 0] += 30.0  # off the map: recalcWeight's off-map factor (point_cloud_scanner.cpp:205-229)
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
    raw = s.copy()
    total = sc.applyModelToSampleSet(data, raw)
    idx = np.unique(np.concatenate([np.linspace(0, n - 1, 56).astype(int), np.arange(0, 8000, 1000)]))
    olut = orc.OctoMapLUT(mn, mx, 0.05, 0.3, pi, dr)
    op = orc.cloud(orc.CLOUD_MODEL, 65536, tf_xyz, tf_quat, z_hit=0.5, z_rand=0.05, sigma_hit=0.1)
    op.off_map_factor = 0.95
    want = np.ascontiguousarray(s[idx])
    stats = {}
    orc.cloud_apply(op, olut, want, pts, stats)
    assert stats["evals"] == idx.size * 65536
    bad = rel_err(raw[idx, 3], want[:, 3]) > W_TOL
    assert bad.sum() <= 1, idx[np.flatnonzero(bad)]  # a point within rounding of a voxel face
    # resident path + resample
    pf = bpf.ParticleFilter(engine, 100, n, 0.0, 0.0, 85.0)
    pf.srand48(5)
    pf.initWithSamples(s)
    assert sc.updateSensor(pf, data)
    cur = pf.getCurrentSet().samples
    st = pf.getState()
    assert abs(st.total - total) <= 1e-12 * total
    assert rel_err(cur[:, 3], raw[:, 3] / total).max() < 1e-13
    rng0 = pf.getRngState()
    pf.updateResample()
    st1 = pf.getState()
    opf = orc.ParticleFilter(100, n, 0.0, 0.0, 85.0)
    opf.pf.rng = rng0
    opf.set_samples(cur, leaf_count=st.leaf_count)
    opf.pf.w_slow, opf.pf.w_fast = st.w_slow, st.w_fast
    out = opf.update_resample()
    assert out.status == 0 and st1.sample_count == out.sample_count and st1.leaf_count == out.leaf_count
    assert np.array_equal(pf.getCurrentSet().samples[:, :3], opf.samples[:out.sample_count, :3])
    assert pf.getRngState() == opf.pf.rng
