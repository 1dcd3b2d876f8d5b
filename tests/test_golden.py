"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py): the oracle must
keep reproducing them (CPU), and the HIP path must hit them without the oracle in the loop (GPU)."""
import glob
import os

import numpy as np
import pytest

from scenario import Scenario, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "planar_*.npz")))


def _model_of(path):
    return os.path.basename(path)[len("planar_"):-len(".npz")]


def test_fixtures_present():
    assert {_model_of(p) for p in FIXTURES} == {"lf", "gompertz", "prob", "beam"}


@pytest.mark.parametrize("path", FIXTURES, ids=_model_of)
def test_oracle_reproduces_golden(orc, path):
    g = np.load(path)
    model = _model_of(path)
    sc = Scenario(orc, size=120, n=200, beams=61, cloud="mixture", max_dist=1.0, seed=17,
                      frac_nan=0.0 if model == "beam" else 0.02)  # the beam model does not skip NaN ranges
    assert np.array_equal(sc.cells.astype(np.int8), g["cells"]) and np.array_equal(sc.lut, g["lut"])
    assert np.array_equal(sc.samples, g["samples"])
    mb = int(g["max_beams"])
    w = sc.samples.copy()
    total = sc.oracle_apply(sc.oracle_planar(mb, model), w)
    assert np.array_equal(w[:, 3], g["weights_after_apply"]) and total == float(g["total"])
    opf = orc.ParticleFilter(20, 200, 0.0, 0.0, 85.0, seed=5)
    opf.set_population_size_parameters(0.05, 2.0)
    opf.set_samples(sc.samples)
    p = sc.oracle_planar(mb, model)
    opf.update_sensor(lambda s, c: sc.oracle_apply(p, s, c))
    assert np.array_equal(opf.samples[:200, 3], g["weights_normalized"])
    out = opf.update_resample()
    assert out.sample_count == int(g["sample_count"]) and out.leaf_count == int(g["leaf_count"])
    assert np.array_equal(opf.samples[:out.sample_count], g["resampled"])
    assert int(opf.pf.rng) == int(g["rng_after"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=_model_of)
def test_gpu_hits_golden(path):
    import badger_amcl_amd as bpf
    from badger_amcl_amd import synth
    g = np.load(path)
    model = _model_of(path)
    e = bpf.Engine(0)
    m = bpf.OccupancyMap(e, 0.05)
    m.setCells(g["cells"].astype(np.int32))
    m.setOrigin(g["origin"])
    m.setDistancesLUT(g["lut"], float(g["max_dist"]))
    sc = bpf.PlanarScanner(e)
    sc.init(int(g["max_beams"]), m)
    helper = Scenario.__new__(Scenario)
    helper.max_dist = float(g["max_dist"])
    helper.configure_gpu_model(sc, model)
    sc.setMapFactors(*g["map_factors"])
    sc.setPlanarScannerPose(g["scanner_pose"])
    data = bpf.PlanarData(g["ranges"], g["angles"], float(g["range_max"]))
    s = np.ascontiguousarray(g["samples"])
    got = s.copy()
    total = sc.applyModelToSampleSet(data, got, 0)
    assert rel_err(got[:, 3], g["weights_after_apply"]).max() <= 1e-9
    assert abs(total - float(g["total"])) <= 1e-9 * float(g["total"])
    pf = bpf.ParticleFilter(e, 20, 200, 0.0, 0.0, 85.0)
    pf.setPopulationSizeParameters(0.05, 2.0)
    pf.srand48(5)
    pf.initWithSamples(s)
    sc.updateSensor(pf, data)
    assert rel_err(pf.getCurrentSet().samples[:, 3], g["weights_normalized"]).max() <= 1e-9
    pf.updateResample()
    st = pf.getState()
    assert st.sample_count == int(g["sample_count"]) and st.leaf_count == int(g["leaf_count"])
    assert st.bin_count == int(g["bin_count"]) and st.converged == int(g["converged"])
    cur = pf.getCurrentSet().samples
    assert np.array_equal(cur[:, :3], g["resampled"][:, :3]) and np.array_equal(cur[:, 3], g["resampled"][:, 3])
    assert pf.getRngState() == int(g["rng_after"])
    e.close()


# ---- full cycles: motion -> sensor -> resample (second cycle with recovery random poses) + set statistics
CYCLES = sorted(glob.glob(os.path.join(HERE, "golden", "cycle_*.npz")))


def _resampler_of(path):
    return ("multinomial", "systematic").index(os.path.basename(path)[len("cycle_"):-len(".npz")])


def test_cycle_fixtures_present():
    assert len(CYCLES) == 2


@pytest.mark.parametrize("path", CYCLES, ids=_resampler_of)
def test_oracle_reproduces_cycle_golden(orc, path):
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden
    g = np.load(path)
    sc = Scenario(orc, size=120, n=600, beams=61, cloud="mixture", max_dist=1.0, seed=23)
    assert np.array_equal(sc.samples, g["samples"]) and np.array_equal(sc.lut, g["lut"])
    rec = make_golden.run_cycle(orc, sc, _resampler_of(path))
    for k, v in rec.items():
        if k.startswith("moved"):
            # glibc's sincos differs from sin / cos by an ulp for a few arguments and gcc may pick either
            assert np.abs(np.asarray(v) - g[k]).max() <= 1e-14, k
        else:
            assert np.array_equal(np.asarray(v), g[k], equal_nan=True), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", CYCLES, ids=_resampler_of)
def test_gpu_hits_cycle_golden(path):
    """The whole predict -> score -> resample cycle on the device against the committed vectors, no oracle in the
    loop.  Each stage starts from the fixture's own input of that stage, so the motion update's few-ulp pose
    differences (device log / sin / cos) do not leak into the exact comparisons after it."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    import badger_amcl_amd as bpf
    import badger_amcl_amd.pf as hpf
    g = np.load(path)
    resampler = _resampler_of(path)
    e = bpf.Engine(0)
    e.set_option(hpf.OPT_CDF_SERIAL, 1)
    m = bpf.OccupancyMap(e, 0.05)
    m.setCells(g["cells"].astype(np.int32))
    m.setOrigin(g["origin"])
    m.setDistancesLUT(g["lut"], float(g["max_dist"]))
    sc = bpf.PlanarScanner(e)
    sc.init(61, m)
    helper = Scenario.__new__(Scenario)
    helper.max_dist = float(g["max_dist"])
    helper.configure_gpu_model(sc, "lf")
    sc.setMapFactors(*g["map_factors"])
    sc.setPlanarScannerPose(g["scanner_pose"])
    n = g["samples"].shape[0]
    pf = bpf.ParticleFilter(e, 50, n, mg.CYCLE_ALPHA[0], mg.CYCLE_ALPHA[1], 85.0)
    pf.setResampleModel(resampler)
    pf.setRandomPoseGenerator(hpf.RANDOM_POSE_FREE_SPACE_2D)
    pf.srand48(77)
    pf.initWithSamples(np.ascontiguousarray(g["samples"]))
    od = bpf.Odom(e)
    od.setModel(mg.CYCLE_ODOM[0], *mg.CYCLE_ODOM[1])
    scans = [g["ranges"], np.clip(g["ranges"] * 0.6, 0.05, 29.0)]
    for c, ranges in enumerate(scans):
        od.updateAction(pf, bpf.OdomData(*mg.CYCLE_ODATA))
        moved = pf.getCurrentSet().samples
        assert np.abs(moved[:, :3] - g["moved%d" % c][:, :3]).max() <= 1e-12
        sc.updateSensor(pf, bpf.PlanarData(ranges, g["angles"], float(g["range_max"])))
        w = pf.getCurrentSet().samples[:, 3]
        assert rel_err(w, g["weights%d" % c]).max() <= 1e-9
        pf.updateResample()
        st = pf.getState()
        M, leaf, bins, clusters, conv = (int(v) for v in g["scalars%d" % c])
        assert abs(st.w_diff - float(g["w_diff%d" % c])) <= 1e-9
        assert st.sample_count == M
        assert pf.getRngState() == int(g["rng%d" % c])
        cur = pf.getCurrentSet().samples
        # poses are copies of moved poses (few-ulp device libm) or random free-space poses (exact)
        assert np.abs(cur[:, :3] - g["resampled%d" % c][:, :3]).max() <= 1e-12
        assert np.all(cur[:, 3] == 1.0 / M)
        assert (st.leaf_count, st.bin_count, st.converged) == (leaf, bins, conv)
        k, mean, cov = pf.computeClusterStats()
        assert k == clusters
        assert np.abs(mean - g["mean%d" % c]).max() <= 1e-9
    e.close()
