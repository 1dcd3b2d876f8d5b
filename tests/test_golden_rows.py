"""Committed fixtures tests/golden/rows_*.npz (made by tests/golden/make_golden_rows.py) for the rows of SURVEY.md
section 8 the planar_* / cycle_* fixtures do not cover: beam skipping, the ray walk, the 3-D models, the five motion
models and the 2-D distance-LUT builder in reference order.  CPU: the oracle keeps reproducing them.  GPU: the HIP
path hits them through the C-ABI without the oracle in the loop."""
import os
import sys

import numpy as np
import pytest

from scenario import Scenario, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
sys.path.insert(0, GOLD)
import make_golden_rows as mg  # noqa: E402  (the constants the fixtures were made with)

NAMES = ["beamskip", "calc_range", "lut_reference", "odom", "cloud3d"]


def _load(name):
    return np.load(os.path.join(GOLD, "rows_%s.npz" % name))


def test_row_fixtures_present():
    for n in NAMES:
        assert os.path.exists(os.path.join(GOLD, "rows_%s.npz" % n)), n


# ------------------------------------------------------------------ CPU: the oracle against its own pins
def test_oracle_reproduces_beamskip(orc):
    g = _load("beamskip")
    rec = mg.beamskip_case(orc, Scenario)
    assert np.array_equal(rec["samples"], g["samples"]) and np.array_equal(rec["lut"], g["lut"])
    assert np.array_equal(rec["weights"], g["weights"]) and rec["total"] == float(g["total"])


def test_oracle_reproduces_calc_range(orc):
    g = _load("calc_range")
    rec = mg.calc_range_case(orc, Scenario)
    assert np.array_equal(rec["ox"], g["ox"]) and np.array_equal(rec["ranges"], g["ranges"])
    assert (g["ranges"] < g["max_range"]).sum() > 40  # the fan really hits things


def test_oracle_reproduces_lut_reference(orc):
    g = _load("lut_reference")
    rec = mg.lut_case(orc)
    assert np.array_equal(rec["cells"], g["cells"]) and np.array_equal(rec["lut"], g["lut"])


def test_oracle_reproduces_odom(orc):
    g = _load("odom")
    rec = mg.odom_case(orc)
    for model in range(5):
        assert np.array_equal(rec["moved%d" % model], g["moved%d" % model]), model
        assert int(rec["rng_after%d" % model]) == int(g["rng_after%d" % model])


def test_oracle_reproduces_cloud3d(orc):
    g = _load("cloud3d")
    rec = mg.cloud_case(orc)
    assert np.array_equal(rec["pose_indices"], g["pose_indices"])
    assert np.array_equal(rec["distance_ratios"], g["distance_ratios"])
    for name in ("plain", "gompertz"):
        assert np.array_equal(rec["weights_" + name], g["weights_" + name]), name
        assert float(rec["total_" + name]) == float(g["total_" + name])


# ------------------------------------------------------------------ GPU: the HIP path against the fixtures
@pytest.fixture(scope="module")
def engine():
    import badger_amcl_amd as bpf
    e = bpf.Engine(0)
    yield e
    e.close()


def _map2d(bpf, engine, g, with_lut=True):
    m = bpf.OccupancyMap(engine, 0.05)
    m.setCells(g["cells"].astype(np.int32))
    m.setOrigin(g["origin"])
    if with_lut:
        m.setDistancesLUT(g["lut"], float(g["max_dist"]))
    return m


@pytest.mark.gpu
def test_gpu_hits_beamskip(engine):
    import badger_amcl_amd as bpf
    g = _load("beamskip")
    m = _map2d(bpf, engine, g)
    sc = bpf.PlanarScanner(engine)
    sc.init(60, m)
    k = mg.BEAMSKIP
    sc.setModelLikelihoodFieldProb(0.95, 0.05, 0.2, float(g["max_dist"]), k["do_beamskip"], k["beam_skip_distance"],
                                   k["beam_skip_threshold"], k["beam_skip_error_threshold"])
    sc.setMapFactors(*g["map_factors"])
    sc.setPlanarScannerPose(tuple(g["scanner_pose"]))
    got = g["samples"].copy()
    total = sc.applyModelToSampleSet(bpf.PlanarData(g["ranges"], g["angles"], float(g["range_max"])), got, 1)
    assert rel_err(got[:, 3], g["weights"]).max() <= 1e-9
    assert abs(total - float(g["total"])) <= 1e-9 * float(g["total"])


@pytest.mark.gpu
def test_gpu_hits_calc_range(engine):
    """The device's jumping walk (chessboard-distance grid) against the cell-by-cell walk of the oracle on the ray
    fan: all octants and their borders, starts on and off the map, zero and long max ranges.  Exact."""
    import badger_amcl_amd as bpf
    g = _load("calc_range")
    m = _map2d(bpf, engine, g, with_lut=False)
    got = m.calcRange(g["ox"], g["oy"], g["oa"], g["max_range"])
    assert np.array_equal(got, g["ranges"]), np.flatnonzero(got != g["ranges"])[:10]


@pytest.mark.gpu
def test_gpu_hits_lut_reference(engine):
    import badger_amcl_amd as bpf
    g = _load("lut_reference")
    m = _map2d(bpf, engine, g, with_lut=False)
    m.updateDistancesLUTReference(float(g["max_dist"]))
    assert np.array_equal(m.getDistancesLUT().reshape(-1), g["lut"].reshape(-1))


@pytest.mark.gpu
@pytest.mark.parametrize("model", [0, 1, 2, 3, 4])
def test_gpu_hits_odom(engine, model):
    import badger_amcl_amd as bpf
    g = _load("odom")
    s = g["samples"]
    pf = bpf.ParticleFilter(engine, 10, s.shape[0], 0.0, 0.0, 85.0)
    pf.setRngState(int(g["rng_start"]) ^ (model * 0x1111))
    pf.initWithSamples(s, leaf_count=1)
    od = bpf.Odom(engine)
    od.setModel(model, *g["alpha"])
    od.updateAction(pf, bpf.OdomData(tuple(g["pose"]), tuple(g["delta"]), tuple(g["absolute_motion"])))
    got = pf.getCurrentSet().samples
    want = g["moved%d" % model]
    assert pf.getRngState() == int(g["rng_after%d" % model])        # every rejected Box-Muller attempt accounted for
    assert np.array_equal(got[:, 3], want[:, 3])
    assert np.abs(got[:, :3] - want[:, :3]).max() <= 1e-12     # device log / sin / cos vs glibc: an ulp or two


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["plain", "gompertz"])
def test_gpu_hits_cloud3d(engine, name):
    import badger_amcl_amd as bpf
    g = _load("cloud3d")
    om = bpf.OctoMap(engine, 0.05)
    om.setDistancesLUT(g["pose_indices"], g["distance_ratios"], g["min_cells"], g["max_cells"], float(g["max_dist"]))
    sc = bpf.PointCloudScanner(engine)
    sc.init(128, om)
    if name == "plain":
        sc.setPointCloudModel(0.5, 0.05, 0.1)
    else:
        z = mg.GOMPERTZ_3D
        sc.setPointCloudModelGompertz(0.5, 0.5, 0.1, z["gompertz_a"], z["gompertz_b"], z["gompertz_c"],
                                      z["input_shift"], z["input_scale"], z["output_shift"])
    sc.setMapFactors(0.95, 0.95, 0.3)
    sc.setPointCloudScannerToFootprintTF(tuple(g["tf_xyz"]), tuple(g["tf_quat"]))
    got = g["samples"].copy()
    total = sc.applyModelToSampleSet(bpf.PointCloudData(g["points"]), got)
    want = g["weights_" + name]
    # a point within rounding of a voxel face may resolve differently (corrected reciprocal vs true division)
    assert (rel_err(got[:, 3], want) > 1e-9).sum() <= 1
    assert abs(total - float(g["total_" + name])) <= 1e-7 * float(g["total_" + name])
