"""The C++ host side (include/badger_amcl_amd/adapter.hpp) above the C-ABI: compile a small
program that follows the reference's call order and compare what it produces with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from scenario import Scenario, rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    exe = tmp_path / "adapter_smoke"
    libdir = os.path.join(ROOT, "badger_amcl_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_smoke.cpp"), "-o", str(exe),
                           "-L", libdir, "-lbadger_pf_hip", "-Wl,-rpath," + libdir])
    return exe


def test_adapter_compiles_against_the_c_abi(tmp_path):
    """CPU: the adapter header is valid C++17 and links against the shared library."""
    from badger_amcl_amd import build
    build.build()
    _compile(tmp_path)


@pytest.mark.gpu
def test_adapter_run_matches_oracle(tmp_path, orc):
    exe = _compile(tmp_path)
    sc = Scenario(orc, size=200, n=1500, beams=91, cloud="converged")
    paths = {}
    for name, arr in (("cells", sc.cells.astype(np.int32)), ("lut", sc.lut.astype(np.float32)),
                      ("samples", sc.samples), ("ranges", sc.ranges), ("angles", sc.angles)):
        paths[name] = str(tmp_path / (name + ".bin"))
        np.ascontiguousarray(arr).tofile(paths[name])
    out_w, out_r = str(tmp_path / "w.bin"), str(tmp_path / "r.bin")
    res = subprocess.run([str(exe), paths["cells"], paths["lut"], paths["samples"], paths["ranges"], paths["angles"],
                          "200", out_w, out_r], capture_output=True, text=True, check=True)
    m, leaf, conv = (int(v) for v in res.stdout.split())
    opf = orc.ParticleFilter(100, 1500, 0.0, 0.0, 85.0, seed=42)
    opf.set_samples(sc.samples)
    p = sc.oracle_planar(91, "lf")
    opf.update_sensor(lambda s, c: sc.oracle_apply(p, s, c))
    got_w = np.fromfile(out_w, dtype=np.float64).reshape(-1, 4)
    assert rel_err(got_w[:, 3], opf.samples[:1500, 3]).max() <= 1e-9
    out = opf.update_resample()
    got_r = np.fromfile(out_r, dtype=np.float64).reshape(-1, 4)
    assert (m, leaf, conv) == (out.sample_count, out.leaf_count, out.converged)
    assert np.array_equal(got_r[:, :3], opf.samples[:m, :3])


@pytest.mark.gpu
def test_adapter_default_lut_is_the_reference_brushfire(tmp_path, orc):
    """No LUT handed over: the reference-named call sequence (setModelLikelihoodField -> map->updateDistancesLUT,
    planar_scanner.cpp:74, occupancy_map.cpp:138-252) must leave the oracle's brushfire values behind, bit for bit,
    and the weights that follow from them."""
    exe = _compile(tmp_path)
    sc = Scenario(orc, size=200, n=1500, beams=91, cloud="mixture")
    rng = np.random.default_rng(9)
    sc.cells[rng.random(sc.cells.shape) < 0.002] = 1   # scattered obstacles: many equidistant ties
    sc.omap = orc.OccupancyMap(sc.cells, sc.res, sc.origin)
    sc.lut = sc.omap.update_distances_lut(2.0)
    paths = {}
    for name, arr in (("cells", sc.cells.astype(np.int32)), ("samples", sc.samples), ("ranges", sc.ranges),
                      ("angles", sc.angles)):
        paths[name] = str(tmp_path / (name + ".bin"))
        np.ascontiguousarray(arr).tofile(paths[name])
    out_w, out_r, out_l = str(tmp_path / "w.bin"), str(tmp_path / "r.bin"), str(tmp_path / "l.bin")
    subprocess.run([str(exe), paths["cells"], "-", paths["samples"], paths["ranges"], paths["angles"], "200", out_w,
                    out_r, out_l], capture_output=True, text=True, check=True)
    got_l = np.fromfile(out_l, dtype=np.float32)
    assert np.array_equal(got_l, np.asarray(sc.lut, dtype=np.float32).reshape(-1))
    opf = orc.ParticleFilter(100, 1500, 0.0, 0.0, 85.0, seed=42)
    opf.set_samples(sc.samples)
    p = sc.oracle_planar(91, "lf")
    opf.update_sensor(lambda s, c: sc.oracle_apply(p, s, c))
    got_w = np.fromfile(out_w, dtype=np.float64).reshape(-1, 4)
    assert rel_err(got_w[:, 3], opf.samples[:1500, 3]).max() <= 1e-9


@pytest.mark.gpu
def test_adapter_scores_a_pinned_host_set_in_place(tmp_path, orc):
    """INTEGRATION Option 1 through the C++ adapter: the host-resident std::vector<PFSample> of the reference, pinned
    once (PinnedSamples), handed to applyModelToSampleSet -- one scoring launch reads and writes the records in place."""
    exe = _compile(tmp_path)
    sc = Scenario(orc, size=200, n=20000, beams=91, cloud="mixture")
    paths = {}
    for name, arr in (("cells", sc.cells.astype(np.int32)), ("lut", sc.lut.astype(np.float32)),
                      ("samples", sc.samples), ("ranges", sc.ranges), ("angles", sc.angles)):
        paths[name] = str(tmp_path / (name + ".bin"))
        np.ascontiguousarray(arr).tofile(paths[name])
    out_w, out_r, out_l, out_h = (str(tmp_path / f) for f in ("w.bin", "r.bin", "l.bin", "h.bin"))
    res = subprocess.run([str(exe), paths["cells"], paths["lut"], paths["samples"], paths["ranges"], paths["angles"], "200",
                          out_w, out_r, out_l, out_h], capture_output=True, text=True, check=True)
    assert "chunks -1 pinned 1" in res.stderr, res.stderr
    want = sc.samples.copy()
    want_total = sc.oracle_apply(sc.oracle_planar(91, "lf"), want)
    got = np.fromfile(out_h, dtype=np.float64).reshape(-1, 4)
    assert np.array_equal(got[:, :3], want[:, :3])
    assert (rel_err(got[:, 3], want[:, 3]) > 1e-9).sum() <= 1
    total = float(res.stderr.split("total")[1].split()[0])
    assert abs(total - want_total) <= 1e-9 * abs(want_total)
