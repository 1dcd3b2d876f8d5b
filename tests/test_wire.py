"""Wire-format shaping: the bpf_wire_* host functions against numpy
restatements of the node code they replace (file:line cited per test).  No GPU is touched."""
import math

import numpy as np

from badger_amcl_amd import wire


def test_laserscan_to_planar_matches_node2d():
    """node_2d.cpp:531-560: range limits narrowed through float, short readings -> max range,
    bearing = angle_min + i * increment (double)."""
    rng = np.random.default_rng(0)
    r = rng.uniform(0.0, 35.0, 1081).astype(np.float32)
    r[::50] = 0.05
    r[7] = np.float32(0.1)      # == range_min: "<=" sends it to max
    r[9] = np.nan               # NaN compares false: passed through
    r[11] = np.inf
    amin, ainc = -2.356194490192345, 0.004363323129985824
    for smin, smax in [(-1.0, -1.0), (0.25, 25.0), (0.05, 60.0)]:
        ro, ao, rmax = wire.laserscan_to_planar(r, np.float32(0.1), np.float32(30.0), amin, ainc, smin, smax)
        want_max = float(min(np.float32(30.0), np.float32(smax))) if smax > 0 else float(np.float32(30.0))
        want_min = float(max(np.float32(0.1), np.float32(smin))) if smin > 0 else float(np.float32(0.1))
        assert rmax == want_max
        want = r.astype(np.float64)
        want[r.astype(np.float64) <= want_min] = want_max
        assert np.array_equal(ro, want, equal_nan=True)
        assert np.array_equal(ao, amin + np.arange(1081) * ainc)


def test_scan_angle_stats_upright_and_upside_down():
    """node_2d.cpp:497-529: an upright scanner yawed by 0.3 shifts angle_min; one rolled by pi
    (mounted upside-down) mirrors the sweep: angle_min -> -angle_min, increment -> -increment."""
    amin, ainc = -2.0, 0.005
    a, b = wire.scan_angle_stats(amin, ainc, (0.0, 0.0, 0.0, 1.0))
    assert abs(a - amin) < 1e-15 and abs(b - ainc) < 1e-15
    a, b = wire.scan_angle_stats(amin, ainc, (0.0, 0.0, math.sin(0.15), math.cos(0.15)))
    assert abs(a - (amin + 0.3)) < 1e-15 and abs(b - ainc) < 1e-15
    a, b = wire.scan_angle_stats(amin, ainc, (1.0, 0.0, 0.0, 0.0))  # roll = pi
    assert abs(a + amin) < 1e-15 and abs(b + ainc) < 1e-15
    # wrap of the increment into [-pi, pi): first bearing near +pi
    a, b = wire.scan_angle_stats(math.pi - 0.001, 0.004, (0.0, 0.0, 0.0, 1.0))
    assert abs(b - 0.004) < 1e-12


def test_laserscan_empty():
    ro, ao, rmax = wire.laserscan_to_planar(np.zeros(0, np.float32), 0.1, 30.0, 0.0, 0.01)
    assert ro.size == 0 and ao.size == 0 and rmax == float(np.float32(30.0))


def test_occupancy_grid_to_cells_matches_convert_map():
    """node_2d.cpp:265-295: 0 -> free(-1), 100 -> occupied(+1), anything else unknown(0); integer
    up-scaling replicates cells; origin is the map centre, narrowed to float."""
    rng = np.random.default_rng(1)
    w, h = 37, 23
    data = rng.choice(np.array([0, 100, -1, 50, 99], dtype=np.int8), size=w * h)
    for f in (1, 3):
        cells, origin, res = wire.occupancy_grid_to_cells(data, w, h, 0.05, -3.2, 1.7, f)
        g = data.reshape(h, w)
        tri = np.where(g == 0, -1, np.where(g == 100, 1, 0)).astype(np.int32)
        want = np.repeat(np.repeat(tri, f, axis=0), f, axis=1)
        assert np.array_equal(cells, want)
        assert res == 0.05 / f
        assert origin[0] == np.float32(-3.2 + ((w * f) // 2) * (0.05 / f))
        assert origin[1] == np.float32(1.7 + ((h * f) // 2) * (0.05 / f))


def test_decimate_cloud_matches_node3d():
    """node_3d.cpp:467-480: step = max((n - 1) / (max_beams - 1), 1), points 0, step, 2*step, ..."""
    rng = np.random.default_rng(2)
    for n, mb in [(65536, 1024), (1000, 4096), (1, 30), (0, 30), (2049, 2), (31, 30)]:
        p = rng.normal(size=(n, 3)).astype(np.float32)
        got = wire.decimate_cloud(p, mb)
        step = max((n - 1) // (mb - 1), 1) if n > 0 else 1
        assert np.array_equal(got, p[::step])


def test_samples_to_pose_array_matches_publish_particle_cloud():
    """node.cpp:335-357: position (x, y, 0), orientation setRPY(0, 0, theta)."""
    rng = np.random.default_rng(3)
    s = np.zeros((500, 4))
    s[:, :2] = rng.uniform(-50, 50, (500, 2))
    s[:, 2] = rng.uniform(-math.pi, math.pi, 500)
    out = wire.samples_to_pose_array(s)
    assert np.array_equal(out[:, 0], s[:, 0]) and np.array_equal(out[:, 1], s[:, 1])
    assert np.all(out[:, 2:5] == 0.0)
    # libm sin / cos of theta/2 (numpy uses its own SIMD kernels: allow 1 ulp)
    assert np.allclose(out[:, 5], np.sin(s[:, 2] / 2), rtol=0, atol=2.3e-16)
    assert np.allclose(out[:, 6], np.cos(s[:, 2] / 2), rtol=0, atol=2.3e-16)
    # round trip: yaw recovered from the quaternion
    yaw = 2 * np.arctan2(out[:, 5], out[:, 6])
    assert np.allclose(yaw, s[:, 2], atol=1e-15)


# ---- the same functions against the oracle's restatement of the node code (oracle/amcl_oracle.c, orc_wire_*)
def test_wire_functions_match_the_oracle():
    """node_2d.cpp:265-295,497-560, node_3d.cpp:467-480, node.cpp:335-357: product (bpf_wire_*) vs oracle, bit for
    bit, on message fields with the awkward values: readings at and below range_min, NaN / inf readings, sensor
    limits that bind and that do not, an upside-down and a yawed mounting, scale-up factors, odd cloud sizes."""
    from oracle import pyoracle as orc
    rng = np.random.default_rng(5)
    r = rng.uniform(0.0, 35.0, 1081).astype(np.float32)
    r[::40] = 0.02
    r[3] = np.float32(0.1)
    r[5], r[6] = np.nan, np.inf
    amin, ainc = -2.356194490192345, 0.004363323129985824
    for smin, smax in [(-1.0, -1.0), (0.25, 25.0), (0.05, 60.0), (0.1, 30.0)]:
        got = wire.laserscan_to_planar(r, np.float32(0.1), np.float32(30.0), amin, ainc, smin, smax)
        want = orc.wire_laserscan_to_planar(r, np.float32(0.1), np.float32(30.0), amin, ainc, smin, smax)
        assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    for q in [(0.0, 0.0, 0.0, 1.0), (0.0, 0.0, math.sin(0.15), math.cos(0.15)), (1.0, 0.0, 0.0, 0.0),
              (math.sin(0.05), 0.0, 0.0, math.cos(0.05)), (0.5, 0.5, 0.5, 0.5)]:
        for a0, da in [(-2.0, 0.005), (math.pi - 0.001, 0.004), (0.3, -0.01)]:
            assert wire.scan_angle_stats(a0, da, q) == orc.wire_scan_angle_stats(a0, da, q)
    w, h = 41, 29
    data = rng.choice(np.array([0, 100, -1, 50, 99, 1], dtype=np.int8), size=w * h)
    for f in (1, 2, 3):
        cells, origin, res = wire.occupancy_grid_to_cells(data, w, h, 0.05, -3.2, 1.7, f)
        ocells, oorigin, ores = orc.wire_convert_map(data, w, h, 0.05, -3.2, 1.7, f)
        assert np.array_equal(cells, ocells) and res == ores
        assert origin[0] == oorigin[0] and origin[1] == oorigin[1]
    for n, mb in [(65536, 128), (1000, 999), (10, 64), (129, 128), (2, 2)]:
        pts = rng.normal(size=(n, 3)).astype(np.float32)
        assert np.array_equal(wire.decimate_cloud(pts, mb), orc.wire_decimate_cloud(pts, mb))
    s = rng.normal(size=(500, 4))
    s[:, 2] = rng.uniform(-7.0, 7.0, 500)
    got, want = wire.samples_to_pose_array(s), orc.wire_pose_array(s)
    assert np.array_equal(got[:, :5], want[:, :5])
    assert np.allclose(got[:, 5:], want[:, 5:], rtol=0, atol=3e-16)  # sin / cos vs sincos of the host libm: an ulp
