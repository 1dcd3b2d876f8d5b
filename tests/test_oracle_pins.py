"""Pins the CPU oracle against every known answer the reference's own tests hold
for this path (test/test_badger_amcl.cpp).  EXPECT_DOUBLE_EQ there is a 4-ULP
window; the same window is used here."""
import ctypes
import math

import numpy as np
import pytest


def ulps(a, b):
    ia = np.array([a], dtype=np.float64).view(np.int64)[0]
    ib = np.array([b], dtype=np.float64).view(np.int64)[0]
    return abs(int(ia) - int(ib))


def test_drand48_matches_glibc(orc):
    """R16: the clone against the libc this process runs on, seeded and unseeded."""
    libc = ctypes.CDLL("libc.so.6")
    libc.drand48.restype = ctypes.c_double
    libc.srand48.argtypes = [ctypes.c_long]
    for seed in (0, 1, 42, 123456789, 2**31 - 1):
        libc.srand48(seed)
        r = orc.Rng(seed)
        for _ in range(64):
            assert r.drand48() == libc.drand48()


def test_pdf_gaussian_known_answer(orc):
    """test_badger_amcl.cpp:29-49: identity covariance, mean (1,1,1), unseeded stream.
    sample[i] = 1 + draw(1) (the rotation is the identity)."""
    r = orc.Rng(None)
    want = [0.26562654174915334, 0.97172090090793528, -1.5856194295513539,
            1.6262083813236745, 1.1142314205031041, 0.37407538872488655]
    got = [1.0 + 1.0 * r.gaussian(1.0) for _ in range(6)]
    for g, w in zip(got, want):
        assert ulps(g, w) <= 4, (g, w)


def test_kdtree_known_answer(orc):
    """test_badger_amcl.cpp:51-82"""
    t = orc.KDTree()
    assert t.leaf_count() == 0
    pose = (1, 1, 1)
    t.insert_pose(pose, 0.0)
    assert t.leaf_count() == 1
    t.clear()
    assert t.leaf_count() == 0
    t.insert_pose(pose, 0.0)
    assert t.get_cluster(pose) == -1
    t.cluster()
    assert t.get_cluster(pose) == 0
    pose2, pose3 = (0, 1, 1), (3, 0, 0)
    t.insert_pose(pose2, 0.0)
    t.insert_pose(pose3, 0.0)
    t.cluster()
    assert t.get_cluster(pose) == 0
    assert t.get_cluster(pose2) == 1
    assert t.get_cluster(pose3) == 2
    assert t.leaf_count() == 2
    pose4 = (0.5, 1, 1)
    t.insert_pose(pose4, 0.0)
    t.cluster()
    assert t.get_cluster(pose) == 0
    assert t.get_cluster(pose2) == 0
    assert t.get_cluster(pose3) == 1
    assert t.get_cluster(pose4) == 0
    assert t.leaf_count() == 2


def test_octomap_conversions_known_answer(orc):
    """test_badger_amcl.cpp:84-111"""
    lut = orc.OctoMapLUT((0, 0, 0), (0, 0, 0), 0.05, 0.3, np.zeros(1, np.uint32), np.full(1, 255, np.uint8))
    w = orc.map3d_map_to_world(0.05, (1, 2, 0))
    assert ulps(w[0], .05) <= 4 and ulps(w[1], .1) <= 4
    c = lut.world_to_map((.05, .1, 0.0))
    assert (c[0], c[1]) == (1, 2)
    w = orc.map3d_map_to_world(0.05, (3, 5, -1))
    for got, want in zip(w, (.15, .25, -.05)):
        assert ulps(got, want) <= 4
    assert tuple(lut.world_to_map((.15, .25, -.05))) == (3, 5, -1)


def test_occupancy_map_conversions_known_answer(orc):
    """test_badger_amcl.cpp:113-129: default (0,0) origin, size 0x0"""
    m = orc.OccupancyMap(np.zeros((0, 0), np.int32), 0.05)
    x, y = m.map_to_world(1, 2)
    assert ulps(x, .05) <= 4 and ulps(y, .1) <= 4
    assert m.world_to_map(.05, .1) == (1, 2)


def _test_map(orc):
    res = 0.05
    sx, sy = 100, 150
    cells = np.full((sy, sx), -1, np.int32)
    for x in range(sx):
        for y in range(sy):
            if x == 1 and 2 < y < 12:
                cells[y, x] = 0
            elif 4 < x < 14 and y in (10, 15):
                cells[y, x] = 1
    origin = (sx // 2 * res, sy // 2 * res)
    return orc.OccupancyMap(cells, res, origin)


def test_occupancy_map_distances_known_answer(orc):
    """test_badger_amcl.cpp:131-171"""
    m = _test_map(orc)
    assert m.is_valid(0, 0)
    assert not m.is_valid(-1, 5)
    assert m.is_valid(99, 149)
    assert not m.is_valid(100, 150)
    assert not m.is_valid(149, 99)
    m.update_distances_lut(0.3)
    assert m.cells[0, 0] == -1 and m.cells[3, 1] == 0 and m.cells[10, 5] == 1
    assert m.calc_range(0, 0, 0, 0) == 0.0
    assert ulps(m.calc_range(0.05, 0, 1.5708, 0.5), 0.15) <= 4


def test_brushfire_properties(orc):
    """The reference asserts no LUT value (PARITY UNPINNED for the brushfire); check what
    must hold by construction: zeros exactly on occupied cells, values are
    sqrt(a^2+b^2)*res for integer a,b within the radius, never below the exact EDT."""
    m = _test_map(orc)
    lut = m.update_distances_lut(0.3)
    occ = np.argwhere(m.cells == 1)
    assert np.all(lut[m.cells == 1] == 0.0)
    assert np.all(lut[m.cells != 1] > 0.0)
    radius = int(math.floor(0.3 / 0.05))
    allowed = {np.float32(math.sqrt(a * a + b * b) * 0.05) for a in range(radius + 2) for b in range(radius + 2)
               if math.sqrt(a * a + b * b) <= radius}
    allowed.add(np.float32(0.3))
    assert set(np.unique(lut).tolist()) <= {float(v) for v in allowed}
    ys, xs = np.mgrid[0:150, 0:100]
    d2 = np.min((ys[..., None] - occ[:, 0]) ** 2 + (xs[..., None] - occ[:, 1]) ** 2, axis=-1)
    exact = np.sqrt(d2) * 0.05
    near = exact <= radius * 0.05
    assert np.all(lut[near] >= np.float32(exact[near]) - 1e-7)
    # the brushfire is exact on this simple layout for the vast majority of cells
    assert np.mean(np.abs(lut[near] - exact[near].astype(np.float32)) < 1e-6) > 0.95


def test_brushfire_heap_matches_libstdcxx(orc, tmp_path):
    """The brushfire's tie order comes from libstdc++'s binary heap (third party:
    GCC libstdc++ bits/stl_heap.h).  Build the same map with a small C++ program
    that uses std::priority_queue with the reference's comparator shape and compare
    LUTs bit for bit."""
    import subprocess
    src = tmp_path / "pq.cpp"
    src.write_text(r'''
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <vector>
static std::vector<float> lut; static int SX, SY;
struct Cell { int i, j, si, sj;
  bool operator<(const Cell& b) const { return lut[i + j * SX] > lut[b.i + b.j * SX]; } };
int main(int argc, char** argv) {
  SX = atoi(argv[1]); SY = atoi(argv[2]); double res = atof(argv[3]), md = atof(argv[4]);
  std::vector<int> cells(SX * SY);
  FILE* f = fopen(argv[5], "rb"); if (fread(cells.data(), 4, cells.size(), f) != cells.size()) return 2; fclose(f);
  int radius = (int)std::floor(md / res);
  lut.assign(SX * SY, 0.f); std::vector<bool> marked(SX * SY, false);
  std::priority_queue<Cell> q;
  for (int i = 0; i < SX; i++) for (int j = 0; j < SY; j++) {
    if (cells[i + j * SX] == 1) { lut[i + j * SX] = 0.f; marked[i + j * SX] = true; q.push(Cell{i, j, i, j}); }
    else lut[i + j * SX] = (float)md; }
  auto visit = [&](int i, int j, const Cell& c) {
    if (marked[i + j * SX]) return;
    int di = std::abs(i - c.si), dj = std::abs(j - c.sj);
    double d = std::sqrt((double)(di * di + dj * dj));
    if (d <= radius) { lut[i + j * SX] = (float)(d * res); q.push(Cell{i, j, c.si, c.sj}); marked[i + j * SX] = true; } };
  while (!q.empty()) { Cell c = q.top();
    if (c.i > 0) visit(c.i - 1, c.j, c);
    if (c.j > 0) visit(c.i, c.j - 1, c);
    if (c.i < SX - 1) visit(c.i + 1, c.j, c);
    if (c.j < SY - 1) visit(c.i, c.j + 1, c);
    q.pop(); }
  f = fopen(argv[6], "wb"); fwrite(lut.data(), 4, lut.size(), f); fclose(f); return 0; }
''')
    exe = tmp_path / "pq"
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-o", str(exe), str(src)])
    rng = np.random.default_rng(5)
    cells = np.full((90, 130), -1, np.int32)
    cells[rng.random(cells.shape) < 0.02] = 1
    cells[40, 10:100] = 1
    cells[rng.random(cells.shape) < 0.01] = 0
    m = orc.OccupancyMap(cells, 0.05)
    mine = m.update_distances_lut(0.5)
    (tmp_path / "cells.bin").write_bytes(cells.tobytes())
    subprocess.check_call([str(exe), "130", "90", "0.05", "0.5", str(tmp_path / "cells.bin"),
                           str(tmp_path / "lut.bin")])
    theirs = np.frombuffer((tmp_path / "lut.bin").read_bytes(), dtype=np.float32).reshape(90, 130)
    assert np.array_equal(mine, theirs)


def test_resample_limit_shape(orc):
    """particle_filter.cpp:475-502: k<=1 -> max; clamped to [min, max]; non-decreasing in k."""
    pf = orc.ParticleFilter(100, 5000)
    assert pf.resample_limit(0) == 5000 and pf.resample_limit(1) == 5000
    vals = [pf.resample_limit(k) for k in range(2, 400)]
    assert all(b >= a for a, b in zip(vals, vals[1:]))
    assert min(vals) >= 100 and max(vals) <= 5000
    k = 50
    x = 1 - 2 / (9 * (k - 1.0)) + math.sqrt(2 / (9 * (k - 1.0))) * 3
    assert pf.resample_limit(k) == int(math.ceil((k - 1) / (2 * 0.01) * x * x * x))
