"""Recovery branch of the resamplers (w_diff > 0), CPU side: the oracle's C restatement of
resampleMultinomial / resampleSystematic with random_pose_fn_ = Node::randomFreeSpacePose against a second,
independent restatement in plain Python.  PARITY UNPINNED: the reference holds no test for this branch."""
import math

import numpy as np
import pytest

from badger_amcl_amd import synth

A, C_, MASK = 0x5DEECE66D, 0xB, (1 << 48) - 1


class Rng:
    def __init__(self, s):
        self.s = s

    def drand48(self):
        self.s = (A * self.s + C_) & MASK
        return self.s / float(1 << 48)


def free_cells(cells, lut, radius):
    """node_2d.cpp:317-337: i outer, j inner, FREE and distance > radius."""
    sy, sx = cells.shape
    out = []
    for i in range(sx):
        for j in range(sy):
            if cells[j, i] == -1 and float(lut[j, i]) > radius:
                out.append((i, j))
    return out


def random_pose(rng, fs, sx, sy, origin, res):
    """node.cpp:823-845 + occupancy_map.cpp:75-88."""
    idx = int(rng.drand48() * len(fs))
    i, j = fs[idx]
    x = float(origin[0]) + (i - sx // 2) * res
    y = float(origin[1]) + (j - sy // 2) * res
    return [x, y, rng.drand48() * 2 * math.pi - math.pi]


@pytest.mark.parametrize("resampler", [0, 1])
def test_oracle_recovery_matches_python_restatement(orc, resampler):
    size, res, n = 60, 0.05, 400
    cells, origin = synth.make_map(size, res)
    omap = orc.OccupancyMap(cells, res, origin)
    lut = omap.update_distances_lut(1.0)
    radius = 0.3
    fs = free_cells(cells, lut, radius)
    s = synth.spread_cloud(n, size, res, seed=3, margin=0.2)
    s[:, 3] = np.random.default_rng(1).uniform(0.5, 1.5, n)
    s[:, 3] /= s[:, 3].sum()
    opf = orc.ParticleFilter(50, n, 0.001, 0.1, 85.0, seed=77)
    opf.set_resample_model(resampler)
    opf.set_samples(s)
    assert opf.set_random_pose_source(omap, radius) == len(fs)
    opf.pf.w_slow, opf.pf.w_fast = 1.0, 0.8  # w_diff = 0.2
    rng0 = int(opf.pf.rng)
    leaf0 = opf.leaf_count
    out = opf.update_resample()
    assert out.status == 0 and abs(out.w_diff - 0.2) < 1e-15
    # --- the same in Python
    r = Rng(rng0)
    w_diff = 1.0 - 0.8 / 1.0
    c = [0.0]
    for w in s[:, 3]:
        c.append(c[-1] + float(w))
    t = orc.KDTree()
    want = []

    def find(u):
        for i in range(n):
            if c[i] <= u < c[i + 1]:
                return i
        raise AssertionError("CDF miss")

    if resampler == 0:
        while len(want) < n:
            if r.drand48() < w_diff:
                pose = random_pose(r, fs, size, size, origin, res)
            else:
                pose = [float(v) for v in s[find(r.drand48()), :3]]
            want.append(pose)
            t.insert_pose(pose, 1.0)
            if len(want) > opf.resample_limit(t.leaf_count()):
                break
    else:
        count = opf.resample_limit(leaf0)
        count = int(count * (1.0 + w_diff))
        count = min(count, n)
        n_random = int(w_diff * count)
        n_sys = count - n_random
        start = r.drand48()
        delta = 1.0 / n_sys
        for _ in range(n_random):
            want.append(random_pose(r, fs, size, size, origin, res))
        target = start
        for _ in range(n_sys):
            want.append([float(v) for v in s[find(target), :3]])
            target += delta
            if target > 1.0:
                target -= 1.0
        for pose in want:
            t.insert_pose(pose, 1.0)
    M = len(want)
    assert out.sample_count == M and out.leaf_count == t.leaf_count()
    assert np.array_equal(opf.samples[:M, :3], np.array(want))
    assert int(opf.pf.rng) == r.s
    assert opf.pf.w_slow == 0.0 and opf.pf.w_fast == 0.0
    assert (opf.last_idx < 0).sum() > 0
