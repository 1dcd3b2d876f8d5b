"""Seeded synthetic inputs for the hot path: map, scan, particle clouds.

Pure numpy input generation -- nothing here scores, normalises or resamples.  The same
arrays are handed to the HIP engine and, in tests / the cpu_baseline leg, to the oracle.
"""
import math

import numpy as np


def make_map(size, resolution=0.05, seed=0):
    """size x size tri-state grid: border walls, a lattice of wall segments, an unknown strip.
    Returns (cells[int32, size_y x size_x], origin_xy) with the map spanning [0, size*res)^2,
    i.e. origin = (size/2)*res narrowed to float32 like node_2d.cpp:275-277."""
    s = int(size)
    ys, xs = np.mgrid[0:s, 0:s]
    cells = np.full((s, s), -1, dtype=np.int32)
    wall = ((xs % 80 == 0) & (ys % 40 < 30)) | ((ys % 100 == 0) & (xs % 50 < 35))
    wall |= (xs == 0) | (ys == 0) | (xs == s - 1) | (ys == s - 1)
    # keep the neighbourhood of the map centre free so the true pose is in free space
    c = s // 2
    wall[max(c - 6, 1):c + 7, max(c - 6, 1):c + 7] = False
    cells[wall] = 1
    strip = (xs >= s // 3) & (xs < s // 3 + 3) & (ys > s // 8) & (ys < s // 8 + s // 5) & ~wall
    cells[strip] = 0
    origin = (np.float32((s // 2) * resolution), np.float32((s // 2) * resolution))
    return cells, origin


def true_pose(size, resolution=0.05):
    c = (size // 2) * resolution
    return np.array([c + 0.1, c + 0.0, 0.3], dtype=np.float64)


def _world_to_cell(v, origin, res, half):
    return np.floor((v - float(origin)) / res + 0.5).astype(np.int64) + half


def cast_scan(cells, origin, resolution, pose, n_beams, range_max=30.0, fov=1.5 * math.pi, noise=0.01, seed=1,
              frac_max=0.0, frac_nan=0.0):
    """Ranges seen from `pose` by marching each ray in quarter-cell steps to the first
    non-free or off-map cell (input generation only; not the reference's Bresenham)."""
    rng = np.random.default_rng(seed)
    sy, sx = cells.shape
    angles = np.linspace(-fov / 2, fov / 2, n_beams)
    step = resolution * 0.25
    n_steps = int(range_max / step) + 1
    ranges = np.full(n_beams, range_max, dtype=np.float64)
    alive = np.ones(n_beams, dtype=bool)
    ca, sa = np.cos(pose[2] + angles), np.sin(pose[2] + angles)
    for k in range(1, n_steps):
        if not alive.any():
            break
        d = k * step
        ix = _world_to_cell(pose[0] + d * ca[alive], origin[0], resolution, sx // 2)
        iy = _world_to_cell(pose[1] + d * sa[alive], origin[1], resolution, sy // 2)
        inside = (ix >= 0) & (ix < sx) & (iy >= 0) & (iy < sy)
        hit = ~inside
        hit[inside] = cells[iy[inside], ix[inside]] != -1
        idx = np.flatnonzero(alive)[hit]
        ranges[idx] = d
        alive[idx] = False
    ranges = ranges + rng.normal(0.0, noise, n_beams)
    ranges = np.clip(ranges, 0.05, np.nextafter(range_max, 0.0))
    ranges = np.float32(ranges).astype(np.float64)  # LaserScan ranges are float32 (node_2d.cpp:549-559)
    ranges = np.minimum(ranges, np.nextafter(range_max, 0.0))
    if frac_max > 0:
        ranges[rng.random(n_beams) < frac_max] = range_max
    if frac_nan > 0:
        ranges[rng.random(n_beams) < frac_nan] = np.nan
    return ranges, angles.astype(np.float64)


def converged_cloud(n, pose, seed=42, sigma=(0.3, 0.3, 0.1)):
    rng = np.random.default_rng(seed)
    s = np.zeros((n, 4), dtype=np.float64)
    s[:, 0] = pose[0] + rng.normal(0, sigma[0], n)
    s[:, 1] = pose[1] + rng.normal(0, sigma[1], n)
    s[:, 2] = pose[2] + rng.normal(0, sigma[2], n)
    s[:, 3] = 1.0 / n
    return s


def spread_cloud(n, size, resolution=0.05, seed=43, margin=-1.0):
    """Uniform over the map (margin < 0 pushes some particles off the map edge)."""
    rng = np.random.default_rng(seed)
    extent = size * resolution
    s = np.zeros((n, 4), dtype=np.float64)
    s[:, 0] = rng.uniform(margin, extent - margin, n)
    s[:, 1] = rng.uniform(margin, extent - margin, n)
    s[:, 2] = rng.uniform(-math.pi, math.pi, n)
    s[:, 3] = 1.0 / n
    return s


# model parameter sets --------------------------------------------------------------
LF_DEFAULTS = dict(z_hit=0.95, z_rand=0.05, sigma_hit=0.2)                       # node_2d.cpp:53-58
BEAM_DEFAULTS = dict(z_hit=0.95, z_short=0.1, z_max=0.05, z_rand=0.05, sigma_hit=0.2, lambda_short=0.1)
GOMPERTZ_LAUNCH = dict(z_hit=0.5, z_rand=0.5, sigma_hit=0.05, gompertz_a=0.941, gompertz_b=5.0, gompertz_c=3.0,
                       input_shift=-0.97, input_scale=2.0, output_shift=0.25)    # badger_amcl_2d.launch:69-123
MAP_FACTORS = (0.95, 0.95, 0.3)                                                   # badger_amcl_2d.launch:125-129
SCANNER_POSE = (0.1, 0.0, 0.0)


# 3-D ----------------------------------------------------------------------------------
def box_room_voxels(lo=(-40, -30, -2), hi=(40, 30, 20), pillar=True):
    """Occupied voxel list (i, j, k) of a box room: floor, ceiling, four walls, optionally a pillar."""
    occ = []
    for i in range(lo[0], hi[0] + 1):
        for j in range(lo[1], hi[1] + 1):
            occ.append((i, j, lo[2]))
            occ.append((i, j, hi[2]))
    for k in range(lo[2], hi[2] + 1):
        for i in range(lo[0], hi[0] + 1):
            occ.append((i, lo[1], k))
            occ.append((i, hi[1], k))
        for j in range(lo[1], hi[1] + 1):
            occ.append((lo[0], j, k))
            occ.append((hi[0], j, k))
        if pillar:
            for i in range(8, 12):
                for j in range(-3, 1):
                    occ.append((i, j, k))
    return np.unique(np.array(occ, dtype=np.int32), axis=0)


def sphere_cloud(rows, cols, pose_xyz, occupied, resolution, max_range=10.0, seed=3, noise=0.01):
    """rows x cols spherical-grid cloud in the SCANNER frame: each ray is marched to the first
    occupied voxel of `occupied` (input generation only)."""
    rng = np.random.default_rng(seed)
    occ = {tuple(v) for v in occupied.tolist()}
    el = np.linspace(-0.4, 0.4, rows)
    az = np.linspace(-np.pi, np.pi, cols, endpoint=False)
    pts = []
    step = resolution * 0.5
    for e in el:
        for a in az:
            d = np.array([np.cos(e) * np.cos(a), np.cos(e) * np.sin(a), np.sin(e)])
            r = step
            hit = None
            while r < max_range:
                p = np.asarray(pose_xyz) + r * d
                if tuple(np.floor(p / resolution + 0.5).astype(int)) in occ:
                    hit = r
                    break
                r += step
            if hit is not None:
                pts.append(d * (hit + rng.normal(0, noise)))
    return np.asarray(pts, dtype=np.float32)


def box_room_lut(resolution=0.05, max_dist=0.3, lo=(-40, -30, -2), hi=(40, 30, 20), pad=3):
    """OctoMap-style two-level uint8 distance LUT of the box room, built with an exact Euclidean
    distance transform (bench input only; the reference builds its LUT with a FIFO brushfire from
    an octree, which is outside this engine's scope).  Returns (pose_indices, distance_ratios,
    min_cells, max_cells) in the layout of octomap.cpp:315-355: column 0 is the shared all-255
    column, every (x, y) column that holds a value < 255 gets its own run of num_z bytes."""
    from scipy import ndimage
    occ = box_room_voxels(lo, hi)
    mn = occ.min(axis=0) - pad
    mx = occ.max(axis=0) + pad
    shape = tuple((mx - mn + 1).tolist())  # (x, y, z)
    grid = np.ones(shape, dtype=bool)
    grid[tuple((occ - mn).T)] = False
    d = ndimage.distance_transform_edt(grid) * resolution
    ratio = np.floor(np.minimum(d, max_dist) / max_dist * 255).astype(np.uint8)
    w, h, nz = shape
    pose_indices = np.zeros(w * h, dtype=np.uint32)
    cols = [np.full(nz, 255, dtype=np.uint8)]
    nxt = nz
    for j in range(h):
        for i in range(w):
            col = ratio[i, j, :]
            if (col < 255).any():
                pose_indices[j * w + i] = nxt
                cols.append(col)
                nxt += nz
    return pose_indices, np.concatenate(cols), mn.astype(np.int32), mx.astype(np.int32)


def grid_cloud(rows=64, cols=1024, origin=(0.3, 0.2, 0.6), room_lo=(-2.0, -1.5, -0.1), room_hi=(2.0, 1.5, 1.0),
               seed=7, noise=0.01):
    """rows x cols spherical-grid cloud (scanner frame, scanner at `origin` with zero rotation) of
    the axis-aligned box room, by analytic ray / box intersection."""
    rng = np.random.default_rng(seed)
    el = np.linspace(-0.5, 0.5, rows)[:, None]
    az = np.linspace(-np.pi, np.pi, cols, endpoint=False)[None, :]
    d = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el) * np.ones_like(az)], axis=-1)
    o = np.asarray(origin)
    with np.errstate(divide="ignore", invalid="ignore"):
        t_lo = (np.asarray(room_lo) - o) / d
        t_hi = (np.asarray(room_hi) - o) / d
    t = np.where(d > 0, t_hi, t_lo)
    t = np.where(np.isfinite(t) & (t > 0), t, np.inf).min(axis=-1)
    pts = d * (t + rng.normal(0, noise, t.shape))[..., None]
    return pts.reshape(-1, 3).astype(np.float32)
