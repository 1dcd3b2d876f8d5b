"""ctypes front end of the CPU oracle (oracle/amcl_oracle.c).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke()
and the cpu_baseline leg of bench.py -- never by badger_amcl_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libamcl_oracle.so")

MODEL_BEAM, MODEL_LF, MODEL_LF_PROB, MODEL_LF_GOMPERTZ = 0, 1, 2, 3
RESAMPLE_MULTINOMIAL, RESAMPLE_SYSTEMATIC = 0, 1
CLOUD_MODEL, CLOUD_MODEL_GOMPERTZ = 0, 1


def build(force=False):
    src = os.path.join(_HERE, "amcl_oracle.c")
    hdr = os.path.join(_HERE, "amcl_oracle.h")
    if (not force and os.path.exists(_SO)
            and (not os.path.exists(src)
                 or os.path.getmtime(_SO) >= max(os.path.getmtime(src), os.path.getmtime(hdr)))):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libamcl_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Map2D(C.Structure):
    _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("origin_x", C.c_float), ("origin_y", C.c_float),
                ("resolution", C.c_double), ("max_dist", C.c_double),
                ("cells", C.POINTER(C.c_int32)), ("lut", C.POINTER(C.c_float))]


class Planar(C.Structure):
    _fields_ = [("model", C.c_int), ("max_beams", C.c_int),
                ("z_hit", C.c_double), ("z_short", C.c_double), ("z_max", C.c_double), ("z_rand", C.c_double),
                ("sigma_hit", C.c_double), ("lambda_short", C.c_double),
                ("gompertz_a", C.c_double), ("gompertz_b", C.c_double), ("gompertz_c", C.c_double),
                ("input_shift", C.c_double), ("input_scale", C.c_double), ("output_shift", C.c_double),
                ("do_beamskip", C.c_int),
                ("beam_skip_distance", C.c_double), ("beam_skip_threshold", C.c_double),
                ("beam_skip_error_threshold", C.c_double),
                ("off_map_factor", C.c_double), ("non_free_space_factor", C.c_double),
                ("non_free_space_radius", C.c_double),
                ("scanner_pose", C.c_double * 3)]


class PF(C.Structure):
    _fields_ = [("min_samples", C.c_int), ("max_samples", C.c_int),
                ("pop_err", C.c_double), ("pop_z", C.c_double),
                ("alpha_slow", C.c_double), ("alpha_fast", C.c_double),
                ("w_slow", C.c_double), ("w_fast", C.c_double),
                ("dist_threshold", C.c_double), ("convergence_threshold", C.c_double),
                ("resample_model", C.c_int), ("rng", C.c_uint64), ("converged", C.c_int),
                ("random_source", C.c_void_p)]


class FreeSpace(C.Structure):
    _fields_ = [("n", C.c_int), ("ij", C.POINTER(C.c_int)), ("size_x", C.c_int), ("size_y", C.c_int),
                ("origin_x", C.c_double), ("origin_y", C.c_double), ("resolution", C.c_double)]


class ResampleOut(C.Structure):
    _fields_ = [("sample_count", C.c_int), ("leaf_count", C.c_int), ("node_count", C.c_int),
                ("cluster_count", C.c_int), ("converged", C.c_int),
                ("mean", C.c_double * 3), ("cov", C.c_double * 4), ("cov_theta", C.c_double),
                ("w_diff", C.c_double), ("status", C.c_int), ("percent_converged", C.c_float)]


class Map3D(C.Structure):
    _fields_ = [("min_cells", C.c_int * 3), ("max_cells", C.c_int * 3), ("width", C.c_int), ("num_z", C.c_int),
                ("resolution", C.c_double), ("max_dist", C.c_double),
                ("pose_indices", C.POINTER(C.c_uint32)), ("distance_ratios", C.POINTER(C.c_uint8))]


class Cloud(C.Structure):
    _fields_ = [("model", C.c_int), ("max_beams", C.c_int),
                ("z_hit", C.c_double), ("z_rand", C.c_double), ("sigma_hit", C.c_double),
                ("gompertz_a", C.c_double), ("gompertz_b", C.c_double), ("gompertz_c", C.c_double),
                ("input_shift", C.c_double), ("input_scale", C.c_double), ("output_shift", C.c_double),
                ("off_map_factor", C.c_double), ("tf_xyz", C.c_double * 3), ("tf_quat", C.c_double * 4)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp = C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int)
    lp = C.POINTER(C.c_long)
    u64p = C.POINTER(C.c_uint64)
    L.orc_srand48.argtypes = [u64p, C.c_long]
    L.orc_drand48.argtypes = [u64p]
    L.orc_drand48.restype = C.c_double
    L.orc_gaussian_draw.argtypes = [u64p, C.c_double]
    L.orc_gaussian_draw.restype = C.c_double
    L.orc_normalize_angle.argtypes = [C.c_double]
    L.orc_normalize_angle.restype = C.c_double
    L.orc_map2d_world_to_map.argtypes = [C.POINTER(Map2D), C.c_double, C.c_double, ip, ip]
    L.orc_map2d_map_to_world.argtypes = [C.POINTER(Map2D), C.c_int, C.c_int, dp, dp]
    L.orc_map2d_is_valid.argtypes = [C.POINTER(Map2D), C.c_int, C.c_int]
    L.orc_map2d_distance.argtypes = [C.POINTER(Map2D), C.c_int, C.c_int]
    L.orc_map2d_distance.restype = C.c_float
    L.orc_map2d_calc_range.argtypes = [C.POINTER(Map2D), C.c_double, C.c_double, C.c_double, C.c_double, lp]
    L.orc_map2d_calc_range.restype = C.c_double
    L.orc_map2d_build_lut.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_double, C.c_double,
                                      C.POINTER(C.c_float)]
    L.orc_planar_defaults.argtypes = [C.POINTER(Planar)]
    L.orc_planar_apply.argtypes = [C.POINTER(Planar), C.POINTER(Map2D), dp, C.c_int, C.c_int, dp, dp, C.c_int,
                                   C.c_double, lp]
    L.orc_planar_apply.restype = C.c_double
    L.orc_kdtree_new.restype = C.c_void_p
    L.orc_kdtree_free.argtypes = [C.c_void_p]
    L.orc_kdtree_clear.argtypes = [C.c_void_p]
    L.orc_kdtree_insert.argtypes = [C.c_void_p, dp, C.c_double]
    L.orc_kdtree_insert_key.argtypes = [C.c_void_p, ip, C.c_double]
    L.orc_kdtree_leaf_count.argtypes = [C.c_void_p]
    L.orc_kdtree_node_count.argtypes = [C.c_void_p]
    L.orc_kdtree_cluster.argtypes = [C.c_void_p]
    L.orc_kdtree_get_cluster.argtypes = [C.c_void_p, dp]
    L.orc_pf_init.argtypes = [C.POINTER(PF), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
    L.orc_pf_normalize.argtypes = [C.POINTER(PF), dp, C.c_int, C.c_double]
    L.orc_pf_resample_limit.argtypes = [C.POINTER(PF), C.c_int]
    L.orc_pf_update_resample.argtypes = [C.POINTER(PF), dp, C.c_int, C.c_int, dp, ip, C.POINTER(ResampleOut)]
    L.orc_pf_update_converged.argtypes = [C.POINTER(PF), dp, C.c_int, C.POINTER(C.c_float)]
    L.orc_pf_init_with_free_space_poses.argtypes = [C.POINTER(PF), C.POINTER(FreeSpace), dp, C.c_int, ip]
    L.orc_pf_init_with_gaussian.argtypes = [C.POINTER(PF), dp, dp, dp, dp, C.c_int, ip]
    L.orc_free_space_indices.argtypes = [C.POINTER(Map2D), C.c_double, ip, C.c_int]
    L.orc_random_free_space_pose.argtypes = [C.POINTER(FreeSpace), C.POINTER(C.c_uint64), dp]
    L.orc_random_free_space_pose.restype = None
    L.orc_odom_update_action.argtypes = [C.c_int, dp, dp, dp, dp, dp, C.c_int, C.POINTER(C.c_uint64)]
    L.orc_odom_update_action.restype = None
    L.orc_pf_cluster_stats.argtypes = [C.c_void_p, dp, C.c_int, C.c_int, ip, dp, dp, dp, dp, dp]
    L.orc_map3d_world_to_map.argtypes = [C.POINTER(Map3D), dp, ip]
    L.orc_map3d_map_to_world.argtypes = [C.c_double, ip, dp]
    L.orc_map3d_distance.argtypes = [C.POINTER(Map3D), C.c_int, C.c_int, C.c_int]
    L.orc_map3d_distance.restype = C.c_double
    L.orc_map3d_build_lut.argtypes = [ip, ip, C.c_double, C.c_double, ip, C.c_size_t, C.POINTER(C.c_uint32),
                                      C.POINTER(C.c_uint8), C.c_size_t]
    L.orc_map3d_build_lut.restype = C.c_size_t
    L.orc_cloud_apply.argtypes = [C.POINTER(Cloud), C.POINTER(Map3D), dp, C.c_int, C.POINTER(C.c_float), C.c_int,
                                  lp]
    L.orc_cloud_apply.restype = C.c_double
    fp = C.POINTER(C.c_float)
    L.orc_wire_laserscan_to_planar.argtypes = [fp, C.c_int, C.c_float, C.c_float, C.c_double, C.c_double, C.c_double,
                                               C.c_double, dp, dp, dp]
    L.orc_wire_laserscan_to_planar.restype = None
    L.orc_wire_scan_angle_stats.argtypes = [C.c_double, C.c_double, dp, dp, dp]
    L.orc_wire_scan_angle_stats.restype = None
    L.orc_wire_convert_map.argtypes = [C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                       C.c_int, C.POINTER(C.c_int32), ip, fp, dp]
    L.orc_wire_convert_map.restype = None
    L.orc_wire_decimate_cloud.argtypes = [fp, C.c_int, C.c_int, fp]
    L.orc_wire_pose_array.argtypes = [dp, C.c_int, dp]
    L.orc_wire_pose_array.restype = None
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


# --------------------------------------------------------------------------- RNG
class Rng:
    """glibc drand48 stream.  seed=None -> unseeded glibc state (X = 0)."""

    def __init__(self, seed=None):
        self.state = C.c_uint64(0)
        if seed is not None:
            lib().orc_srand48(C.byref(self.state), seed)

    def drand48(self):
        return lib().orc_drand48(C.byref(self.state))

    def gaussian(self, sigma):
        return lib().orc_gaussian_draw(C.byref(self.state), sigma)


# ------------------------------------------------------------------------ 2-D map
class OccupancyMap:
    """Flat-array stand-in for the reference OccupancyMap state (T3)."""

    def __init__(self, cells, resolution, origin=(0.0, 0.0), max_dist=0.0, lut=None):
        cells = np.ascontiguousarray(cells, dtype=np.int32)  # [size_y, size_x], index i + j*size_x
        self.size_y, self.size_x = cells.shape
        self.cells = cells
        self.resolution = float(resolution)
        self.origin = (np.float32(origin[0]), np.float32(origin[1]))
        self.max_dist = float(max_dist)
        self.lut = None if lut is None else np.ascontiguousarray(lut, dtype=np.float32)

    def update_distances_lut(self, max_dist):
        self.max_dist = float(max_dist)
        lut = np.zeros((self.size_y, self.size_x), dtype=np.float32)
        lib().orc_map2d_build_lut(self.size_x, self.size_y, self.cells.ctypes.data_as(C.POINTER(C.c_int32)),
                                  self.resolution, self.max_dist, lut.ctypes.data_as(C.POINTER(C.c_float)))
        self.lut = lut
        return lut

    def struct(self):
        m = Map2D()
        m.size_x, m.size_y = self.size_x, self.size_y
        m.origin_x, m.origin_y = float(self.origin[0]), float(self.origin[1])
        m.resolution, m.max_dist = self.resolution, self.max_dist
        m.cells = self.cells.ctypes.data_as(C.POINTER(C.c_int32))
        m.lut = (self.lut.ctypes.data_as(C.POINTER(C.c_float)) if self.lut is not None
                 else C.POINTER(C.c_float)())
        return m

    def world_to_map(self, x, y):
        i, j = C.c_int(), C.c_int()
        m = self.struct()
        lib().orc_map2d_world_to_map(C.byref(m), x, y, C.byref(i), C.byref(j))
        return i.value, j.value

    def map_to_world(self, i, j):
        x, y = C.c_double(), C.c_double()
        m = self.struct()
        lib().orc_map2d_map_to_world(C.byref(m), i, j, C.byref(x), C.byref(y))
        return x.value, y.value

    def is_valid(self, i, j):
        m = self.struct()
        return bool(lib().orc_map2d_is_valid(C.byref(m), i, j))

    def calc_range(self, ox, oy, oa, max_range):
        m = self.struct()
        return lib().orc_map2d_calc_range(C.byref(m), ox, oy, oa, max_range, None)


def planar(model=MODEL_LF, max_beams=30, **kw):
    p = Planar()
    lib().orc_planar_defaults(C.byref(p))
    p.model = model
    p.max_beams = max_beams
    for k, v in kw.items():
        if k == "scanner_pose":
            p.scanner_pose[:] = v
        else:
            if not hasattr(p, k):
                raise AttributeError(k)
            setattr(p, k, v)
    return p


def planar_apply(p, omap, samples, ranges, angles, range_max, set_converged=0, stats=None):
    """samples: float64 [N,4] (x, y, theta, weight), modified in place.  Returns total."""
    assert samples.dtype == np.float64 and samples.flags.c_contiguous
    ranges = np.ascontiguousarray(ranges, dtype=np.float64)
    angles = np.ascontiguousarray(angles, dtype=np.float64)
    m = omap.struct()
    st = (C.c_long * 2)(0, 0)
    total = lib().orc_planar_apply(C.byref(p), C.byref(m), _dp(samples), samples.shape[0], int(set_converged),
                                   _dp(ranges), _dp(angles), ranges.shape[0], float(range_max), st)
    if stats is not None:
        stats["evals"] = stats.get("evals", 0) + st[0]
        stats["cells"] = stats.get("cells", 0) + st[1]
    return total


# ------------------------------------------------------------------------ kd-tree
class KDTree:
    def __init__(self):
        self.h = lib().orc_kdtree_new()

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_kdtree_free(self.h)
            self.h = None

    def clear(self):
        lib().orc_kdtree_clear(self.h)

    def insert_pose(self, pose, value=0.0):
        a = np.asarray(pose, dtype=np.float64)
        lib().orc_kdtree_insert(self.h, _dp(a), value)

    def insert_key(self, key, value=0.0):
        a = np.asarray(key, dtype=np.int32)
        lib().orc_kdtree_insert_key(self.h, _ip(a), value)

    def leaf_count(self):
        return lib().orc_kdtree_leaf_count(self.h)

    def node_count(self):
        return lib().orc_kdtree_node_count(self.h)

    def cluster(self):
        lib().orc_kdtree_cluster(self.h)

    def get_cluster(self, pose):
        a = np.asarray(pose, dtype=np.float64)
        return lib().orc_kdtree_get_cluster(self.h, _dp(a))

    def cluster_stats(self, samples, max_clusters):
        """computeClusterStatsForSet on this tree: dict(count, weight, mean, cov, set_mean, set_cov)."""
        s = np.ascontiguousarray(samples, dtype=np.float64)
        cnt = np.zeros(max_clusters, dtype=np.int32)
        w = np.zeros(max_clusters)
        mean = np.zeros((max_clusters, 3))
        cov = np.zeros((max_clusters, 5))
        sm, sc = np.zeros(3), np.zeros(5)
        k = lib().orc_pf_cluster_stats(self.h, _dp(s), s.shape[0], max_clusters, _ip(cnt), _dp(w), _dp(mean),
                                       _dp(cov), _dp(sm), _dp(sc))
        return dict(n=k, count=cnt[:k], weight=w[:k], mean=mean[:k], cov=cov[:k], set_mean=sm, set_cov=sc)


# ----------------------------------------------------------------- motion model
ODOM_MODEL_DIFF, ODOM_MODEL_OMNI, ODOM_MODEL_DIFF_CORRECTED, ODOM_MODEL_OMNI_CORRECTED, ODOM_MODEL_GAUSSIAN = range(5)


def odom_update_action(model, alpha, pose, delta, absolute_motion, samples, rng_state):
    """Odom::updateAction on samples[n,4] in place; returns the drand48 state afterwards."""
    a = np.asarray(alpha, dtype=np.float64)
    po, de, am = (np.asarray(v, dtype=np.float64) for v in (pose, delta, absolute_motion))
    assert samples.dtype == np.float64 and samples.flags.c_contiguous
    st = C.c_uint64(rng_state)
    lib().orc_odom_update_action(model, _dp(a), _dp(po), _dp(de), _dp(am), _dp(samples), samples.shape[0],
                                 C.byref(st))
    return st.value


# ----------------------------------------------------------------- particle filter
class ParticleFilter:
    """Flat-array restatement of the reference ParticleFilter state machine."""

    def __init__(self, min_samples, max_samples, alpha_slow=0.0, alpha_fast=0.0, convergence_threshold=85.0,
                 seed=None):
        self.pf = PF()
        lib().orc_pf_init(C.byref(self.pf), min_samples, max_samples, alpha_slow, alpha_fast,
                          convergence_threshold)
        if seed is not None:
            st = C.c_uint64(0)
            lib().orc_srand48(C.byref(st), seed)
            self.pf.rng = st.value
        self.samples = np.zeros((max_samples, 4), dtype=np.float64)
        self.samples[:, 3] = 1.0 / max_samples
        self.sample_count = max_samples
        self.leaf_count = 0
        self.set_converged = 0
        self.last = None

    def set_samples(self, samples, leaf_count=None):
        n = samples.shape[0]
        self.samples = np.zeros((self.pf.max_samples, 4), dtype=np.float64)
        self.samples[:n] = samples
        self.sample_count = n
        if leaf_count is None:
            t = KDTree()
            for k in range(n):
                t.insert_pose(self.samples[k, :3], self.samples[k, 3])
            leaf_count = t.leaf_count()
        self.leaf_count = leaf_count

    def set_population_size_parameters(self, pop_err, pop_z):
        self.pf.pop_err, self.pf.pop_z = pop_err, pop_z

    def init_with_free_space_poses(self):
        """initWithPoseFn(Node::randomFreeSpacePose); needs set_random_pose_source."""
        n = self.pf.max_samples
        self.samples = np.zeros((n, 4), dtype=np.float64)
        nodes = C.c_int()
        self.leaf_count = lib().orc_pf_init_with_free_space_poses(C.byref(self.pf), C.byref(self._free),
                                                                  _dp(self.samples), n, C.byref(nodes))
        self.sample_count, self.node_count, self.set_converged = n, nodes.value, 0

    def init_with_gaussian(self, mean, cr, cd):
        n = self.pf.max_samples
        self.samples = np.zeros((n, 4), dtype=np.float64)
        nodes = C.c_int()
        m, r, d = (np.ascontiguousarray(v, dtype=np.float64) for v in (mean, cr, cd))
        self.leaf_count = lib().orc_pf_init_with_gaussian(C.byref(self.pf), _dp(m), _dp(r.reshape(-1)), _dp(d),
                                                          _dp(self.samples), n, C.byref(nodes))
        self.sample_count, self.node_count, self.set_converged = n, nodes.value, 0

    def set_random_pose_source(self, omap, non_free_space_radius):
        """random_pose_fn_ = Node::randomFreeSpacePose over Node2D::updateFreeSpaceIndices of `omap`."""
        m = omap.struct()
        n = lib().orc_free_space_indices(C.byref(m), non_free_space_radius, None, 0)
        self._free_ij = np.zeros((max(n, 1), 2), dtype=np.int32)
        lib().orc_free_space_indices(C.byref(m), non_free_space_radius, _ip(self._free_ij), n)
        self._free = FreeSpace(n, self._free_ij.ctypes.data_as(C.POINTER(C.c_int)), omap.size_x, omap.size_y,
                               float(omap.origin[0]), float(omap.origin[1]), omap.resolution)
        self._free_map = omap  # keep the arrays alive
        self.pf.random_source = C.cast(C.pointer(self._free), C.c_void_p)
        return n

    def set_resample_model(self, model):
        self.pf.resample_model = model

    def set_decay_rates(self, a_slow, a_fast):
        self.pf.alpha_slow, self.pf.alpha_fast = a_slow, a_fast

    def resample_limit(self, k):
        return lib().orc_pf_resample_limit(C.byref(self.pf), k)

    def update_sensor(self, sensor_fn):
        """sensor_fn(samples_view[N,4], set_converged) -> total (mutates weights)."""
        view = self.samples[:self.sample_count]
        total = sensor_fn(view, self.set_converged)
        lib().orc_pf_normalize(C.byref(self.pf), _dp(self.samples), self.sample_count, total)
        return total

    def update_resample(self):
        out = ResampleOut()
        set_b = np.zeros((self.pf.max_samples, 4), dtype=np.float64)
        idx = np.full(self.pf.max_samples, -1, dtype=np.int32)
        lib().orc_pf_update_resample(C.byref(self.pf), _dp(self.samples), self.sample_count, self.leaf_count,
                                     _dp(set_b), _ip(idx), C.byref(out))
        self.samples = set_b
        self.sample_count = out.sample_count
        self.leaf_count = out.leaf_count
        self.set_converged = out.converged
        self.last = out
        self.last_idx = idx[:out.sample_count].copy()
        return out


# --------------------------------------------------------------------------- 3-D
class OctoMapLUT:
    """Two-level uint8 distance LUT of the reference OctoMap (octomap.cpp:315-355)."""

    def __init__(self, min_cells, max_cells, resolution, max_dist, pose_indices=None, distance_ratios=None):
        self.min_cells = np.asarray(min_cells, dtype=np.int32)
        self.max_cells = np.asarray(max_cells, dtype=np.int32)
        self.resolution = float(resolution)
        self.max_dist = float(max_dist)
        self.width = int(self.max_cells[0] - self.min_cells[0] + 1)
        self.height = int(self.max_cells[1] - self.min_cells[1] + 1)
        self.num_z = int(self.max_cells[2] - self.min_cells[2] + 1)
        self.pose_indices = pose_indices
        self.distance_ratios = distance_ratios

    def build(self, occupied_ijk, cap_columns=None):
        occ = np.ascontiguousarray(occupied_ijk, dtype=np.int32)
        nposes = self.width * self.height
        cap = (cap_columns if cap_columns is not None else nposes + 1) * self.num_z
        pi = np.zeros(nposes, dtype=np.uint32)
        dr = np.zeros(cap, dtype=np.uint8)
        n = lib().orc_map3d_build_lut(_ip(self.min_cells), _ip(self.max_cells), self.resolution, self.max_dist,
                                      _ip(occ), occ.shape[0], pi.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      dr.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
        if n == 0:
            raise RuntimeError("3-D LUT capacity overflow")
        self.pose_indices = pi
        self.distance_ratios = dr[:n].copy()
        return self

    def struct(self):
        m = Map3D()
        m.min_cells[:] = [int(v) for v in self.min_cells]
        m.max_cells[:] = [int(v) for v in self.max_cells]
        m.width, m.num_z = self.width, self.num_z
        m.resolution, m.max_dist = self.resolution, self.max_dist
        m.pose_indices = self.pose_indices.ctypes.data_as(C.POINTER(C.c_uint32))
        m.distance_ratios = self.distance_ratios.ctypes.data_as(C.POINTER(C.c_uint8))
        return m

    def world_to_map(self, w):
        w = np.asarray(w, dtype=np.float64)
        c = np.zeros(3, dtype=np.int32)
        m = self.struct()
        lib().orc_map3d_world_to_map(C.byref(m), _dp(w), _ip(c))
        return c

    def distance(self, i, j, k):
        m = self.struct()
        return lib().orc_map3d_distance(C.byref(m), i, j, k)


def map3d_map_to_world(resolution, c):
    c = np.asarray(c, dtype=np.int32)
    w = np.zeros(3, dtype=np.float64)
    lib().orc_map3d_map_to_world(resolution, _ip(c), _dp(w))
    return w


def cloud(model=CLOUD_MODEL, max_beams=128, tf_xyz=(0, 0, 0), tf_quat=(0, 0, 0, 1), **kw):
    p = Cloud()
    p.model, p.max_beams = model, max_beams
    p.off_map_factor = 1.0
    p.tf_xyz[:] = tf_xyz
    p.tf_quat[:] = tf_quat
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def cloud_apply(p, lut, samples, points, stats=None):
    assert samples.dtype == np.float64 and samples.flags.c_contiguous
    pts = np.ascontiguousarray(points, dtype=np.float32)
    m = lut.struct()
    st = (C.c_long * 2)(0, 0)
    total = lib().orc_cloud_apply(C.byref(p), C.byref(m), _dp(samples), samples.shape[0],
                                  pts.ctypes.data_as(C.POINTER(C.c_float)), pts.shape[0], st)
    if stats is not None:
        stats["evals"] = stats.get("evals", 0) + st[0]
    return total


# ------------------------------------------------------------------- wire formats
def wire_laserscan_to_planar(ranges_f32, range_min, range_max, angle_min, angle_increment, sensor_min_range=-1.0,
                             sensor_max_range=-1.0):
    r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
    ro, ao = np.zeros(r.size), np.zeros(r.size)
    rmax = C.c_double()
    lib().orc_wire_laserscan_to_planar(r.ctypes.data_as(C.POINTER(C.c_float)), r.size, range_min, range_max,
                                       sensor_min_range, sensor_max_range, angle_min, angle_increment, _dp(ro),
                                       _dp(ao), C.byref(rmax))
    return ro, ao, rmax.value


def wire_scan_angle_stats(angle_min, angle_increment, q_base_scanner):
    q = np.ascontiguousarray(q_base_scanner, dtype=np.float64)
    a, b = C.c_double(), C.c_double()
    lib().orc_wire_scan_angle_stats(angle_min, angle_increment, _dp(q), C.byref(a), C.byref(b))
    return a.value, b.value


def wire_convert_map(data_i8, width, height, resolution, origin_x, origin_y, scale_up=1):
    d = np.ascontiguousarray(data_i8, dtype=np.int8)
    cells = np.zeros(width * scale_up * height * scale_up, dtype=np.int32)
    size = (C.c_int * 2)()
    origin = (C.c_float * 2)()
    res = C.c_double()
    lib().orc_wire_convert_map(d.ctypes.data_as(C.POINTER(C.c_int8)), width, height, resolution, origin_x, origin_y,
                               scale_up, cells.ctypes.data_as(C.POINTER(C.c_int32)), size, origin, C.byref(res))
    return cells.reshape(size[1], size[0]), (np.float32(origin[0]), np.float32(origin[1])), res.value


def wire_decimate_cloud(points, max_beams):
    p = np.ascontiguousarray(points, dtype=np.float32)
    out = np.zeros_like(p)
    k = lib().orc_wire_decimate_cloud(p.ctypes.data_as(C.POINTER(C.c_float)), p.shape[0], max_beams,
                                      out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:k].copy()


def wire_pose_array(samples):
    s = np.ascontiguousarray(samples, dtype=np.float64)
    out = np.zeros((s.shape[0], 7))
    lib().orc_wire_pose_array(_dp(s), s.shape[0], _dp(out))
    return out
