"""Host-side mirror of the reference interface for the hot path, on top of the C-ABI.

Class and method names follow the reference (include/amcl/...): OccupancyMap,
PlanarData, PlanarScanner, ParticleFilter -- so a parity test reads like the reference's
own call sequence:

    map = OccupancyMap(engine, resolution); map.setSize(...); map.setOrigin(...); ...
    scanner = PlanarScanner(engine); scanner.init(max_beams, map)
    scanner.setModelLikelihoodField(z_hit, z_rand, sigma_hit, max_dist)
    pf = ParticleFilter(engine, min_samples, max_samples, alpha_slow, alpha_fast, thr)
    scanner.updateSensor(pf, data); pf.updateResample()

All computation happens in libbadger_pf_hip.so on the GPU.  This module only marshals.
"""
import ctypes as C
import math

import numpy as np

from . import _lib

MODEL_BEAM, MODEL_LIKELIHOOD_FIELD, MODEL_LIKELIHOOD_FIELD_PROB, MODEL_LIKELIHOOD_FIELD_GOMPERTZ = 0, 1, 2, 3
PF_RESAMPLE_MULTINOMIAL, PF_RESAMPLE_SYSTEMATIC = 0, 1
RANDOM_POSE_NONE, RANDOM_POSE_FREE_SPACE_2D = 0, 1
OPT_CDF_SERIAL, OPT_COUNT_CELLS, OPT_WINDOW_PATH, OPT_KLD_DEVICE_MIN, OPT_GRADED_SHARES = 0, 1, 2, 3, 4
OPT_FUSED_RESAMPLE = 5
OPT_CLOUD_DENSE = 6
OPT_STATS_HOST = 7
OPT_LUT_HOST = 8
OPT_KLD_PERSISTENT = 9
OPT_LUT_EXACT_EDT = 10
OPT_HOST_AUTO_REGISTER = 11
OPT_SEAM_CHUNKS = 12
OPT_KLD_LOCAL = 13
OPT_TILE_SORT = 14
OPT_HOST_DIRECT_PAGEABLE = 15  # pageable buffers straight to the HIP runtime (the caller promises they never move)
CELL_FREE, CELL_UNKNOWN, CELL_OCCUPIED = -1, 0, 1


class BpfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bpf error %d: %s" % (code, msg))
        self.code = code


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Engine:
    """One HIP engine = one GPU-resident map + scanner model + particle filter."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.bpf_create(device, C.byref(h))
        if rc != 0:
            raise BpfError(rc, "bpf_create failed (%s)" % self.lib.bpf_error_string(rc).decode())
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.bpf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise BpfError(rc, self.lib.bpf_last_error_message(self.h).decode() or
                           self.lib.bpf_error_string(rc).decode())

    def set_stream(self, hip_stream):
        """hip_stream: a hipStream_t handle as an int (0 = HIP's default stream), or None for the
        engine's own stream."""
        handle = C.c_void_p(-1) if hip_stream is None else C.c_void_p(hip_stream)
        self.check(self.lib.bpf_set_stream(self.h, handle))

    def synchronize(self):
        self.check(self.lib.bpf_synchronize(self.h))

    def profile_enable(self, on=1):
        """1 = time every 8th launch of the scoring kernel, 3 = every launch of it, 2 = every kernel class, 0 = off."""
        self.check(self.lib.bpf_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self.check(self.lib.bpf_profile_reset(self.h))

    def profile_get(self):
        p = _lib.Profile()
        self.check(self.lib.bpf_profile_get(self.h, C.byref(p)))
        names = ["score", "reduce", "normalize", "cdf", "draw", "finalize", "score_window", "score_aux", "motion"]
        return {n: {"ms": p.ms[i], "launches": p.launches[i]} for i, n in enumerate(names)}

    def registerHostBuffer(self, array):
        """Pins a caller-owned numpy buffer for the host-buffer entry points (bpf_host_buffer_register): the owner
        keeps it allocated until unregisterHostBuffer / close."""
        self.check(self.lib.bpf_host_buffer_register(self.h, C.c_void_p(array.ctypes.data), array.nbytes))

    def unregisterHostBuffer(self, array):
        self.check(self.lib.bpf_host_buffer_unregister(self.h, C.c_void_p(array.ctypes.data)))

    def isHostBufferRegistered(self, array):
        return bool(self.lib.bpf_host_buffer_is_registered(self.h, C.c_void_p(array.ctypes.data), array.nbytes))

    def score_last_form(self):
        """3 = the last scoring launch of a resident set walked the particles in map-tile order, 0 = index order."""
        a = C.c_int()
        self.check(self.lib.bpf_score_last_form(self.h, C.byref(a)))
        return a.value

    def kld_last_form(self):
        """2 = the last device-side histogram tree was grown in LDS-sized pieces, 1 = level loop, 3 = persistent."""
        a = C.c_int()
        self.check(self.lib.bpf_kld_last_form(self.h, C.byref(a)))
        return a.value

    def seam_last_plan(self):
        """(chunks, pinned) of the last applyModelToSampleSet: 0 chunks = the plain upload / score / download, -1 = one
        launch that read and wrote the registered records in place."""
        a, b = C.c_int(), C.c_int()
        self.check(self.lib.bpf_seam_last_plan(self.h, C.byref(a), C.byref(b)))
        return a.value, bool(b.value)

    def set_option(self, option, value):
        self.check(self.lib.bpf_set_option(self.h, option, int(value)))

    def cells_walked(self, reset=True):
        v = C.c_ulonglong()
        self.check(self.lib.bpf_get_cells_walked(self.h, C.byref(v), int(reset)))
        return v.value

    def window_plan(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.check(self.lib.bpf_get_window_plan(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(used_window=bool(a.value), chunks_covered=b.value, chunks_total=c.value)

    def score_kernel_name(self):
        return self.lib.bpf_score_kernel_name(self.h).decode()


class OccupancyMap:
    """include/amcl/map/occupancy_map.h:54-123 (state only; queries run on the GPU)."""

    def __init__(self, engine, resolution):
        self.e = engine
        self.resolution = float(resolution)
        self.size_x = self.size_y = 0
        self.origin = (np.float32(0), np.float32(0))
        self.cells = None
        self.max_distance_to_object = 0.0
        self.lut = None
        self._dirty = True

    def setSize(self, size_vec):
        self.size_x, self.size_y = int(size_vec[0]), int(size_vec[1])
        self.cells = np.zeros((self.size_y, self.size_x), dtype=np.int32)
        self._dirty = True

    def getSize(self):
        return [self.size_x, self.size_y]

    def setOrigin(self, origin_xy):
        self.origin = (np.float32(origin_xy[0]), np.float32(origin_xy[1]))  # pcl::PointXYZ is float
        self._dirty = True

    def computeCellIndex(self, i, j):
        return i + j * self.size_x

    def setCellState(self, index, state):
        self.cells.reshape(-1)[index] = state
        self._dirty = True

    def setCells(self, cells):
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.size_y, self.size_x = cells.shape
        self.cells = cells
        self._dirty = True

    def setDistancesLUT(self, lut, max_distance_to_object):
        """Adopt a host-built distances_lut_ (e.g. the reference's brushfire output)."""
        self.lut = np.ascontiguousarray(lut, dtype=np.float32)
        self.max_distance_to_object = float(max_distance_to_object)
        self._dirty = True

    def upload(self):
        if not self._dirty:
            return
        lutp = self.lut.ctypes.data_as(C.POINTER(C.c_float)) if self.lut is not None else None
        self.e.check(self.e.lib.bpf_map2d_set(self.e.h, self.cells.ctypes.data_as(C.POINTER(C.c_int32)), lutp,
                                              self.size_x, self.size_y, float(self.origin[0]), float(self.origin[1]),
                                              self.resolution, self.max_distance_to_object))
        self._dirty = False

    def updateDistancesLUT(self, max_distance_to_object):
        """OccupancyMap::updateDistancesLUT (occupancy_map.cpp:138-252): the reference's priority-queue brushfire, its
        values bit for bit (host builder, once per map as in the reference)."""
        self.upload()
        self.e.check(self.e.lib.bpf_map2d_build_distances_lut_reference(self.e.h, float(max_distance_to_object)))
        self.max_distance_to_object = float(max_distance_to_object)
        self.lut = None

    updateDistancesLUTReference = updateDistancesLUT

    def updateDistancesLUTExact(self, max_distance_to_object):
        """NOT a reference method: the exact capped EDT built on the device (milliseconds); <= the brushfire's values
        and different from them in < 1 % of the cells -- the caller's explicit choice."""
        self.upload()
        self.e.check(self.e.lib.bpf_map2d_build_distances_lut(self.e.h, float(max_distance_to_object)))
        self.max_distance_to_object = float(max_distance_to_object)
        self.lut = None

    def calcRange(self, ox, oy, oa, max_range):
        """OccupancyMap::calcRange (occupancy_map.cpp:257-364), batched: arrays of origins, angles and max ranges."""
        self.upload()  # the cells alone are enough (no distance LUT needed), as for the reference's calcRange
        ox, oy, oa = (np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (ox, oy, oa))
        mr = np.ascontiguousarray(np.broadcast_to(np.asarray(max_range, dtype=np.float64), ox.shape))
        # libm's cos / sin (what the reference calls), not numpy's vector loops, which may differ in the last bit
        ca = np.array([math.cos(a) for a in oa], dtype=np.float64)
        sa = np.array([math.sin(a) for a in oa], dtype=np.float64)
        out = np.zeros(ox.shape[0], dtype=np.float64)
        self.e.check(self.e.lib.bpf_map2d_calc_range(self.e.h, _dp(ox), _dp(oy), _dp(ca), _dp(sa), _dp(mr),
                                                     ox.shape[0], _dp(out)))
        return out

    def getDistancesLUT(self):
        out = np.zeros((self.size_y, self.size_x), dtype=np.float32)
        self.e.check(self.e.lib.bpf_map2d_get_distances_lut(self.e.h, out.ctypes.data_as(C.POINTER(C.c_float)),
                                                            out.size))
        return out


class PlanarData:
    """include/amcl/sensors/planar_scanner.h:45-54"""

    def __init__(self, ranges, angles, range_max):
        self.ranges_ = np.ascontiguousarray(ranges, dtype=np.float64)
        self.angles_ = np.ascontiguousarray(angles, dtype=np.float64)
        self.range_count_ = int(self.ranges_.shape[0])
        self.range_max_ = float(range_max)
        self._ptr_of = None

    def pointers(self):
        """(ranges, angles) as C pointers; formed once per pair of arrays (numpy's ctypes view costs ~2 us a time)."""
        if self._ptr_of is None or self._ptr_of[0] is not self.ranges_ or self._ptr_of[1] is not self.angles_:
            self._ptr_of = (self.ranges_, self.angles_, _dp(self.ranges_), _dp(self.angles_))
        return self._ptr_of[2], self._ptr_of[3]


class PFSampleSet:
    """View of the current sample set copied to the host (particle_filter.h:70-87)."""

    def __init__(self, samples, state):
        self.samples = samples  # [N,4] x, y, theta, weight
        self.sample_count = state.sample_count
        self.converged = state.converged
        self.leaf_count = state.leaf_count


class ParticleFilter:
    """include/amcl/pf/particle_filter.h:92-184 on the GPU-resident set."""

    def __init__(self, engine, min_samples, max_samples, alpha_slow, alpha_fast,
                 global_localization_convergence_threshold, random_pose_fn=None):
        self.e = engine
        self.max_samples = max_samples
        self.min_samples = min_samples
        self.random_pose_fn = random_pose_fn
        engine.check(engine.lib.bpf_pf_create(engine.h, min_samples, max_samples, alpha_slow, alpha_fast,
                                              global_localization_convergence_threshold))

    resample_model = PF_RESAMPLE_MULTINOMIAL

    def setResampleModel(self, model):
        self.resample_model = int(model)
        self._setResampleModel(model)

    def _setResampleModel(self, model):
        self.e.check(self.e.lib.bpf_pf_set_resample_model(self.e.h, model))

    def setRandomPoseGenerator(self, mode):
        """random_pose_fn of the reference's constructor: RANDOM_POSE_FREE_SPACE_2D = Node::randomFreeSpacePose."""
        self.e.check(self.e.lib.bpf_pf_set_random_pose_generator(self.e.h, int(mode)))

    def setPopulationSizeParameters(self, pop_err, pop_z):
        self.e.check(self.e.lib.bpf_pf_set_population_size_parameters(self.e.h, pop_err, pop_z))

    def setDecayRates(self, alpha_slow, alpha_fast):
        self.e.check(self.e.lib.bpf_pf_set_decay_rates(self.e.h, alpha_slow, alpha_fast))

    def srand48(self, seed):
        self.e.check(self.e.lib.bpf_pf_srand48(self.e.h, seed))

    def getRngState(self):
        s = C.c_uint64()
        self.e.check(self.e.lib.bpf_pf_get_rng_state(self.e.h, C.byref(s)))
        return s.value

    def setRngState(self, state):
        self.e.check(self.e.lib.bpf_pf_set_rng_state(self.e.h, state))

    def initWithSamples(self, samples, leaf_count=-1):
        """What initWithPoseFn / initWithGaussian leave behind: poses + weights of the current set."""
        s = np.ascontiguousarray(samples, dtype=np.float64)
        assert s.ndim == 2 and s.shape[1] == 4
        self.e.check(self.e.lib.bpf_pf_set_samples(self.e.h, _dp(s), s.shape[0], leaf_count))

    def initWithGaussian(self, mean, rotation, sigma):
        """ParticleFilter::initWithGaussian given PDFGaussian's decomposition (cr_ row-major 3x3, cd_)."""
        m, r, d = (np.ascontiguousarray(v, dtype=np.float64).reshape(-1) for v in (mean, rotation, sigma))
        assert m.size == 3 and r.size == 9 and d.size == 3
        self.e.check(self.e.lib.bpf_pf_init_with_gaussian(self.e.h, _dp(m), _dp(r), _dp(d)))

    def initWithRandomPoses(self):
        """ParticleFilter::initWithPoseFn with the generator of setRandomPoseGenerator (global localisation)."""
        self.e.check(self.e.lib.bpf_pf_init_with_random_poses(self.e.h))

    def initWithPoseFn(self, pose_fn):
        s = np.zeros((self.max_samples, 4), dtype=np.float64)
        for i in range(self.max_samples):
            s[i, :3] = pose_fn()
        s[:, 3] = 1.0 / self.max_samples
        self.initWithSamples(s)

    def snapshot(self):
        self.e.check(self.e.lib.bpf_pf_snapshot(self.e.h))

    def restore(self):
        self.e.check(self.e.lib.bpf_pf_restore(self.e.h))

    def fillWeights(self, w):
        self.e.check(self.e.lib.bpf_pf_fill_weights(self.e.h, w))

    def updateResample(self):
        self.e.check(self.e.lib.bpf_pf_update_resample(self.e.h))

    def getState(self):
        st = _lib.PFState()
        self.e.check(self.e.lib.bpf_pf_get_state(self.e.h, C.byref(st)))
        return st

    def getCurrentSet(self, out=None):
        """The current set copied to the host; `out` = the caller's own [max_samples, 4] buffer (e.g. the registered
        `samples` storage of a host-side set): the result is then a view of it, not a copy."""
        st = self.getState()
        mine = out is None
        if mine:
            out = np.empty((self.max_samples, 4), dtype=np.float64)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.shape[0] >= st.sample_count
        n = C.c_int()
        self.e.check(self.e.lib.bpf_pf_get_samples(self.e.h, _dp(out), out.shape[0], C.byref(n)))
        return PFSampleSet(out[:n.value].copy() if mine else out[:n.value], st)

    def isConverged(self):
        return bool(self.getState().converged)

    # ---- cluster statistics (particle_filter.cpp:505-660)
    def computeClusterStats(self):
        """Returns (cluster_count, set_mean[3], set_cov[5]); cov entries (0,0) (0,1) (1,0) (1,1) (2,2)."""
        n = C.c_int()
        mean = np.zeros(3)
        cov = np.zeros(5)
        self.e.check(self.e.lib.bpf_pf_compute_cluster_stats(self.e.h, C.byref(n), _dp(mean), _dp(cov)))
        return n.value, mean, cov

    def getClusterStats(self, cidx):
        """ParticleFilter::getClusterStats: (weight, mean) or None when cidx is past the last cluster."""
        c = _lib.Cluster()
        rc = self.e.lib.bpf_pf_get_cluster(self.e.h, int(cidx), C.byref(c))
        if rc == 1:  # BPF_ERR_INVALID_ARGUMENT
            return None
        self.e.check(rc)
        return c.weight, np.array(c.mean[:]), c.count, np.array(c.cov[:])

    def getMaxWeightPose(self):
        """Node2D::getMaxWeightPose (node_2d.cpp:588-617): (max_weight, pose)."""
        w = C.c_double()
        pose = np.zeros(3)
        self.e.check(self.e.lib.bpf_pf_get_max_weight_pose(self.e.h, C.byref(w), _dp(pose)))
        return w.value, pose


ODOM_MODEL_DIFF, ODOM_MODEL_OMNI, ODOM_MODEL_DIFF_CORRECTED, ODOM_MODEL_OMNI_CORRECTED, ODOM_MODEL_GAUSSIAN = range(5)


class OdomData:
    """include/amcl/sensors/odom.h:43-52."""

    def __init__(self, pose, delta, absolute_motion=None):
        self.pose = np.ascontiguousarray(pose, dtype=np.float64)
        self.delta = np.ascontiguousarray(delta, dtype=np.float64)
        # Node::updateOdom passes the delta when the odometry integrator is off (node.cpp:1083-1087)
        self.absolute_motion = np.ascontiguousarray(delta if absolute_motion is None else absolute_motion,
                                                    dtype=np.float64)


class Odom:
    """Odom sensor (src/amcl/sensors/odom.cpp): setModel + updateAction on the resident set."""

    def __init__(self, engine):
        self.e = engine

    def setModel(self, model_type, alpha1, alpha2, alpha3, alpha4, alpha5=0.0):
        self.e.check(self.e.lib.bpf_odom_set_model(self.e.h, int(model_type), alpha1, alpha2, alpha3, alpha4, alpha5))

    def updateAction(self, pf, data):
        self.e.check(self.e.lib.bpf_pf_update_action(self.e.h, _dp(data.pose), _dp(data.delta),
                                                     _dp(data.absolute_motion)))
        return True

    def updateActionShard(self, data, global_first, global_count):
        self.e.check(self.e.lib.bpf_shard_update_action(self.e.h, _dp(data.pose), _dp(data.delta),
                                                        _dp(data.absolute_motion), int(global_first),
                                                        int(global_count)))
        return True


class PlanarScanner:
    """include/amcl/sensors/planar_scanner.h:57-168"""

    def __init__(self, engine):
        self.e = engine
        self.max_beams = 0
        self.map = None
        engine.check(engine.lib.bpf_planar_set_map_factors(engine.h, 1.0, 1.0, 0.0))

    def init(self, max_beams, occupancy_map):
        self.max_beams = max_beams
        self.map = occupancy_map
        occupancy_map.upload()
        self.e.check(self.e.lib.bpf_planar_init(self.e.h, max_beams))

    def setModelBeam(self, z_hit, z_short, z_max, z_rand, sigma_hit, lambda_short):
        self.map.upload()
        self.e.check(self.e.lib.bpf_planar_set_model_beam(self.e.h, z_hit, z_short, z_max, z_rand, sigma_hit,
                                                          lambda_short))

    def setModelLikelihoodField(self, z_hit, z_rand, sigma_hit, max_distance_to_object):
        self.map.upload()
        self.e.check(self.e.lib.bpf_planar_set_model_likelihood_field(self.e.h, z_hit, z_rand, sigma_hit,
                                                                      max_distance_to_object))

    def setModelLikelihoodFieldProb(self, z_hit, z_rand, sigma_hit, max_distance_to_object, do_beamskip,
                                    beam_skip_distance, beam_skip_threshold, beam_skip_error_threshold):
        self.map.upload()
        self.e.check(self.e.lib.bpf_planar_set_model_likelihood_field_prob(
            self.e.h, z_hit, z_rand, sigma_hit, max_distance_to_object, int(do_beamskip), beam_skip_distance,
            beam_skip_threshold, beam_skip_error_threshold))

    def setModelLikelihoodFieldGompertz(self, z_hit, z_rand, sigma_hit, max_distance_to_object, gompertz_a,
                                        gompertz_b, gompertz_c, input_shift, input_scale, output_shift):
        self.map.upload()
        self.e.check(self.e.lib.bpf_planar_set_model_likelihood_field_gompertz(
            self.e.h, z_hit, z_rand, sigma_hit, max_distance_to_object, gompertz_a, gompertz_b, gompertz_c,
            input_shift, input_scale, output_shift))

    def setMapFactors(self, off_map_factor, non_free_space_factor, non_free_space_radius):
        self.e.check(self.e.lib.bpf_planar_set_map_factors(self.e.h, off_map_factor, non_free_space_factor,
                                                           non_free_space_radius))

    def setPlanarScannerPose(self, scanner_pose):
        p = np.ascontiguousarray(scanner_pose, dtype=np.float64)
        self.e.check(self.e.lib.bpf_planar_set_scanner_pose(self.e.h, _dp(p)))

    def updateSensor(self, pf, data):
        """PlanarScanner::updateSensor(pf, data): false (and no effect) when max_beams < 2."""
        if self.max_beams < 2:
            return False
        rp, ap = data.pointers()
        self.e.check(self.e.lib.bpf_pf_update_sensor_planar(self.e.h, rp, ap, data.range_count_, data.range_max_))
        return True

    def applyModelToSampleSet(self, data, samples, set_converged=0):
        """Seam A with host buffers: samples [N,4] float64, weights multiplied in place; returns total."""
        assert samples.dtype == np.float64 and samples.flags.c_contiguous
        status = C.c_int(0)
        total = self.e.lib.bpf_planar_apply_model_to_sample_set(
            self.e.h, _dp(samples), samples.shape[0], int(set_converged), _dp(data.ranges_), _dp(data.angles_),
            data.range_count_, data.range_max_, C.byref(status))
        if status.value != 0:
            self.e.check(status.value)
        return total


# --------------------------------------------------------------------------------- 3-D
class OctoMap:
    """LUT state of the reference OctoMap (include/amcl/map/octomap.h:96-110): pose_indices_,
    distance_ratios_, cropped min/max cells.  Building it from an octree is the caller's job."""

    def __init__(self, engine, resolution):
        self.e = engine
        self.resolution = float(resolution)

    def setDistancesLUT(self, pose_indices, distance_ratios, min_cells, max_cells, max_distance_to_object):
        pi = np.ascontiguousarray(pose_indices, dtype=np.uint32)
        dr = np.ascontiguousarray(distance_ratios, dtype=np.uint8)
        mn = np.ascontiguousarray(min_cells, dtype=np.int32)
        mx = np.ascontiguousarray(max_cells, dtype=np.int32)
        self.max_distance_to_object = float(max_distance_to_object)
        self.e.check(self.e.lib.bpf_map3d_set(self.e.h, pi.ctypes.data_as(C.POINTER(C.c_uint32)), pi.size,
                                              dr.ctypes.data_as(C.POINTER(C.c_uint8)), dr.size,
                                              mn.ctypes.data_as(C.POINTER(C.c_int)),
                                              mx.ctypes.data_as(C.POINTER(C.c_int)), self.resolution,
                                              self.max_distance_to_object))

    def updateDistancesLUT(self, occupied_ijk, min_cells, max_cells, max_distance_to_object):
        """OctoMap::updateDistancesLUT (octomap.cpp:175-333) from the occupied voxel indices (octree leaf order)."""
        occ = np.ascontiguousarray(occupied_ijk, dtype=np.int32).reshape(-1, 3)
        mn = np.ascontiguousarray(min_cells, dtype=np.int32)
        mx = np.ascontiguousarray(max_cells, dtype=np.int32)
        self.max_distance_to_object = float(max_distance_to_object)
        self.e.check(self.e.lib.bpf_map3d_build_distances_lut(
            self.e.h, occ.ctypes.data_as(C.POINTER(C.c_int)), occ.shape[0], mn.ctypes.data_as(C.POINTER(C.c_int)),
            mx.ctypes.data_as(C.POINTER(C.c_int)), self.resolution, self.max_distance_to_object))

    def getDistancesLUT(self):
        """(pose_indices uint32, distance_ratios uint8) as held on the device."""
        npi, ndr = C.c_size_t(), C.c_size_t()
        self.e.check(self.e.lib.bpf_map3d_get_distances_lut(self.e.h, None, 0, C.byref(npi), None, 0, C.byref(ndr)))
        pi = np.zeros(npi.value, dtype=np.uint32)
        dr = np.zeros(ndr.value, dtype=np.uint8)
        self.e.check(self.e.lib.bpf_map3d_get_distances_lut(
            self.e.h, pi.ctypes.data_as(C.POINTER(C.c_uint32)), pi.size, None,
            dr.ctypes.data_as(C.POINTER(C.c_uint8)), dr.size, None))
        return pi, dr


class PointCloudData:
    """include/amcl/sensors/point_cloud_scanner.h:45-51 (points_ as packed float xyz)"""

    def __init__(self, points_xyz):
        self.points_ = np.ascontiguousarray(points_xyz, dtype=np.float32).reshape(-1, 3)


class PointCloudScanner:
    """include/amcl/sensors/point_cloud_scanner.h:53-120"""

    def __init__(self, engine):
        self.e = engine
        self.max_beams = 0

    def init(self, max_beams, octomap):
        self.max_beams = max_beams
        self.map = octomap
        self.e.check(self.e.lib.bpf_cloud_init(self.e.h, max_beams))

    def setPointCloudModel(self, z_hit, z_rand, sigma_hit):
        self.e.check(self.e.lib.bpf_cloud_set_model(self.e.h, z_hit, z_rand, sigma_hit))

    def setPointCloudModelGompertz(self, z_hit, z_rand, sigma_hit, gompertz_a, gompertz_b, gompertz_c, input_shift,
                                   input_scale, output_shift):
        self.e.check(self.e.lib.bpf_cloud_set_model_gompertz(self.e.h, z_hit, z_rand, sigma_hit, gompertz_a,
                                                             gompertz_b, gompertz_c, input_shift, input_scale,
                                                             output_shift))

    def setMapFactors(self, off_map_factor, non_free_space_factor, non_free_space_radius):
        self.e.check(self.e.lib.bpf_cloud_set_map_factors(self.e.h, off_map_factor, non_free_space_factor,
                                                          non_free_space_radius))

    def setPointCloudScannerToFootprintTF(self, xyz, quat_xyzw):
        t = np.ascontiguousarray(xyz, dtype=np.float64)
        q = np.ascontiguousarray(quat_xyzw, dtype=np.float64)
        self.e.check(self.e.lib.bpf_cloud_set_scanner_to_footprint_tf(self.e.h, _dp(t), _dp(q)))

    def updateSensor(self, pf, data):
        if self.max_beams < 2:
            return False
        pts = data.points_
        self.e.check(self.e.lib.bpf_pf_update_sensor_cloud(self.e.h, pts.ctypes.data_as(C.POINTER(C.c_float)),
                                                           pts.shape[0]))
        return True

    def applyModelToSampleSet(self, data, samples):
        assert samples.dtype == np.float64 and samples.flags.c_contiguous
        status = C.c_int(0)
        pts = data.points_
        total = self.e.lib.bpf_cloud_apply_model_to_sample_set(self.e.h, _dp(samples), samples.shape[0],
                                                               pts.ctypes.data_as(C.POINTER(C.c_float)),
                                                               pts.shape[0], C.byref(status))
        if status.value != 0:
            self.e.check(status.value)
        return total
