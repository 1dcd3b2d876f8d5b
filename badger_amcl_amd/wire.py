"""Wire-format shaping either side of the hot path: what Node2D /
Node3D / Node do to a message before the sensor update and to the sample set after it.  Thin ctypes
marshalling over the bpf_wire_* host functions of include/badger_pf.h; no ROS types here -- the
caller hands over the message fields.
"""
import ctypes as C

import numpy as np

from . import _lib


def _check(rc):
    if rc != 0:
        raise ValueError("bpf_wire call failed with code %d" % rc)


def laserscan_to_planar(ranges_f32, msg_range_min, msg_range_max, angle_min, angle_increment,
                        sensor_min_range=-1.0, sensor_max_range=-1.0):
    """Node2D::updateLatestScanData (node_2d.cpp:531-560) -> (ranges f64, angles f64, range_max)."""
    r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
    n = r.size
    ro, ao = np.zeros(n), np.zeros(n)
    rm = C.c_double()
    _check(_lib.load().bpf_wire_laserscan_to_planar(
        r.ctypes.data_as(C.POINTER(C.c_float)), n, msg_range_min, msg_range_max, sensor_min_range, sensor_max_range,
        angle_min, angle_increment, ro.ctypes.data_as(C.POINTER(C.c_double)),
        ao.ctypes.data_as(C.POINTER(C.c_double)), C.byref(rm)))
    return ro, ao, rm.value


def scan_angle_stats(msg_angle_min, msg_angle_increment, q_base_from_scanner_xyzw):
    """Node2D::getAngleStats (node_2d.cpp:497-529) -> (angle_min, angle_increment) in the base frame."""
    q = np.ascontiguousarray(q_base_from_scanner_xyzw, dtype=np.float64)
    a, b = C.c_double(), C.c_double()
    _check(_lib.load().bpf_wire_scan_angle_stats(msg_angle_min, msg_angle_increment,
                                                 q.ctypes.data_as(C.POINTER(C.c_double)), C.byref(a), C.byref(b)))
    return a.value, b.value


def occupancy_grid_to_cells(data_i8, width, height, resolution, origin_x, origin_y, map_scale_up_factor=1):
    """Node2D::convertMap (node_2d.cpp:265-295) -> (cells int32 [size_y, size_x], origin f32[2], resolution)."""
    d = np.ascontiguousarray(data_i8, dtype=np.int8).reshape(-1)
    if d.size != width * height:
        raise ValueError("data size does not match width * height")
    f = int(map_scale_up_factor)
    cells = np.zeros((height * f, width * f), dtype=np.int32)
    sx, sy = C.c_int(), C.c_int()
    origin = np.zeros(2, dtype=np.float32)
    res = C.c_double()
    _check(_lib.load().bpf_wire_occupancy_grid_to_cells(
        d.ctypes.data_as(C.POINTER(C.c_int8)), width, height, resolution, origin_x, origin_y, f,
        cells.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(sx), C.byref(sy),
        origin.ctypes.data_as(C.POINTER(C.c_float)), C.byref(res)))
    assert (sy.value, sx.value) == cells.shape
    return cells, origin, res.value


def decimate_cloud(points_xyz_f32, max_beams):
    """Node3D::updateLatestScanData (node_3d.cpp:467-480): every step-th point."""
    p = np.ascontiguousarray(points_xyz_f32, dtype=np.float32).reshape(-1, 3)
    out = np.zeros_like(p)
    k = _lib.load().bpf_wire_decimate_cloud(p.ctypes.data_as(C.POINTER(C.c_float)), p.shape[0], int(max_beams),
                                            out.ctypes.data_as(C.POINTER(C.c_float)), out.shape[0])
    if k < 0:
        raise ValueError("bpf_wire_decimate_cloud: bad arguments")
    return out[:k].copy()


def samples_to_pose_array(samples):
    """Node::publishParticleCloud (node.cpp:335-357): [n,4] samples -> [n,7] (x y z qx qy qz qw)."""
    s = np.ascontiguousarray(samples, dtype=np.float64).reshape(-1, 4)
    out = np.zeros((s.shape[0], 7))
    _check(_lib.load().bpf_wire_samples_to_pose_array(s.ctypes.data_as(C.POINTER(C.c_double)), s.shape[0],
                                                      out.ctypes.data_as(C.POINTER(C.c_double))))
    return out
