"""ctypes binding of include/badger_pf.h.  Loading fails loudly when the HIP library is
missing; nothing here computes on the CPU."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("BPF_LIB") or os.path.join(HERE, "libbadger_pf_hip.so")  # BPF_LIB: experiment builds

BPF_K_COUNT = 9


class PFState(C.Structure):
    _fields_ = [("sample_count", C.c_int), ("leaf_count", C.c_int), ("bin_count", C.c_int),
                ("converged", C.c_int), ("percent_converged", C.c_float),
                ("total", C.c_double), ("w_slow", C.c_double), ("w_fast", C.c_double), ("w_diff", C.c_double),
                ("last_status", C.c_int), ("resample_windows", C.c_int), ("evals", C.c_longlong),
                ("kld_on_device", C.c_int), ("reserved", C.c_int)]


class Cluster(C.Structure):
    _fields_ = [("count", C.c_int), ("weight", C.c_double), ("mean", C.c_double * 3), ("cov", C.c_double * 5)]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * BPF_K_COUNT), ("launches", C.c_longlong * BPF_K_COUNT)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_vp = C.c_void_p

# name -> (restype, argtypes); every symbol include/badger_pf.h declares
SIGNATURES = {
    "bpf_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "bpf_destroy": (None, [_vp]),
    "bpf_error_string": (C.c_char_p, [C.c_int]),
    "bpf_last_error_message": (C.c_char_p, [_vp]),
    "bpf_set_stream": (C.c_int, [_vp, _vp]),
    "bpf_synchronize": (C.c_int, [_vp]),
    "bpf_map2d_set": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_float,
                                C.c_float, C.c_double, C.c_double]),
    "bpf_map2d_build_distances_lut": (C.c_int, [_vp, C.c_double]),
    "bpf_map2d_get_distances_lut": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_size_t]),
    "bpf_map2d_calc_range": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, C.c_int, _dp]),
    "bpf_planar_init": (C.c_int, [_vp, C.c_int]),
    "bpf_planar_set_model_beam": (C.c_int, [_vp] + [C.c_double] * 6),
    "bpf_planar_set_model_likelihood_field": (C.c_int, [_vp] + [C.c_double] * 4),
    "bpf_planar_set_model_likelihood_field_prob": (C.c_int, [_vp] + [C.c_double] * 4 + [C.c_int] + [C.c_double] * 3),
    "bpf_planar_set_model_likelihood_field_gompertz": (C.c_int, [_vp] + [C.c_double] * 10),
    "bpf_planar_set_map_factors": (C.c_int, [_vp] + [C.c_double] * 3),
    "bpf_planar_set_scanner_pose": (C.c_int, [_vp, _dp]),
    "bpf_planar_apply_model_to_sample_set": (C.c_double, [_vp, _dp, C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_double,
                                                          _ip]),
    "bpf_pf_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]),
    "bpf_pf_set_resample_model": (C.c_int, [_vp, C.c_int]),
    "bpf_pf_set_population_size_parameters": (C.c_int, [_vp, C.c_double, C.c_double]),
    "bpf_pf_set_decay_rates": (C.c_int, [_vp, C.c_double, C.c_double]),
    "bpf_pf_srand48": (C.c_int, [_vp, C.c_long]),
    "bpf_pf_set_rng_state": (C.c_int, [_vp, C.c_uint64]),
    "bpf_pf_get_rng_state": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "bpf_pf_set_samples": (C.c_int, [_vp, _dp, C.c_int, C.c_int]),
    "bpf_pf_get_samples": (C.c_int, [_vp, _dp, C.c_int, _ip]),
    "bpf_pf_snapshot": (C.c_int, [_vp]),
    "bpf_pf_restore": (C.c_int, [_vp]),
    "bpf_pf_fill_weights": (C.c_int, [_vp, C.c_double]),
    "bpf_pf_update_sensor_planar": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double]),
    "bpf_pf_set_random_pose_generator": (C.c_int, [_vp, C.c_int]),
    "bpf_pf_update_resample": (C.c_int, [_vp]),
    "bpf_set_option": (C.c_int, [_vp, C.c_int, C.c_int]),
    "bpf_get_cells_walked": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.c_int]),
    "bpf_pf_get_state": (C.c_int, [_vp, C.POINTER(PFState)]),
    "bpf_pf_init_with_gaussian": (C.c_int, [_vp, _dp, _dp, _dp]),
    "bpf_pf_init_with_random_poses": (C.c_int, [_vp]),
    "bpf_odom_set_model": (C.c_int, [_vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "bpf_pf_update_action": (C.c_int, [_vp, _dp, _dp, _dp]),
    "bpf_shard_update_action": (C.c_int, [_vp, _dp, _dp, _dp, C.c_longlong, C.c_longlong]),
    "bpf_pf_compute_cluster_stats": (C.c_int, [_vp, _ip, _dp, _dp]),
    "bpf_pf_get_cluster": (C.c_int, [_vp, C.c_int, C.POINTER(Cluster)]),
    "bpf_pf_get_max_weight_pose": (C.c_int, [_vp, _dp, _dp]),
    "bpf_map2d_build_distances_lut_reference": (C.c_int, [_vp, C.c_double]),
    "bpf_host_buffer_register": (C.c_int, [_vp, C.c_void_p, C.c_size_t]),
    "bpf_host_buffer_unregister": (C.c_int, [_vp, C.c_void_p]),
    "bpf_host_buffer_is_registered": (C.c_int, [_vp, C.c_void_p, C.c_size_t]),
    "bpf_seam_last_plan": (C.c_int, [_vp, _ip, _ip]),
    "bpf_kld_last_form": (C.c_int, [_vp, _ip]),
    "bpf_score_last_form": (C.c_int, [_vp, _ip]),
    "bpf_wire_laserscan_to_planar": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, C.c_double,
                                               C.c_double, C.c_double, C.c_double, _dp, _dp, _dp]),
    "bpf_wire_scan_angle_stats": (C.c_int, [C.c_double, C.c_double, _dp, _dp, _dp]),
    "bpf_wire_occupancy_grid_to_cells": (C.c_int, [C.POINTER(C.c_int8), C.c_int, C.c_int, C.c_double, C.c_double,
                                                   C.c_double, C.c_int, C.POINTER(C.c_int32), _ip, _ip,
                                                   C.POINTER(C.c_float), _dp]),
    "bpf_wire_decimate_cloud": (C.c_int, [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]),
    "bpf_wire_samples_to_pose_array": (C.c_int, [_dp, C.c_int, _dp]),
    "bpf_map3d_set": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_uint8), C.c_size_t, _ip, _ip,
                                C.c_double, C.c_double]),
    "bpf_map3d_build_distances_lut": (C.c_int, [_vp, _ip, C.c_size_t, _ip, _ip, C.c_double, C.c_double]),
    "bpf_map3d_get_distances_lut": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(C.c_size_t),
                                              C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)]),
    "bpf_cloud_init": (C.c_int, [_vp, C.c_int]),
    "bpf_cloud_set_model": (C.c_int, [_vp] + [C.c_double] * 3),
    "bpf_cloud_set_model_gompertz": (C.c_int, [_vp] + [C.c_double] * 9),
    "bpf_cloud_set_map_factors": (C.c_int, [_vp] + [C.c_double] * 3),
    "bpf_cloud_set_scanner_to_footprint_tf": (C.c_int, [_vp, _dp, _dp]),
    "bpf_cloud_apply_model_to_sample_set": (C.c_double, [_vp, _dp, C.c_int, C.POINTER(C.c_float), C.c_int, _ip]),
    "bpf_pf_update_sensor_cloud": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int]),
    "bpf_shard_score_planar": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double]),
    "bpf_shard_beam_counts_dev": (C.c_int, [_vp, C.POINTER(_vp), _ip]),
    "bpf_shard_score_planar_finish": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_longlong]),
    "bpf_shard_score_cloud": (C.c_int, [_vp, C.POINTER(C.c_float), C.c_int]),
    "bpf_shard_scalars_dev": (C.c_int, [_vp, C.POINTER(_vp)]),
    "bpf_shard_normalize_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "bpf_shard_build_cdf": (C.c_int, [_vp, _vp]),
    "bpf_shard_draw_window_dev": (C.c_int, [_vp, C.c_uint64, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp,
                                            C.c_int, _vp]),
    "bpf_shard_tail_small_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bpf_shard_adopt_dev": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "bpf_shard_converged_dev": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "bpf_shard_mailbox_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_longlong, _vp]),
    "bpf_shard_mailbox_connect": (C.c_int, [_vp, _vp]),
    "bpf_shard_mailbox_selftest": (C.c_int, [_vp, C.c_int]),
    "bpf_map3d_builder_generations": (C.c_int, [_vp, _ip]),
    "bpf_shard_mailbox_set_timeout_ms": (C.c_int, [_vp, C.c_int]),
    "bpf_shard_mailbox_error_stage": (C.c_int, [_vp, _ip, _ip]),
    "bpf_shard_bootstrap": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, C.c_longlong, C.c_int, _ip]),
    "bpf_shard_shutdown": (C.c_int, [_vp]),
    "bpf_shard_update_sensor_planar": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_longlong]),
    "bpf_shard_update_resample": (C.c_int, [_vp, _ip, _ip, _ip, _ip, _ip, _ip]),
    "bpf_shard_mailbox_update_sensor_planar": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_double, C.c_longlong]),
    "bpf_shard_mailbox_update_resample": (C.c_int, [_vp, _vp, _ip, _ip, _ip, _ip, _ip]),
    "bpf_shard_mailbox_destroy": (C.c_int, [_vp]),
    "bpf_shard_mailbox_totals": (C.c_int, [_vp, C.POINTER(_vp)]),
    "bpf_shard_mailbox_window": (C.c_int, [_vp, C.POINTER(_vp), _ip]),
    "bpf_drand48_skip": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "bpf_kld_reset": (C.c_int, [_vp]),
    "bpf_kld_feed": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _ip]),
    "bpf_kld_feed_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _ip]),
    "bpf_shard_begin_resample": (C.c_int, [_vp, C.c_uint64, C.c_int, _dp, _ip]),
    "bpf_shard_end_resample": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint64)]),
    "bpf_pf_resample_limit": (C.c_int, [_vp, C.c_int, _ip]),
    "bpf_shard_systematic_window_dev": (C.c_int, [_vp, C.c_uint64, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp,
                                                  C.c_int, _vp]),
    "bpf_kld_insert": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int]),
    "bpf_kld_insert_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int]),
    "bpf_kld_stop_dev": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _ip, _ip, _ip, _ip]),
    "bpf_kld_leaf_count": (C.c_int, [_vp, _ip, _ip]),
    "bpf_device_memory_info": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "bpf_profile_enable": (C.c_int, [_vp, C.c_int]),
    "bpf_profile_reset": (C.c_int, [_vp]),
    "bpf_profile_get": (C.c_int, [_vp, C.POINTER(Profile)]),
    "bpf_score_kernel_name": (C.c_char_p, [_vp]),
    "bpf_get_window_plan": (C.c_int, [_vp, _ip, _ip, _ip]),
}

_lib = None


def load():
    """Returns the loaded library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise RuntimeError(
            "badger_amcl_amd: %s is missing. Build it with `python -m badger_amcl_amd.build` "
            "(needs hipcc). There is no CPU fallback for the sensor-update/resample path." % SO)
    L = C.CDLL(SO)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch, on purpose
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
