"""ctypes binding of include/localization_amd.h. Fails loudly when the HIP library is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (LOCALIZATION_AMD_LIB: an alternative build of the same library, for experiments such as the diagnostic timing build)
_SO = os.environ.get("LOCALIZATION_AMD_LIB") or os.path.join(_HERE, "liblocalization_amd.so")
_LIB = None

LOC_OK = 0
LOC_ERR_NO_DEVICE = -2
JAC_ANALYTIC = 0
JAC_NUMERIC_G2O = 1


class LocalizationAmdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"localization_amd error {code}: {msg}")
        self.code = code


class SnapshotParams(C.Structure):
    _fields_ = [("maximum_iteration", C.c_int32), ("distance_outlier", C.c_double), ("gate_warmup_epochs", C.c_int32), ("jacobian", C.c_int32),
                ("lanes_per_instance", C.c_int32), ("block_threads", C.c_int32)]


def library_path():
    return _SO


# every symbol include/localization_amd.h declares (tests check the .so exports all of them)
EXPORTED_SYMBOLS = [
    "loc_last_error", "loc_abi_version", "loc_device_count", "loc_shard_bounds", "loc_shard_plan",
    "loc_snapshot_default_params", "loc_snapshot_create", "loc_snapshot_destroy", "loc_snapshot_batch",
    "loc_snapshot_anchor_groups", "loc_snapshot_lanes_per_instance", "loc_snapshot_range_floats",
    "loc_snapshot_set_positions", "loc_snapshot_get_positions", "loc_snapshot_positions_device",
    "loc_snapshot_epochs_done", "loc_snapshot_set_epochs_done",
    "loc_snapshot_pack_ranges_host", "loc_snapshot_solve_device", "loc_snapshot_solve_host",
    "loc_snapshot_solve_host_kmb", "loc_host_alloc", "loc_host_free",
    "loc_snapshot_timing_begin", "loc_snapshot_timing_end",
    "loc_window_create", "loc_window_destroy", "loc_window_set_anchors", "loc_window_lds_bytes", "loc_window_solve_host",
    "loc_window_last_kernel_ms", "loc_window_set_endpoint1_offsets", "loc_window_set_jacobian", "loc_window_set_ordering", "loc_window_set_chain_threshold", "loc_window_last_kernel_kind", "loc_window_set_option", "loc_window_last_host_timing", "loc_window_upload",
    "loc_window_solve_resident", "loc_window_download", "loc_window_poses_device", "loc_window_result_device",
    "loc_window_timing_begin", "loc_window_timing_end",
    "loc_node_default_config", "loc_node_create", "loc_node_destroy", "loc_node_add_range", "loc_node_add_imu",
    "loc_node_add_pose", "loc_node_add_twist", "loc_node_add_lidar", "loc_node_add_rl_range", "loc_node_solve", "loc_node_get_path",
    "loc_node_number_measurements", "loc_node_last_timing", "loc_node_last_kernel_kind", "loc_node_flush_tail", "loc_node_set_deferred", "loc_node_solve_pending", "loc_nodes_solve_batch",
    "loc_nodes_release_batch_cache",
    "loc_fusion_default_params", "loc_fusion_create", "loc_fusion_destroy", "loc_fusion_set_poses", "loc_fusion_get_poses",
    "loc_fusion_solve_device", "loc_fusion_solve_host", "loc_fusion_solve_host_kmb", "loc_fusion_last_kernel_ms", "loc_fusion_timing_begin", "loc_fusion_timing_end",
]


def lib():
    """Load liblocalization_amd.so (built by __graft_entry__.build() / make -C localization_amd/csrc)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(_SO):
        raise LocalizationAmdError(-100, f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; "
                                   "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(_SO)
    vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.loc_last_error.restype = C.c_char_p
    L.loc_abi_version.restype = C.c_int32
    L.loc_device_count.restype = C.c_int32
    L.loc_shard_bounds.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.loc_shard_plan.argtypes = [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
    L.loc_snapshot_default_params.argtypes = [C.POINTER(SnapshotParams)]
    L.loc_snapshot_default_params.restype = None
    L.loc_snapshot_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int32, dp, C.POINTER(SnapshotParams)]
    L.loc_snapshot_destroy.argtypes = [vp]
    L.loc_snapshot_batch.argtypes = [vp]; L.loc_snapshot_batch.restype = C.c_int64
    L.loc_snapshot_anchor_groups.argtypes = [vp]; L.loc_snapshot_anchor_groups.restype = C.c_int32
    L.loc_snapshot_lanes_per_instance.argtypes = [vp]; L.loc_snapshot_lanes_per_instance.restype = C.c_int32
    L.loc_snapshot_range_floats.argtypes = [vp, C.c_int32]; L.loc_snapshot_range_floats.restype = C.c_size_t
    L.loc_snapshot_set_positions.argtypes = [vp, dp]
    L.loc_snapshot_get_positions.argtypes = [vp, dp]
    L.loc_snapshot_positions_device.argtypes = [vp]; L.loc_snapshot_positions_device.restype = vp
    L.loc_snapshot_epochs_done.argtypes = [vp]; L.loc_snapshot_epochs_done.restype = C.c_int64
    L.loc_snapshot_set_epochs_done.argtypes = [vp, C.c_int64]
    L.loc_snapshot_pack_ranges_host.argtypes = [vp, C.c_int32, fp, fp, C.c_float]
    L.loc_snapshot_solve_device.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    L.loc_snapshot_solve_host.argtypes = [vp, C.c_int32, fp, fp, dp, dp, C.POINTER(C.c_uint8)]
    L.loc_snapshot_solve_host_kmb.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp]
    L.loc_host_alloc.argtypes = [C.POINTER(vp), C.c_size_t]
    L.loc_host_free.argtypes = [vp]
    L.loc_snapshot_timing_begin.argtypes = [vp, C.c_int32]
    L.loc_snapshot_timing_end.argtypes = [vp, C.POINTER(C.c_int32), dp, dp]
    ip = C.POINTER(C.c_int32)
    L.loc_window_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, vp, C.c_int32, dp, C.c_int32]
    L.loc_window_destroy.argtypes = [vp]
    L.loc_window_set_anchors.argtypes = [vp, C.c_int32, dp]
    L.loc_window_lds_bytes.argtypes = [vp]; L.loc_window_lds_bytes.restype = C.c_size_t
    L.loc_window_solve_host.argtypes = [vp, C.c_int64, ip, dp, ip, dp, ip, dp, ip, dp, dp]
    L.loc_window_last_kernel_ms.argtypes = [vp, dp]
    L.loc_window_set_jacobian.argtypes = [vp, C.c_int32]
    L.loc_window_set_endpoint1_offsets.argtypes = [vp, C.c_int64, dp]
    L.loc_window_set_ordering.argtypes = [vp, C.c_int32]
    L.loc_window_set_chain_threshold.argtypes = [vp, C.c_int64]
    L.loc_window_last_kernel_kind.argtypes = [vp, ip]
    L.loc_window_upload.argtypes = [vp, C.c_int64, ip, dp, ip, dp, ip, dp, ip, dp]
    L.loc_window_solve_resident.argtypes = [vp, vp]
    L.loc_window_download.argtypes = [vp, dp, dp]
    L.loc_window_poses_device.argtypes = [vp]; L.loc_window_poses_device.restype = vp
    L.loc_window_result_device.argtypes = [vp]; L.loc_window_result_device.restype = vp
    L.loc_window_timing_begin.argtypes = [vp, C.c_int32]
    L.loc_window_timing_end.argtypes = [vp, ip, dp, dp]
    _LIB = L
    return L


def check(rc):
    if rc != LOC_OK:
        raise LocalizationAmdError(rc, lib().loc_last_error().decode(errors="replace"))


def abi_version():
    return lib().loc_abi_version()


def device_count():
    return lib().loc_device_count()
