"""`class Localization` behind the C ABI: Python harness over loc_node_* (include/localization_amd.h)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class NodeConfig(C.Structure):
    _fields_ = [("trajectory_length", C.c_int32), ("maximum_velocity", C.c_double), ("distance_outlier", C.c_double),
                ("maximum_iteration", C.c_int32), ("minimum_optimize_error", C.c_double),
                ("publish_range", C.c_int32), ("publish_pose", C.c_int32), ("publish_twist", C.c_int32),
                ("publish_lidar", C.c_int32), ("publish_imu", C.c_int32), ("has_relative_range", C.c_int32), ("jacobian", C.c_int32),
                ("publish_relative_range", C.c_int32)]


class NodeOutput(C.Structure):
    _fields_ = [("solved", C.c_int32), ("published", C.c_int32), ("chi2", C.c_double),
                ("realtime", C.c_double * 8), ("optimized", C.c_double * 8),
                ("outer_iterations", C.c_int32), ("lm_trials", C.c_int32)]


def _bind(L):
    if getattr(L, "_node_bound", False):
        return
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    po = C.POINTER(NodeOutput)
    L.loc_node_default_config.argtypes = [C.POINTER(NodeConfig)]; L.loc_node_default_config.restype = None
    L.loc_node_create.argtypes = [C.POINTER(vp), C.c_int32, C.POINTER(NodeConfig), C.c_int32, C.POINTER(C.c_int32), dp, C.c_int32, dp]
    L.loc_node_destroy.argtypes = [vp]
    L.loc_node_add_range.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, C.c_float, C.c_float, C.c_int32, C.c_char_p, po]
    L.loc_node_add_imu.argtypes = [vp, C.c_double, dp, dp, C.c_char_p, po]
    L.loc_node_add_pose.argtypes = [vp, C.c_double, dp, dp, C.c_char_p, po]
    L.loc_node_add_twist.argtypes = [vp, C.c_double, dp, dp, C.c_char_p, po]
    L.loc_node_add_lidar.argtypes = [vp, C.c_double, C.c_double, C.c_char_p, po]
    L.loc_node_add_rl_range.argtypes = [vp, C.c_int32, C.c_int32, C.c_double, C.c_double, dp, po]
    L.loc_node_solve.argtypes = [vp, po]
    L.loc_node_get_path.argtypes = [vp, C.c_int32, dp, C.c_int32]
    L.loc_node_number_measurements.argtypes = [vp]; L.loc_node_number_measurements.restype = C.c_int32
    L.loc_node_last_timing.argtypes = [vp, C.POINTER(C.c_double)]
    L.loc_node_last_kernel_kind.argtypes = [vp, C.POINTER(C.c_int32)]
    L.loc_node_flush_tail.argtypes = [vp, dp, C.c_int32]
    L.loc_node_set_deferred.argtypes = [vp, C.c_int32]
    L.loc_node_solve_pending.argtypes = [vp]; L.loc_node_solve_pending.restype = C.c_int32
    L.loc_nodes_solve_batch.argtypes = [C.POINTER(vp), C.c_int32, po]
    L.loc_nodes_release_batch_cache.argtypes = []
    L._node_bound = True


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def _out(o, rc):
    return dict(rc=rc, solved=bool(o.solved), published=bool(o.published), chi2=o.chi2,
                realtime=np.array(o.realtime[:]), optimized=np.array(o.optimized[:]),
                outer_iterations=o.outer_iterations, lm_trials=o.lm_trials)


class LocalizationNode:
    """Same constructor surface as the reference node's parameters (cfg/*.yaml + /uwb/*)."""

    def __init__(self, nodes_id, nodes_pos, trajectory_length, maximum_velocity=1.0, distance_outlier=1.0,
                 maximum_iteration=20, minimum_optimize_error=1000.0, publish_range=False, publish_pose=False,
                 publish_twist=False, publish_lidar=False, publish_imu=False, has_relative_range=False,
                 antenna_offsets=None, device=0, jacobian="numeric", publish_relative_range=False):
        L = lib(); _bind(L)
        self.L = L
        cfg = NodeConfig(int(trajectory_length), float(maximum_velocity), float(distance_outlier), int(maximum_iteration),
                         float(minimum_optimize_error), int(publish_range), int(publish_pose), int(publish_twist),
                         int(publish_lidar), int(publish_imu), int(has_relative_range),
                         _lib.JAC_NUMERIC_G2O if jacobian in ("numeric", _lib.JAC_NUMERIC_G2O) else _lib.JAC_ANALYTIC,
                         int(publish_relative_range))
        ids = (C.c_int32 * len(nodes_id))(*[int(i) for i in nodes_id])
        pos = np.ascontiguousarray(nodes_pos, dtype=np.float64).reshape(-1)
        assert pos.size == 3 * len(nodes_id)
        ant, n_ant = None, 0
        if antenna_offsets is not None:
            ant = np.ascontiguousarray(antenna_offsets, dtype=np.float64).reshape(-1); n_ant = ant.size // 3
        h = C.c_void_p()
        check(L.loc_node_create(C.byref(h), device, C.byref(cfg), len(nodes_id), ids, _dp(pos), n_ant, _dp(ant)))
        self.h = h
        self.T = int(trajectory_length)

    @classmethod
    def from_config(cls, cfg, device=0, jacobian="numeric"):
        """cfg: localization_amd.LocalizationConfig (reference yaml keys)."""
        return cls(cfg.nodes_id, cfg.nodes_pos, cfg.trajectory_length, cfg.maximum_velocity, cfg.distance_outlier,
                   cfg.maximum_iteration, cfg.minimum_optimize_error, cfg.publish_range, cfg.publish_pose,
                   cfg.publish_twist, cfg.publish_lidar, cfg.publish_imu, cfg.has_relative_range,
                   None if cfg.antenna_offset is None else np.asarray(cfg.antenna_offset).reshape(-1, 3), device, jacobian)

    def close(self):
        if getattr(self, "h", None):
            self.L.loc_node_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ret(self, rc, o):
        if rc < 0:
            raise _lib.LocalizationAmdError(rc, self.L.loc_last_error().decode(errors="replace"))
        return _out(o, rc)

    def add_range(self, requester_id, responder_id, stamp, distance, distance_err, antenna=0, frame_id="uwb"):
        o = NodeOutput()
        return self._ret(self.L.loc_node_add_range(self.h, requester_id, responder_id, float(stamp), float(distance),
                                                   float(distance_err), int(antenna), frame_id.encode(), C.byref(o)), o)

    def add_imu(self, stamp, q_xyzw, orientation_cov9, frame_id="imu_link"):
        o = NodeOutput()
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64); c = np.ascontiguousarray(orientation_cov9, dtype=np.float64).reshape(-1)
        return self._ret(self.L.loc_node_add_imu(self.h, float(stamp), _dp(q), _dp(c), frame_id.encode(), C.byref(o)), o)

    def add_pose(self, stamp, pose7, cov36, frame_id):
        o = NodeOutput()
        p = np.ascontiguousarray(pose7, dtype=np.float64); c = np.ascontiguousarray(cov36, dtype=np.float64).reshape(-1)
        return self._ret(self.L.loc_node_add_pose(self.h, float(stamp), _dp(p), _dp(c), frame_id.encode(), C.byref(o)), o)

    def add_twist(self, stamp, twist6, cov36, frame_id=""):
        o = NodeOutput()
        p = np.ascontiguousarray(twist6, dtype=np.float64); c = np.ascontiguousarray(cov36, dtype=np.float64).reshape(-1)
        return self._ret(self.L.loc_node_add_twist(self.h, float(stamp), _dp(p), _dp(c), frame_id.encode(), C.byref(o)), o)

    def add_rl_range(self, requester_id, responder_id, stamp, distance, requester_velocity):
        o = NodeOutput()
        v = np.ascontiguousarray(requester_velocity, dtype=np.float64)
        return self._ret(self.L.loc_node_add_rl_range(self.h, int(requester_id), int(responder_id), float(stamp), float(distance), _dp(v), C.byref(o)), o)

    def add_lidar(self, stamp, z, frame_id="lidar"):
        o = NodeOutput()
        return self._ret(self.L.loc_node_add_lidar(self.h, float(stamp), float(z), frame_id.encode(), C.byref(o)), o)

    def solve(self):
        o = NodeOutput()
        return self._ret(self.L.loc_node_solve(self.h, C.byref(o)), o)

    def path(self, node_id):
        out = np.zeros((max(self.T, 1), 8))
        n = self.L.loc_node_get_path(self.h, int(node_id), _dp(out), out.shape[0])
        if n < 0:
            raise _lib.LocalizationAmdError(n, self.L.loc_last_error().decode(errors="replace"))
        return out[:n]

    def set_deferred(self, on=True):
        check(self.L.loc_node_set_deferred(self.h, int(on)))

    @property
    def solve_pending(self):
        return bool(self.L.loc_node_solve_pending(self.h))

    def last_timing(self):
        """(pack, solve call, kernel) milliseconds of the last solve (loc_node_last_timing)"""
        t = (C.c_double * 3)()
        check(self.L.loc_node_last_timing(self.h, t))
        return float(t[0]), float(t[1]), float(t[2])

    def last_kernel_kind(self):
        """name of the kernel the node's last solve ran (loc_node_last_kernel_kind)"""
        from .window import WindowSolver
        k = C.c_int32()
        check(self.L.loc_node_last_kernel_kind(self.h, C.byref(k)))
        return WindowSolver.KERNEL_KINDS.get(k.value, str(k.value))

    def flush_tail(self):
        """path[T/2 .. T-1] of the moving tag: what Localization::~Localization appends to the optimized log (localization.cpp:708-717)"""
        out = np.zeros((max(self.T, 1), 8))
        n = self.L.loc_node_flush_tail(self.h, _dp(out), out.shape[0])
        if n < 0:
            raise _lib.LocalizationAmdError(n, self.L.loc_last_error().decode(errors="replace"))
        return out[:n]

    @property
    def number_measurements(self):
        return self.L.loc_node_number_measurements(self.h)


def release_batch_cache():
    """Free the batch solvers loc_nodes_solve_batch keeps cached for this thread."""
    L = lib(); _bind(L)
    check(L.loc_nodes_release_batch_cache())


def solve_batch(nodes):
    """One launch for every node of `nodes` that has a solve pending (deferred mode). Returns list of outputs."""
    L = lib(); _bind(L)
    arr = (C.c_void_p * len(nodes))(*[n.h for n in nodes])
    outs = (NodeOutput * len(nodes))()
    rc = L.loc_nodes_solve_batch(arr, len(nodes), outs)
    if rc < 0:
        raise _lib.LocalizationAmdError(rc, L.loc_last_error().decode(errors="replace"))
    return rc, [_out(o, 1 if o.solved else 0) for o in outs]
