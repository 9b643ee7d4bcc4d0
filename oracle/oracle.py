"""ctypes wrapper around oracle/liboracle.so.

ORACLE — TEST INFRASTRUCTURE ONLY. Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by localization_amd/.  PARITY UNPINNED (see g2o_graph_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

JAC_NUMERIC_G2O = 0
JAC_ANALYTIC = 1


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


class OgStats(C.Structure):
    _fields_ = [("outer_iterations", C.c_int), ("lm_trials", C.c_int), ("terminated", C.c_int),
                ("lambda_", C.c_double), ("robust_chi2", C.c_double)]


class LoConfig(C.Structure):
    _fields_ = [("trajectory_length", C.c_int), ("maximum_velocity", C.c_double),
                ("distance_outlier", C.c_double), ("maximum_iteration", C.c_int),
                ("minimum_optimize_error", C.c_double),
                ("publish_range", C.c_int), ("publish_pose", C.c_int), ("publish_twist", C.c_int),
                ("publish_lidar", C.c_int), ("publish_imu", C.c_int),
                ("has_relative_range", C.c_int), ("jac_mode", C.c_int), ("publish_relative_range", C.c_int)]


class LoOutput(C.Structure):
    _fields_ = [("solved", C.c_int), ("published", C.c_int), ("chi2", C.c_double),
                ("realtime", C.c_double * 8), ("optimized", C.c_double * 8),
                ("outer_iterations", C.c_int), ("lm_trials", C.c_int)]


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.og_create.restype = C.c_void_p
        L.og_destroy.argtypes = [C.c_void_p]
        L.og_add_vertex.argtypes = [C.c_void_p, C.c_int, dp, dp, C.c_int]
        L.og_remove_vertex.argtypes = [C.c_void_p, C.c_int]
        L.og_set_estimate.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.og_get_estimate.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.og_add_range_edge.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, dp, dp, C.c_int]
        L.og_add_prior_edge.argtypes = [C.c_void_p, C.c_int, dp, dp, dp]
        L.og_add_se3_edge.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp, C.c_int]
        L.og_optimize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(OgStats)]
        L.og_chi2.argtypes = [C.c_void_p]
        L.og_chi2.restype = C.c_double
        L.og_num_vertices.argtypes = [C.c_void_p]
        L.og_num_edges.argtypes = [C.c_void_p]
        L.og_quat_to_R.argtypes = [dp, dp]
        L.og_R_to_quat.argtypes = [dp, dp]
        L.og_from_vector_mqt.argtypes = [dp, dp, dp]
        L.og_to_vector_mqt.argtypes = [dp, dp, dp]
        L.og_debug_linearize.argtypes = [C.c_void_p, C.c_int, C.c_int, dp, dp, dp]
        L.og_cauchy_rho.argtypes = [C.c_double, dp]
        L.og_cauchy_rho.restype = C.c_double
        L.og_snapshot_batch.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                        dp, dp, dp, C.POINTER(C.c_ubyte), C.c_int, C.c_double, C.c_int, C.c_int]
        L.og_fusion_batch.argtypes = [C.c_int, C.c_int, C.c_int, dp, dp, C.POINTER(C.c_float), C.POINTER(C.c_float), dp, dp, dp, dp,
                                      C.POINTER(C.c_ubyte), C.c_int, C.c_double, C.c_int, C.c_int]
        L.lo_create.restype = C.c_void_p
        L.lo_create.argtypes = [C.POINTER(LoConfig), C.c_int, C.POINTER(C.c_int), dp, C.c_int, dp]
        L.lo_destroy.argtypes = [C.c_void_p]
        L.lo_add_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_float, C.c_float, C.c_int,
                                   C.c_char_p, C.POINTER(LoOutput)]
        L.lo_add_imu.argtypes = [C.c_void_p, C.c_double, dp, dp, C.c_char_p, C.POINTER(LoOutput)]
        L.lo_add_pose.argtypes = [C.c_void_p, C.c_double, dp, dp, C.c_char_p, C.POINTER(LoOutput)]
        L.lo_add_twist.argtypes = [C.c_void_p, C.c_double, dp, dp, C.c_char_p, C.POINTER(LoOutput)]
        L.lo_add_lidar.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_char_p, C.POINTER(LoOutput)]
        L.lo_add_rl_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, dp, C.POINTER(LoOutput)]
        L.lo_solve.argtypes = [C.c_void_p, C.POINTER(LoOutput)]
        L.lo_get_path.argtypes = [C.c_void_p, C.c_int, dp]
        L.lo_number_measurements.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


_I3 = np.eye(3)


class Graph:
    """Thin object wrapper over og_graph (g2o::SparseOptimizer restatement)."""

    def __init__(self):
        self.L = lib()
        self.h = self.L.og_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.og_destroy(self.h)
            self.h = None

    def add_vertex(self, vid, t, R=None, fixed=False):
        R = np.ascontiguousarray(_I3 if R is None else R, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        return self.L.og_add_vertex(self.h, vid, _dp(R), _dp(t), int(fixed))

    def remove_vertex(self, vid):
        return self.L.og_remove_vertex(self.h, vid)

    def estimate(self, vid):
        R = np.zeros((3, 3)); t = np.zeros(3)
        rc = self.L.og_get_estimate(self.h, vid, _dp(R), _dp(t))
        assert rc == 0
        return R, t

    def set_estimate(self, vid, t, R=None):
        R = np.ascontiguousarray(_I3 if R is None else R, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        return self.L.og_set_estimate(self.h, vid, _dp(R), _dp(t))

    def add_range_edge(self, v0, v1, meas, info, off0=None, off1=None, robust=True):
        o0 = None if off0 is None else np.ascontiguousarray(off0, dtype=np.float64)
        o1 = None if off1 is None else np.ascontiguousarray(off1, dtype=np.float64)
        return self.L.og_add_range_edge(self.h, v0, v1, float(meas), float(info), _dp(o0), _dp(o1), int(robust))

    def add_prior_edge(self, v, t, R, info):
        R = np.ascontiguousarray(R, dtype=np.float64); t = np.ascontiguousarray(t, dtype=np.float64)
        info = np.ascontiguousarray(info, dtype=np.float64)
        return self.L.og_add_prior_edge(self.h, v, _dp(R), _dp(t), _dp(info))

    def add_se3_edge(self, v0, v1, t, R, info, robust=True):
        R = np.ascontiguousarray(R, dtype=np.float64); t = np.ascontiguousarray(t, dtype=np.float64)
        info = np.ascontiguousarray(info, dtype=np.float64)
        return self.L.og_add_se3_edge(self.h, v0, v1, _dp(R), _dp(t), _dp(info), int(robust))

    def optimize(self, iterations, jac_mode=JAC_NUMERIC_G2O):
        st = OgStats()
        n = self.L.og_optimize(self.h, iterations, jac_mode, C.byref(st))
        return n, st

    def chi2(self):
        return self.L.og_chi2(self.h)

    def linearize(self, edge_index, jac_mode=JAC_NUMERIC_G2O):
        err = np.zeros(6); J0 = np.zeros((6, 6)); J1 = np.zeros((6, 6))
        dim = self.L.og_debug_linearize(self.h, edge_index, jac_mode, _dp(err), _dp(J0), _dp(J1))
        assert dim > 0
        return err[:dim].copy(), J0[:dim].copy(), J1[:dim].copy()


def snapshot_batch(anchors, dist, err, pos, iterations=10, gate=1.0, jac_mode=JAC_NUMERIC_G2O, gate_from_epoch=1):
    """dist/err: [K][M][B] float32, pos: [3][B] float64 prior. Returns (out_pos[K,3,B], chi2[K,B], trials[K,B], pos_last[3,B])."""
    anchors = np.ascontiguousarray(anchors, dtype=np.float64)
    dist = np.ascontiguousarray(dist, dtype=np.float32)
    err = np.ascontiguousarray(err, dtype=np.float32)
    K, M, B = dist.shape
    assert anchors.shape == (M, 3) and err.shape == dist.shape
    p = np.array(pos, dtype=np.float64, order="C", copy=True)
    assert p.shape == (3, B)
    out_pos = np.zeros((K, 3, B)); out_chi2 = np.zeros((K, B)); trials = np.zeros((K, B), dtype=np.uint8)
    rc = lib().og_snapshot_batch(B, K, M, _dp(anchors), dist.ctypes.data_as(C.POINTER(C.c_float)),
                                 err.ctypes.data_as(C.POINTER(C.c_float)), _dp(p), _dp(out_pos), _dp(out_chi2),
                                 trials.ctypes.data_as(C.POINTER(C.c_ubyte)), iterations, float(gate), int(gate_from_epoch), jac_mode)
    assert rc == 0
    return out_pos, out_chi2, trials, p


def fusion_batch(anchors, offset, dist, err, imu, pose, iterations=10, gate=3.0, jac_mode=JAC_NUMERIC_G2O, gate_from_epoch=1):
    """BASELINE config 3. dist/err [K][M][B] f32, imu [K][B][8] f64 (q xyzw, cov diag, pad), pose [7][B] (t, q xyzw).
    Returns (out_pose[K,7,B], chi2[K,B], trials[K,B], pose_last[7,B])."""
    anchors = np.ascontiguousarray(anchors, dtype=np.float64); offset = np.ascontiguousarray(offset, dtype=np.float64)
    dist = np.ascontiguousarray(dist, dtype=np.float32); err = np.ascontiguousarray(err, dtype=np.float32)
    imu = np.ascontiguousarray(imu, dtype=np.float64)
    K, M, B = dist.shape
    assert imu.shape == (K, B, 8) and anchors.shape == (M, 3)
    p = np.array(pose, dtype=np.float64, order="C", copy=True)
    assert p.shape == (7, B)
    out_pose = np.zeros((K, 7, B)); out_chi2 = np.zeros((K, B)); trials = np.zeros((K, B), dtype=np.uint8)
    rc = lib().og_fusion_batch(B, K, M, _dp(anchors), _dp(offset), dist.ctypes.data_as(C.POINTER(C.c_float)),
                               err.ctypes.data_as(C.POINTER(C.c_float)), _dp(imu), _dp(p), _dp(out_pose), _dp(out_chi2),
                               trials.ctypes.data_as(C.POINTER(C.c_ubyte)), iterations, float(gate), int(gate_from_epoch), jac_mode)
    assert rc == 0
    return out_pose, out_chi2, trials, p


class LocalizationOracle:
    """Restatement of class Localization (reference localization.h:99-200) for a stream of messages."""

    def __init__(self, nodes_id, nodes_pos, trajectory_length, maximum_velocity=1.0, distance_outlier=1.0,
                 maximum_iteration=20, minimum_optimize_error=1000.0, publish_range=False, publish_pose=False,
                 publish_twist=False, publish_lidar=False, publish_imu=False, has_relative_range=False,
                 antenna_offsets=None, jac_mode=JAC_NUMERIC_G2O, publish_relative_range=False):
        self.L = lib()
        cfg = LoConfig(trajectory_length, maximum_velocity, distance_outlier, maximum_iteration,
                       minimum_optimize_error, int(publish_range), int(publish_pose), int(publish_twist),
                       int(publish_lidar), int(publish_imu), int(has_relative_range), jac_mode, int(publish_relative_range))
        ids = (C.c_int * len(nodes_id))(*[int(i) for i in nodes_id])
        pos = np.ascontiguousarray(nodes_pos, dtype=np.float64).reshape(-1)
        assert pos.size == 3 * len(nodes_id)
        ant = None; n_ant = 0
        if antenna_offsets is not None:
            ant = np.ascontiguousarray(antenna_offsets, dtype=np.float64).reshape(-1)
            n_ant = ant.size // 3
        self.T = trajectory_length
        self.h = self.L.lo_create(C.byref(cfg), len(nodes_id), ids, _dp(pos), n_ant, _dp(ant))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            self.L.lo_destroy(self.h)
            self.h = None

    @staticmethod
    def _out(o, rc):
        return dict(rc=rc, solved=bool(o.solved), published=bool(o.published), chi2=o.chi2,
                    realtime=np.array(o.realtime[:]), optimized=np.array(o.optimized[:]),
                    outer_iterations=o.outer_iterations, lm_trials=o.lm_trials)

    def add_range(self, requester_id, responder_id, stamp, distance, distance_err, antenna=0, frame_id="uwb"):
        o = LoOutput()
        rc = self.L.lo_add_range(self.h, requester_id, responder_id, stamp, float(distance), float(distance_err),
                                 antenna, frame_id.encode(), C.byref(o))
        return self._out(o, rc)

    def add_imu(self, stamp, q_xyzw, orientation_cov9, frame_id="imu_link"):
        o = LoOutput()
        q = np.ascontiguousarray(q_xyzw, dtype=np.float64); c = np.ascontiguousarray(orientation_cov9, dtype=np.float64).reshape(-1)
        rc = self.L.lo_add_imu(self.h, stamp, _dp(q), _dp(c), frame_id.encode(), C.byref(o))
        return self._out(o, rc)

    def add_pose(self, stamp, pose7, cov36, frame_id):
        o = LoOutput()
        p = np.ascontiguousarray(pose7, dtype=np.float64); c = np.ascontiguousarray(cov36, dtype=np.float64).reshape(-1)
        rc = self.L.lo_add_pose(self.h, stamp, _dp(p), _dp(c), frame_id.encode(), C.byref(o))
        return self._out(o, rc)

    def add_twist(self, stamp, twist6, cov36, frame_id=""):
        o = LoOutput()
        p = np.ascontiguousarray(twist6, dtype=np.float64); c = np.ascontiguousarray(cov36, dtype=np.float64).reshape(-1)
        rc = self.L.lo_add_twist(self.h, stamp, _dp(p), _dp(c), frame_id.encode(), C.byref(o))
        return self._out(o, rc)

    def add_rl_range(self, requester_id, responder_id, stamp, distance, requester_velocity):
        o = LoOutput()
        v = np.ascontiguousarray(requester_velocity, dtype=np.float64)
        rc = self.L.lo_add_rl_range(self.h, requester_id, responder_id, stamp, float(distance), _dp(v), C.byref(o))
        return self._out(o, rc)

    def add_lidar(self, stamp, z, frame_id="lidar"):
        o = LoOutput()
        rc = self.L.lo_add_lidar(self.h, stamp, float(z), frame_id.encode(), C.byref(o))
        return self._out(o, rc)

    def solve(self):
        o = LoOutput()
        rc = self.L.lo_solve(self.h, C.byref(o))
        return self._out(o, rc)

    def path(self, node_id):
        out = np.zeros((max(self.T, 1), 8))
        n = self.L.lo_get_path(self.h, node_id, _dp(out))
        return out[:n]
