"""Batched fusion snapshot solver (BASELINE config 3): Python harness over loc_fusion_*."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib
from .snapshot import pack_ranges


class FusionParams(C.Structure):
    _fields_ = [("maximum_iteration", C.c_int32), ("distance_outlier", C.c_double), ("gate_warmup_epochs", C.c_int32),
                ("antenna_offset", C.c_double * 3), ("block_threads", C.c_int32), ("jacobian", C.c_int32)]


def _bind(L):
    if getattr(L, "_fusion_bound", False):
        return
    vp, dp, fp = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.loc_fusion_default_params.argtypes = [C.POINTER(FusionParams)]; L.loc_fusion_default_params.restype = None
    L.loc_fusion_create.argtypes = [C.POINTER(vp), C.c_int32, C.c_int64, C.c_int32, dp, C.POINTER(FusionParams)]
    L.loc_fusion_destroy.argtypes = [vp]
    L.loc_fusion_set_poses.argtypes = [vp, dp]
    L.loc_fusion_get_poses.argtypes = [vp, dp]
    L.loc_fusion_solve_device.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp, vp]
    L.loc_fusion_solve_host.argtypes = [vp, C.c_int32, fp, fp, dp, dp, dp, C.POINTER(C.c_uint8)]
    L.loc_fusion_last_kernel_ms.argtypes = [vp, dp]
    L.loc_fusion_timing_begin.argtypes = [vp, C.c_int32]
    L.loc_fusion_timing_end.argtypes = [vp, C.POINTER(C.c_int32), dp, dp]
    L._fusion_bound = True


class FusionSolver:
    """B tags, each a 6-DoF pose: M <= 8 anchor ranges with an antenna lever arm + an IMU rotation prior per epoch."""

    def __init__(self, anchors, batch, antenna_offset=(0.0, 0.0, 0.0), maximum_iteration=10, distance_outlier=3.0,
                 gate_warmup_epochs=1, block_threads=0, device=0, jacobian="numeric"):
        L = lib(); _bind(L)
        if L.loc_device_count() <= 0:
            raise _lib.LocalizationAmdError(_lib.LOC_ERR_NO_DEVICE, "no HIP device visible: localization_amd has no CPU fallback")
        anchors = np.ascontiguousarray(anchors, dtype=np.float64)
        self.M, self.B, self.device, self.L = anchors.shape[0], int(batch), int(device), L
        prm = FusionParams()
        L.loc_fusion_default_params(C.byref(prm))
        prm.maximum_iteration = int(maximum_iteration); prm.distance_outlier = float(distance_outlier)
        prm.gate_warmup_epochs = int(gate_warmup_epochs); prm.block_threads = int(block_threads)
        prm.jacobian = _lib.JAC_NUMERIC_G2O if jacobian in ("numeric", _lib.JAC_NUMERIC_G2O) else _lib.JAC_ANALYTIC
        prm.antenna_offset[0], prm.antenna_offset[1], prm.antenna_offset[2] = [float(v) for v in antenna_offset]
        h = C.c_void_p()
        check(L.loc_fusion_create(C.byref(h), self.device, self.B, self.M, anchors.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.loc_fusion_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_poses(self, pose_7b):
        p = np.ascontiguousarray(pose_7b, dtype=np.float64)
        assert p.shape == (7, self.B)
        check(self.L.loc_fusion_set_poses(self.h, p.ctypes.data_as(C.POINTER(C.c_double))))

    def get_poses(self):
        p = np.zeros((7, self.B))
        check(self.L.loc_fusion_get_poses(self.h, p.ctypes.data_as(C.POINTER(C.c_double))))
        return p

    def solve(self, dist_kmb, err_kmb, imu_kb8):
        """Host path. dist/err [K][M][B] f32, imu [K][B][8] f64. Returns (pose[K,7,B], chi2[K,B], trials[K,B])."""
        d = pack_ranges(dist_kmb, 0.0); e = pack_ranges(err_kmb, 0.0)
        K = d.shape[0]
        assert d.shape == (K, 2, self.B, 4), "the fusion kernel takes up to 8 anchors"
        imu = np.ascontiguousarray(imu_kb8, dtype=np.float64)
        assert imu.shape == (K, self.B, 8)
        pose = np.empty((K, 7, self.B)); chi2 = np.empty((K, self.B)); trials = np.empty((K, self.B), dtype=np.uint8)
        fp, dp = C.POINTER(C.c_float), C.POINTER(C.c_double)
        check(self.L.loc_fusion_solve_host(self.h, K, d.ctypes.data_as(fp), e.ctypes.data_as(fp), imu.ctypes.data_as(dp),
                                           pose.ctypes.data_as(dp), chi2.ctypes.data_as(dp), trials.ctypes.data_as(C.POINTER(C.c_uint8))))
        return pose, chi2, trials

    def solve_stream(self, dist_kmb, err_kmb, imu_kb8):
        """The pipelined host path (loc_fusion_solve_host_kmb): natural [K][M][B] arrays, tiles packed on the GPU."""
        d = np.ascontiguousarray(dist_kmb, dtype=np.float32); e = np.ascontiguousarray(err_kmb, dtype=np.float32)
        imu = np.ascontiguousarray(imu_kb8, dtype=np.float64)
        K = d.shape[0]
        assert d.shape == (K, self.M, self.B) and e.shape == d.shape and imu.shape == (K, self.B, 8)
        pose = np.empty((K, 7, self.B)); chi2 = np.empty((K, self.B)); trials = np.empty((K, self.B), dtype=np.uint8)
        self.L.loc_fusion_solve_host_kmb.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
        check(self.L.loc_fusion_solve_host_kmb(self.h, K, d.ctypes.data, e.ctypes.data, imu.ctypes.data, pose.ctypes.data,
                                               chi2.ctypes.data, trials.ctypes.data))
        return pose, chi2, trials

    def solve_device(self, dist_tiles, err_tiles, imu, out_pose, out_chi2, out_trials=None):
        import torch
        K = dist_tiles.shape[0]
        assert tuple(dist_tiles.shape) == (K, 2, self.B, 4) and tuple(imu.shape) == (K, self.B, 8)
        assert tuple(out_pose.shape) == (K, 7, self.B) and tuple(out_chi2.shape) == (K, self.B)
        for x in (dist_tiles, err_tiles, imu, out_pose, out_chi2):
            assert x.is_contiguous()
        stream = torch.cuda.current_stream(dist_tiles.device).cuda_stream
        check(self.L.loc_fusion_solve_device(self.h, K, dist_tiles.data_ptr(), err_tiles.data_ptr(), imu.data_ptr(),
                                             out_pose.data_ptr(), out_chi2.data_ptr(),
                                             out_trials.data_ptr() if out_trials is not None else None, C.c_void_p(stream)))

    def last_kernel_ms(self):
        ms = C.c_double()
        check(self.L.loc_fusion_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def timing_begin(self, max_launches):
        check(self.L.loc_fusion_timing_begin(self.h, int(max_launches)))

    def timing_end(self):
        n = C.c_int32(); tot = C.c_double(); avg = C.c_double()
        check(self.L.loc_fusion_timing_end(self.h, C.byref(n), C.byref(tot), C.byref(avg)))
        return n.value, tot.value, avg.value
