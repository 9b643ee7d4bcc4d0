"""Batched sliding-window graph solver: Python harness over loc_window_* (include/localization_amd.h)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


class WindowCaps(C.Structure):
    _fields_ = [("nv_max", C.c_int32), ("nr_max", C.c_int32), ("np_max", C.c_int32), ("ns_max", C.c_int32), ("bw_max", C.c_int32)]


def _inv_iso(R, t):
    Ri = R.T
    return Ri, -Ri @ t


class WindowBatch:
    """Host-side arrays of B instances in the ABI's layout; fill with add_* then hand to WindowSolver.solve."""

    def __init__(self, batch, nv_max, nr_max, np_max, ns_max):
        self.B = int(batch)
        self.caps = (int(nv_max), int(nr_max), int(np_max), int(ns_max))
        self.counts = np.zeros((self.B, 4), dtype=np.int32)
        self.poses = np.zeros((self.B, nv_max, 12))
        self.poses[:, :, 0] = self.poses[:, :, 4] = self.poses[:, :, 8] = 1.0
        self.r_idx = np.zeros((self.B, max(nr_max, 1), 2), dtype=np.int32)
        self.r_val = np.zeros((self.B, max(nr_max, 1), 5))
        self.p_idx = np.zeros((self.B, max(np_max, 1)), dtype=np.int32)
        self.p_val = np.zeros((self.B, max(np_max, 1), 18))
        self.s_idx = np.zeros((self.B, max(ns_max, 1), 4), dtype=np.int32)
        self.s_val = np.zeros((self.B, max(ns_max, 1), 48))
        self.result = np.zeros((self.B, 8))
        self.r_off1 = None   # optional lever arms of endpoint 1, [B][nr_max][3] (loc_window_set_endpoint1_offsets); created by add_range(off1=...)

    def add_pose(self, i, t, R=None):
        v = int(self.counts[i, 0])
        assert v < self.caps[0]
        self.poses[i, v, :9] = (np.eye(3) if R is None else np.asarray(R)).reshape(9)
        self.poses[i, v, 9:] = t
        self.counts[i, 0] = v + 1
        return v

    def add_range(self, i, v0, v1, meas, info, off=(0.0, 0.0, 0.0), anchor=False, off1=None):
        e = int(self.counts[i, 1])
        assert e < self.caps[1]
        self.r_idx[i, e] = (v0, -1 - v1 if anchor else v1)
        self.r_val[i, e] = (meas, info, off[0], off[1], off[2])
        if off1 is not None:
            if self.r_off1 is None:
                self.r_off1 = np.zeros((self.B, max(self.caps[1], 1), 3))
            self.r_off1[i, e] = off1
        self.counts[i, 1] = e + 1

    def add_prior(self, i, v, t, R, info_diag):
        e = int(self.counts[i, 2])
        assert e < self.caps[2]
        Ri, ti = _inv_iso(np.asarray(R, dtype=float), np.asarray(t, dtype=float))
        self.p_idx[i, e] = v
        self.p_val[i, e, :9] = Ri.reshape(9); self.p_val[i, e, 9:12] = ti; self.p_val[i, e, 12:] = info_diag
        self.counts[i, 2] = e + 1

    def add_se3(self, i, vi, vj, t, R, info, robust=True):
        e = int(self.counts[i, 3])
        assert e < self.caps[3]
        Ri, ti = _inv_iso(np.asarray(R, dtype=float), np.asarray(t, dtype=float))
        self.s_idx[i, e] = (vi, vj, int(robust), 0)
        self.s_val[i, e, :9] = Ri.reshape(9); self.s_val[i, e, 9:12] = ti
        self.s_val[i, e, 12:] = np.asarray(info, dtype=float).reshape(36)
        self.counts[i, 3] = e + 1

    def pose(self, i, v):
        return self.poses[i, v, :9].reshape(3, 3).copy(), self.poses[i, v, 9:].copy()


class WindowSolver:
    def __init__(self, anchors, batch, nv_max, nr_max, np_max=0, ns_max=0, maximum_iteration=10, device=0, bw_max=-1,
                 jacobian="numeric", natural_order=False, chain_threshold=None):
        L = lib()
        if L.loc_device_count() <= 0:
            raise _lib.LocalizationAmdError(_lib.LOC_ERR_NO_DEVICE, "no HIP device visible: localization_amd has no CPU fallback")
        anchors = np.ascontiguousarray(anchors, dtype=np.float64).reshape(-1, 3)
        caps = WindowCaps(nv_max, nr_max, np_max, ns_max, bw_max)
        h = C.c_void_p()
        check(L.loc_window_create(C.byref(h), device, int(batch), C.byref(caps), anchors.shape[0],
                                  anchors.ctypes.data_as(C.POINTER(C.c_double)), int(maximum_iteration)))
        self.h, self.L, self.B = h, L, int(batch)
        self.caps = (nv_max, nr_max, np_max, ns_max)
        self.lds_bytes = L.loc_window_lds_bytes(C.byref(caps))
        check(L.loc_window_set_jacobian(h, _lib.JAC_NUMERIC_G2O if jacobian in ("numeric", _lib.JAC_NUMERIC_G2O) else _lib.JAC_ANALYTIC))
        check(L.loc_window_set_ordering(h, int(bool(natural_order))))
        if chain_threshold is not None:   # smallest batch that takes the one-lane-per-window kernel for chain windows
            check(L.loc_window_set_chain_threshold(h, int(chain_threshold)))

    def close(self):
        if getattr(self, "h", None):
            self.L.loc_window_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _endpoint1(self, wb):
        dp = C.POINTER(C.c_double)
        off1 = getattr(wb, "r_off1", None)
        check(self.L.loc_window_set_endpoint1_offsets(self.h, wb.B, None if off1 is None else np.ascontiguousarray(off1).ctypes.data_as(dp)))

    def solve(self, wb: WindowBatch):
        assert wb.caps == self.caps and wb.B <= self.B
        self._endpoint1(wb)
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        for a in (wb.counts, wb.poses, wb.r_idx, wb.r_val, wb.p_idx, wb.p_val, wb.s_idx, wb.s_val, wb.result):
            assert a.flags["C_CONTIGUOUS"]
        check(self.L.loc_window_solve_host(self.h, wb.B, wb.counts.ctypes.data_as(ip), wb.poses.ctypes.data_as(dp),
                                           wb.r_idx.ctypes.data_as(ip), wb.r_val.ctypes.data_as(dp),
                                           wb.p_idx.ctypes.data_as(ip), wb.p_val.ctypes.data_as(dp),
                                           wb.s_idx.ctypes.data_as(ip), wb.s_val.ctypes.data_as(dp),
                                           wb.result.ctypes.data_as(dp)))
        return wb.result

    # ---- device-resident operation: upload once, solve any number of times from the uploaded estimates, download
    def upload(self, wb: WindowBatch):
        assert wb.caps == self.caps and wb.B <= self.B
        self._endpoint1(wb)
        ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
        check(self.L.loc_window_upload(self.h, wb.B, wb.counts.ctypes.data_as(ip), wb.poses.ctypes.data_as(dp),
                                       wb.r_idx.ctypes.data_as(ip), wb.r_val.ctypes.data_as(dp),
                                       wb.p_idx.ctypes.data_as(ip), wb.p_val.ctypes.data_as(dp),
                                       wb.s_idx.ctypes.data_as(ip), wb.s_val.ctypes.data_as(dp)))
        self._resident = wb.B

    def solve_resident(self, stream=None):
        check(self.L.loc_window_solve_resident(self.h, stream))

    def download(self, wb: WindowBatch):
        dp = C.POINTER(C.c_double)
        check(self.L.loc_window_download(self.h, wb.poses.ctypes.data_as(dp), wb.result.ctypes.data_as(dp)))
        return wb.result

    def timing_begin(self, max_launches):
        check(self.L.loc_window_timing_begin(self.h, int(max_launches)))

    def timing_end(self):
        n = C.c_int32(); tot = C.c_double(); avg = C.c_double()
        check(self.L.loc_window_timing_end(self.h, C.byref(n), C.byref(tot), C.byref(avg)))
        return n.value, tot.value, avg.value

    KERNEL_KINDS = {-1: "none", 0: "window_lm_kernel", 1: "chain_lm_kernel", 2: "chain3_lm_kernel", 3: "arrow3_lm_kernel", 4: "tree_wave_kernel", 5: "tree_lm_kernel", 6: "wave3_lm_kernel", 7: "wave6_lm_kernel", 8: "wave6_lm_kernel<SE3>"}

    def last_kernel_kind(self):
        """name of the kernel the last solve ran (loc_window_last_kernel_kind)"""
        k = C.c_int32()
        check(self.L.loc_window_last_kernel_kind(self.h, C.byref(k)))
        return self.KERNEL_KINDS.get(k.value, str(k.value))

    def last_kernel_ms(self):
        ms = C.c_double()
        check(self.L.loc_window_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    def set_option(self, name, value):
        """loc_window_set_option: the kernel-selection switches of this handle ("chain_min_batch", "arrow3", "tree", "wave3", "wave6",
        "chain3", "zero_copy", "topology_cache")"""
        self.L.loc_window_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        check(self.L.loc_window_set_option(self.h, str(name).encode(), int(value)))

    def last_host_timing(self):
        """(validate ms, structure analysis ms, staging + launch + copy back ms, verdict came from the cache) of the last solve()"""
        t = (C.c_double * 4)()
        check(self.L.loc_window_last_host_timing(self.h, t))
        return float(t[0]), float(t[1]), float(t[2]), bool(t[3])
