"""Batched snapshot solver (BASELINE config 2): Python harness over loc_snapshot_* (include/localization_amd.h).

Device memory is held in torch tensors; the library gets raw device pointers and the current torch HIP stream.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SnapshotParams, check, lib


def pack_ranges(x_kmb, pad_value=0.0):
    """[K][M][B] float32 (anchor-major SoA) -> device tile layout [K][M4][B][4] (numpy, host)."""
    x = np.ascontiguousarray(x_kmb, dtype=np.float32)
    K, M, B = x.shape
    M4 = (M + 3) // 4
    out = np.full((K, M4 * 4, B), pad_value, dtype=np.float32)
    out[:, :M, :] = x
    return np.ascontiguousarray(out.reshape(K, M4, 4, B).transpose(0, 1, 3, 2))


def unpack_ranges(tiles, M):
    """Inverse of pack_ranges: [K][M4][B][4] -> [K][M][B]."""
    t = np.asarray(tiles)
    K, M4, B, _ = t.shape
    return np.ascontiguousarray(t.transpose(0, 1, 3, 2).reshape(K, M4 * 4, B)[:, :M, :])


class SnapshotSolver:
    """B independent tags sharing one anchor map; each `solve` call runs K epochs of
    gate -> Cauchy range factors -> g2o-style LM (reference localization.cpp:297-376 cost, :164-170 solve)."""

    def __init__(self, anchors, batch, maximum_iteration=10, distance_outlier=1.0, jacobian="numeric",
                 lanes_per_instance=0, block_threads=0, device=0, gate_warmup_epochs=1):
        import torch
        self.torch = torch
        L = lib()
        if L.loc_device_count() <= 0 or not torch.cuda.is_available():
            raise _lib.LocalizationAmdError(_lib.LOC_ERR_NO_DEVICE,
                                            "no HIP device visible: localization_amd has no CPU fallback")
        anchors = np.ascontiguousarray(anchors, dtype=np.float64)
        assert anchors.ndim == 2 and anchors.shape[1] == 3
        self.M = anchors.shape[0]
        self.B = int(batch)
        self.device = int(device)
        self.dev = torch.device("cuda", self.device)
        prm = SnapshotParams()
        L.loc_snapshot_default_params(C.byref(prm))
        prm.maximum_iteration = int(maximum_iteration)
        prm.distance_outlier = float(distance_outlier)
        prm.gate_warmup_epochs = int(gate_warmup_epochs)
        prm.jacobian = {"analytic": _lib.JAC_ANALYTIC, "numeric": _lib.JAC_NUMERIC_G2O,
                        "numeric_g2o": _lib.JAC_NUMERIC_G2O}[jacobian]
        prm.lanes_per_instance = int(lanes_per_instance)
        prm.block_threads = int(block_threads)
        h = C.c_void_p()
        check(L.loc_snapshot_create(C.byref(h), self.device, self.B, self.M,
                                    anchors.ctypes.data_as(C.POINTER(C.c_double)), C.byref(prm)))
        self.h = h
        self.L = L
        self._pinned = []
        self.M4 = L.loc_snapshot_anchor_groups(h)
        self.lanes_per_instance = L.loc_snapshot_lanes_per_instance(h)

    def close(self):
        if getattr(self, "h", None):
            for p in getattr(self, "_pinned", []):
                self.L.loc_host_free(p)
            self._pinned = []
            self.L.loc_snapshot_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- state ------------------------------------------------------------------------------------
    def set_positions(self, pos_3b):
        p = np.ascontiguousarray(pos_3b, dtype=np.float64)
        assert p.shape == (3, self.B)
        check(self.L.loc_snapshot_set_positions(self.h, p.ctypes.data_as(C.POINTER(C.c_double))))

    def get_positions(self):
        p = np.zeros((3, self.B), dtype=np.float64)
        check(self.L.loc_snapshot_get_positions(self.h, p.ctypes.data_as(C.POINTER(C.c_double))))
        return p

    # ---- device-resident hot path --------------------------------------------------------------------
    def to_device_tiles(self, x_kmb, pad_value=0.0):
        return self.torch.from_numpy(pack_ranges(x_kmb, pad_value)).to(self.dev)

    def alloc_outputs(self, K, trials=True):
        t = self.torch
        out_pos = t.empty((K, 3, self.B), dtype=t.float64, device=self.dev)
        out_chi2 = t.empty((K, self.B), dtype=t.float64, device=self.dev)
        out_trials = t.empty((K, self.B), dtype=t.uint8, device=self.dev) if trials else None
        return out_pos, out_chi2, out_trials

    def solve_device(self, dist_tiles, err_tiles, out_pos, out_chi2, out_trials=None):
        """Asynchronous on the current torch stream. Tensors: float32 [K][M4][B][4], outputs as alloc_outputs."""
        t = self.torch
        K = dist_tiles.shape[0]
        assert tuple(dist_tiles.shape) == (K, self.M4, self.B, 4) and dist_tiles.dtype == t.float32
        assert tuple(err_tiles.shape) == tuple(dist_tiles.shape) and err_tiles.dtype == t.float32
        assert dist_tiles.is_contiguous() and err_tiles.is_contiguous()
        assert tuple(out_pos.shape) == (K, 3, self.B) and out_pos.dtype == t.float64 and out_pos.is_contiguous()
        assert tuple(out_chi2.shape) == (K, self.B) and out_chi2.dtype == t.float64 and out_chi2.is_contiguous()
        if out_trials is not None:
            assert tuple(out_trials.shape) == (K, self.B) and out_trials.dtype == t.uint8
        for x in (dist_tiles, err_tiles, out_pos, out_chi2):
            assert x.device == self.dev
        stream = t.cuda.current_stream(self.dev).cuda_stream
        check(self.L.loc_snapshot_solve_device(self.h, K, dist_tiles.data_ptr(), err_tiles.data_ptr(),
                                               out_pos.data_ptr(), out_chi2.data_ptr(),
                                               out_trials.data_ptr() if out_trials is not None else None,
                                               C.c_void_p(stream)))

    # ---- host convenience (PCIe staged) --------------------------------------------------------------
    def solve(self, dist_kmb, err_kmb):
        """dist/err: [K][M][B] float32 on the host. Returns (pos[K,3,B], chi2[K,B], trials[K,B]) numpy arrays."""
        d = pack_ranges(dist_kmb, 0.0)
        e = pack_ranges(err_kmb, 0.0)
        K = d.shape[0]
        assert d.shape == (K, self.M4, self.B, 4) and e.shape == d.shape
        out_pos = np.empty((K, 3, self.B)); out_chi2 = np.empty((K, self.B)); trials = np.empty((K, self.B), dtype=np.uint8)
        check(self.L.loc_snapshot_solve_host(self.h, K, d.ctypes.data_as(C.POINTER(C.c_float)),
                                             e.ctypes.data_as(C.POINTER(C.c_float)),
                                             out_pos.ctypes.data_as(C.POINTER(C.c_double)),
                                             out_chi2.ctypes.data_as(C.POINTER(C.c_double)),
                                             trials.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out_pos, out_chi2, trials

    def pinned(self, shape, dtype):
        """A page-locked numpy array (loc_host_alloc); freed when the solver is closed."""
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        check(self.L.loc_host_alloc(C.byref(p), n))
        self._pinned.append(p)
        return np.frombuffer((C.c_char * n).from_address(p.value), dtype=dtype).reshape(shape)

    def solve_stream(self, dist_kmb, err_kmb, out=None):
        """The pipelined host path (loc_snapshot_solve_host_kmb): [K][M][B] float32 host arrays in their natural layout,
        packed into tiles on the GPU, copy-in / solve / copy-out overlapped.  `out` = (pos, chi2, trials) to reuse
        (e.g. pinned) output arrays."""
        d = np.ascontiguousarray(dist_kmb, dtype=np.float32); e = np.ascontiguousarray(err_kmb, dtype=np.float32)
        K = d.shape[0]
        assert d.shape == (K, self.M, self.B) and e.shape == d.shape
        if out is None:
            out = (np.empty((K, 3, self.B)), np.empty((K, self.B)), np.empty((K, self.B), dtype=np.uint8))
        pos, chi2, trials = out
        assert pos.shape == (K, 3, self.B) and pos.dtype == np.float64 and chi2.shape == (K, self.B) and trials.shape == (K, self.B)
        check(self.L.loc_snapshot_solve_host_kmb(self.h, K, d.ctypes.data, e.ctypes.data, pos.ctypes.data, chi2.ctypes.data, trials.ctypes.data))
        return pos, chi2, trials

    # ---- HIP-event kernel timing -------------------------------------------------------------------
    def timing_begin(self, max_launches):
        check(self.L.loc_snapshot_timing_begin(self.h, int(max_launches)))

    def timing_end(self):
        n = C.c_int32(); tot = C.c_double(); avg = C.c_double()
        check(self.L.loc_snapshot_timing_end(self.h, C.byref(n), C.byref(tot), C.byref(avg)))
        return n.value, tot.value, avg.value
