"""CPU-side checks of the C ABI: the library loads, exports every symbol include/localization_amd.h declares, refuses
to run without a HIP device (no CPU fallback), and its host helpers agree with the Python harness.  No compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import localization_amd as la
from localization_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "localization_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(loc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(built):
    L = la.lib()
    names = declared_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/localization_amd.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names
    assert la.abi_version() == 4


def test_no_cpu_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = la.lib()
    assert L.loc_device_count() == 0
    anchors = np.zeros((8, 3))
    h = C.c_void_p()
    rc = L.loc_snapshot_create(C.byref(h), 0, 16, 8, anchors.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == _lib.LOC_ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in L.loc_last_error()
    with pytest.raises(la.LocalizationAmdError):
        la.SnapshotSolver(anchors, 16)


def test_default_params_mirror_reference_defaults(built):
    p = _lib.SnapshotParams()
    la.lib().loc_snapshot_default_params(C.byref(p))
    assert p.maximum_iteration == 20          # localization.cpp:65
    assert p.distance_outlier == 1.0          # localization.cpp:78
    # the reference's EdgeSE3Range has no linearizeOplus (types_edge_se3range.h:45-74): g2o's numeric Jacobians are the default
    assert p.gate_warmup_epochs == 1 and p.jacobian == _lib.JAC_NUMERIC_G2O
    from localization_amd.node import NodeConfig
    from localization_amd.fusion import FusionParams
    nc = NodeConfig()
    la.lib().loc_node_default_config(C.byref(nc))
    assert nc.jacobian == _lib.JAC_NUMERIC_G2O and nc.maximum_iteration == 20 and nc.minimum_optimize_error == 1000.0
    fp = FusionParams()
    la.lib().loc_fusion_default_params(C.byref(fp))
    assert fp.jacobian == _lib.JAC_NUMERIC_G2O


def test_pack_unpack_ranges_round_trip():
    rng = np.random.default_rng(0)
    for (K, M, B) in [(1, 8, 5), (3, 5, 130), (2, 12, 64), (4, 3, 1)]:
        x = rng.random((K, M, B)).astype(np.float32)
        t = la.pack_ranges(x, pad_value=0.0)
        M4 = (M + 3) // 4
        assert t.shape == (K, M4, B, 4)
        assert np.array_equal(la.unpack_ranges(t, M), x)
        # anchor m sits at [m // 4][b][m % 4]; padded lanes hold the pad value
        for m in range(M):
            assert np.array_equal(t[:, m // 4, :, m % 4], x[:, m, :])
        if M % 4:
            assert (t[:, -1, :, M % 4:] == 0).all()


def test_config_loader_reads_reference_style_yaml(tmp_path):
    y = tmp_path / "uwb_only.yaml"
    y.write_text("robot:\n  trajectory_length: 10\n  maximum_velocity: 5.0\n  distance_outlier: 1\n"
                 "optimizer:\n  maximum_iteration: 10\n  minimum_optimize_error: 2000\n  verbose: false\n"
                 "topic:\n  range: /lpsrange\npublish_flag:\n  tf: true\n  range: true\n"
                 "frame:\n  target: /uwb_localization\n  source: /world\n")
    u = tmp_path / "anchor.yaml"
    u.write_text("uwb:\n  nodesId: [100, 101, 102, 103, 200]\n  nodesPos: [3,-3,0.58, 3,3,1.97, -3,3,0.54, -3,-3,1.76, 0,0,1]\n")
    c = la.load_config(str(y), str(u))
    assert (c.trajectory_length, c.maximum_velocity, c.distance_outlier) == (10, 5.0, 1.0)
    assert (c.maximum_iteration, c.minimum_optimize_error) == (10, 2000.0)
    assert c.publish_range and c.publish_tf and not c.publish_imu and not c.has_relative_range
    assert c.topics == {"range": "/lpsrange"} and c.frame_source == "/world"
    assert c.nodes_id[-1] == 200 and len(c.nodes_pos) == 15 and c.antenna_offset is None
    d = la.LocalizationConfig()               # reference defaults (localization.cpp:58-159)
    assert (d.maximum_iteration, d.minimum_optimize_error, d.maximum_velocity, d.distance_outlier) == (20, 1000.0, 1.0, 1.0)
    assert d.trajectory_length is None and d.frame_target == "estimation" and d.frame_source == "local_origin"


def test_header_is_plain_c_and_a_c_program_links(tmp_path, built):
    """The boundary is a C ABI: the header compiles as strict C99 (no C++ types leak through), and a C program links against
    liblocalization_amd.so and runs its non-compute entry points on a machine without a GPU."""
    import subprocess
    src = tmp_path / "abi_c.c"
    src.write_text('#include "localization_amd.h"\n'
                   "int main(void) {\n"
                   "  loc_snapshot_params p; loc_node_config c; loc_fusion_params f;\n"
                   "  loc_snapshot_default_params(&p); loc_node_default_config(&c); loc_fusion_default_params(&f);\n"
                   "  if (loc_abi_version() != LOC_ABI_VERSION) return 1;\n"
                   "  if (p.maximum_iteration != 20 || c.maximum_iteration != 20) return 2;   /* reference default, localization.cpp:65 */\n"
                   "  return 0;\n}\n")
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "localization_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    exe = tmp_path / "abi_c"
    subprocess.check_call(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-llocalization_amd",
                           "-Wl,-rpath," + libdir])
    assert subprocess.call([str(exe)]) == 0
