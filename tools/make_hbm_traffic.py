#!/usr/bin/env python3
"""profiles/hbm_traffic.json from the tests/perf/pmc_summary.py outputs of the headline run in each Jacobian mode (tools/profile_r04.sh):
    python tools/make_hbm_traffic.py numeric=PMC_NUMERIC.json analytic=PMC_ANALYTIC.json [SOURCE_DIR_NOTE]
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950: wide coalesced streaming reads are tallied at half their
bytes, MI355X_MICROARCH.md §HBM; measured again by tools/fetch_calib.hip).  The sha256 of the kernel sources is stored with the
counts: bench.py withholds the PMC-derived fields when the sources have changed since."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
pairs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
notes = [a for a in sys.argv[1:] if "=" not in a]
out = {"modes": {}, "measured_issue_ceiling_lane_slots_per_s": 30000000000000.0,
       "ceiling_source": "profiles/r01_fp64_probe.txt (tools/fp64_probe.hip): 30 T f64 lane-ops/s sustained, reached already at one wave per SIMD",
       "kernel_source_sha256": bench.kernel_source_hash(bench.SNAPSHOT_KERNEL_SOURCES), "kernel_sources": list(bench.SNAPSHOT_KERNEL_SOURCES)}
for mode, path in pairs:
    pj = json.load(open(path))
    rd, wr = 2.0 * pj["FETCH_SIZE"] * 1024.0, pj["WRITE_SIZE"] * 1024.0
    f64 = 64.0 * (pj["SQ_INSTS_VALU_ADD_F64"] + pj["SQ_INSTS_VALU_MUL_F64"] + 2 * pj["SQ_INSTS_VALU_FMA_F64"] + pj["SQ_INSTS_VALU_TRANS_F64"])
    out["modes"][mode] = {"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr,
                          "source": (notes[0] + "/" if notes else "") + os.path.basename(path) + ": rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) and WRITE_SIZE, "
                                    "separate passes, per launch of 65536 tags x 128 epochs",
                          "f64_flop_per_launch": f64, "valu_wave_insts_per_launch": pj["SQ_INSTS_VALU"], "salu_wave_insts_per_launch": pj["SQ_INSTS_SALU"],
                          "issue_lane_slots_per_launch": 64.0 * (pj["SQ_INSTS_VALU"] + pj["SQ_INSTS_SALU"])}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
