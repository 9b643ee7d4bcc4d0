#!/usr/bin/env python3
"""profiles/window_traffic.json from the tests/perf/pmc_summary.py outputs of the three window legs (tools/profile_window.sh):
    python tools/make_window_traffic.py cfg1_windows=PMC.json cfg4=PMC.json cfg5=PMC.json
FETCH_SIZE / WRITE_SIZE are in KB.  The read factor is MEASURED, not assumed: tools/fetch_calib.hip moves a known GiB with each access
pattern of these kernels under the same counters (profiles/r03_fetch_calibration.json): FETCH_SIZE reports 0.500 of the bytes of a
16-B-per-lane stream (the guide's case), 0.500 of an 8-B-per-lane coalesced stream (the [entry][lane] workspaces: 512-byte lines), and
64 B per cache line touched by a scattered 8-byte read (lanes 128 B or 4 KB apart: window_lm_kernel's per-instance workspaces) — i.e.
FETCH_SIZE x 2 = 128-byte lines fetched, for every pattern; WRITE_SIZE is exact for full lines and counts 32-byte sectors for
scattered 8-byte stores.  The sha256 of the kernel sources is stored with the counts: bench.py withholds the traffic figures of the legs
when the sources have changed since."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
cal = json.load(open(os.path.join(ROOT, "profiles", "r03_fetch_calibration.json")))
calib_read = 1.0 / cal["read8_coalesced"]["ratio_counted_over_known"]
assert abs(calib_read - 1.0 / cal["read16_coalesced"]["ratio_counted_over_known"]) < 0.01 and abs(cal["read8_stride128"]["counted_bytes_per_line"] * calib_read - 128.0) < 1.0
legs = {}
for arg in sys.argv[1:]:
    leg, path = arg.split("=", 1)
    pj = json.load(open(path))
    legs[leg] = {"hbm_bytes_per_launch": (calib_read * pj["FETCH_SIZE"] + pj["WRITE_SIZE"]) * 1024.0,
                 "read_bytes": calib_read * pj["FETCH_SIZE"] * 1024.0, "write_bytes": pj["WRITE_SIZE"] * 1024.0,
                 "valu_wave_instructions_per_launch": pj.get("SQ_INSTS_VALU"),
                 "calibration": {"file": "profiles/r03_fetch_calibration.json", "read_factor": calib_read, "write_factor": 1.0,
                                 "note": "FETCH_SIZE x read_factor = bytes of the 128-B lines fetched (measured 0.5000 counted / moved for 16-B and 8-B per lane "
                                         "coalesced streams, 64 B counted per line for scattered 8-B reads); WRITE_SIZE exact for full lines, 32-B sectors for scattered stores"},
                 "kernel_source_sha256": bench.kernel_source_hash(bench.WINDOW_LEG_SOURCES[leg]), "kernel_sources": list(bench.WINDOW_LEG_SOURCES[leg]),
                 "source": os.path.relpath(os.path.abspath(path), ROOT) + ": rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE, separate passes, per launch of the "
                           "leg's single-GPU batch; factors from profiles/r03_fetch_calibration.json"}
path = os.path.join(ROOT, "profiles", "window_traffic.json")
old = json.load(open(path)).get("legs", {}) if os.path.exists(path) else {}
old.update(legs)   # (legs not named on the command line keep their entries: each carries the hash of its own kernel's sources)
out = {"legs": old}
json.dump(out, open(os.path.join(ROOT, "profiles", "window_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
