#!/usr/bin/env python3
"""profiles/window_traffic.json from the tests/perf/pmc_summary.py outputs of the three window legs (tools/profile_window.sh):
    python tools/make_window_traffic.py cfg1_windows=PMC.json cfg4=PMC.json cfg5=PMC.json
FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (gfx950, MI355X_MICROARCH.md §HBM).  The sha256 of the kernel sources is
stored with the counts: bench.py withholds the traffic figures of the legs when the sources have changed since."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
legs = {}
for arg in sys.argv[1:]:
    leg, path = arg.split("=", 1)
    pj = json.load(open(path))
    legs[leg] = {"hbm_bytes_per_launch": (2.0 * pj["FETCH_SIZE"] + pj["WRITE_SIZE"]) * 1024.0,
                 "read_bytes": 2.0 * pj["FETCH_SIZE"] * 1024.0, "write_bytes": pj["WRITE_SIZE"] * 1024.0,
                 "valu_wave_instructions_per_launch": pj.get("SQ_INSTS_VALU"),
                 "source": os.path.relpath(os.path.abspath(path), ROOT) + ": rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) and WRITE_SIZE, "
                           "separate passes, per launch of the leg's single-GPU batch"}
out = {"kernel_source_sha256": bench.kernel_source_hash(bench.WINDOW_KERNEL_SOURCES), "kernel_sources": list(bench.WINDOW_KERNEL_SOURCES), "legs": legs}
json.dump(out, open(os.path.join(ROOT, "profiles", "window_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
