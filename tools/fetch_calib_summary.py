#!/usr/bin/env python3
"""profiles/r03_fetch_calibration.json from the two --pmc passes of tools/fetch_calib.bin:
    python tools/fetch_calib_summary.py FETCH_DIR WRITE_DIR > profiles/r03_fetch_calibration.json
FETCH_SIZE / WRITE_SIZE are in KB.  ratio = counted bytes / bytes the kernel is known to move (for the strided patterns: per
requested byte AND per 64-byte / 128-byte line touched)."""
import csv, glob, json, os, sys
GiB = 1 << 30
def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]) * 1024.0)
    return out
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
def first(d, name, k=0):
    v = [x for key, xs in d.items() if name in key for x in xs]
    return v[k] if len(v) > k else None
res = {"unit": "bytes (counter value x 1024)", "known_bytes_per_kernel": GiB}
r16, r8 = first(fetch, "calib_read16"), first(fetch, "calib_read8(") or first(fetch, "calib_read8")
s128, s4k = first(fetch, "calib_read8_strided", 0), first(fetch, "calib_read8_strided", 1)
w8, ws = first(write, "calib_write8(") or first(write, "calib_write8"), first(write, "calib_write8_strided")
res["read16_coalesced"] = {"FETCH_SIZE_bytes": r16, "ratio_counted_over_known": r16 / GiB if r16 else None}
res["read8_coalesced"] = {"FETCH_SIZE_bytes": r8, "ratio_counted_over_known": r8 / GiB if r8 else None}
for name, v, lines in (("read8_stride128", s128, GiB // 128), ("read8_stride4096", s4k, GiB // 4096)):
    res[name] = {"FETCH_SIZE_bytes": v, "requested_bytes": lines * 8, "lines_touched": lines,
                 "counted_per_requested_byte": v / (lines * 8) if v else None, "counted_bytes_per_line": v / lines if v else None}
res["write8_coalesced"] = {"WRITE_SIZE_bytes": w8, "ratio_counted_over_known": w8 / GiB if w8 else None}
res["write8_stride128"] = {"WRITE_SIZE_bytes": ws, "requested_bytes": GiB // 128 * 8, "counted_bytes_per_line": ws / (GiB // 128) if ws else None}
res["raw"] = {"fetch": fetch, "write": write}
print(json.dumps(res, indent=1))
