#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel: python tests/perf/pmc_summary.py KERNEL_SUBSTRING DIR [DIR ...]
Prints {counter: value per launch} for dispatches whose kernel name contains KERNEL_SUBSTRING (each DIR = one --pmc pass)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

key, dirs = sys.argv[1], sys.argv[2:]
out = {}
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        tot, disp = defaultdict(float), defaultdict(set)
        for row in csv.DictReader(open(f)):
            if key in row["Kernel_Name"]:
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                disp[row["Counter_Name"]].add(row["Dispatch_Id"])
        for c, v in tot.items():
            out[c] = v / max(1, len(disp[c]))
            out.setdefault("_launches", len(disp[c]))
print(json.dumps(out, indent=1))
