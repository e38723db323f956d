"""Effective shader clock per kernel from one `rocprofv3 --pmc GRBM_GUI_ACTIVE` pass: GRBM_GUI_ACTIVE (summed over the 8 XCDs)
/ 8 / (End - Start).  Under PMC every dispatch carries ~25k cycles of overhead, so only kernels of >= ~50 us are meaningful.

    python tools/pmc_clock.py <dir>"""
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_rocprof import category  # noqa: E402

cyc, dur, n = defaultdict(float), defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        k = category(r["Kernel_Name"])
        cyc[k] += float(r["Counter_Value"]) / 8.0
        dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        n[k] += 1
for k in sorted(cyc, key=lambda k: -dur[k]):
    print("%-12s launches %4d  avg %8.1f us  %.3f GHz" % (k, n[k], dur[k] / n[k] / 1e3, cyc[k] / dur[k]))
