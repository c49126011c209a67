"""GRBM_GUI_ACTIVE / duration per launch of tools/ubench/valu_rate -> the clock each kernel ran at.

usage: python tools/valu_clock.py <rocprofv3 output dir>   (finds *_counter_collection.csv and *_kernel_trace.csv)
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
name = {}
with open(kt) as f:
    for r in csv.DictReader(f):
        k = r.get("Dispatch_Id") or r.get("Dispatch_ID")
        dur[k] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        name[k] = r["Kernel_Name"]
act = defaultdict(float)
inst = defaultdict(int)
with open(cc) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            k = r.get("Dispatch_Id") or r.get("Dispatch_ID")
            act[k] += float(r["Counter_Value"])
            inst[k] += 1
# the counter is reported once per XCD (or summed over the 8 XCDs in one row): busy cycles of ONE die = sum / 8
XCDS = 8
for k in act:
    act[k] /= XCDS
rows = defaultdict(list)
for k, a in act.items():
    if k in dur and dur[k] > 0:
        rows[name[k]].append((a, dur[k]))
print("# kernel  launches  mean duration us  GRBM_GUI_ACTIVE / duration = MHz (min .. max over launches)")
for n, v in sorted(rows.items()):
    mhz = [a / t * 1e3 for a, t in v]
    print("%-60s %3d  %9.1f us  %7.0f MHz (%5.0f .. %5.0f)" % (
        n[:60], len(v), sum(t for _, t in v) / len(v) / 1e3, sum(mhz) / len(mhz), min(mhz), max(mhz)))
