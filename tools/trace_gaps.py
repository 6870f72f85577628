"""Gaps between consecutive kernels of the last timed steps in a rocprofv3 --kernel-trace CSV directory."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "bf_gather" in r["Kernel_Name"]]
sel = rows[idx[-6] - 1: idx[-6] + 9]
prev_end = None
for r in sel:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%-44s dur %9.1f us  gap before %6.1f us" % (r["Kernel_Name"][:44], (e - s) / 1e3, gap))
    prev_end = e
