#!/usr/bin/env python3
"""Per-launch durations of the decoder kernels of the last full decode in a rocprofv3 kernel trace:  python tools/experiments/jpeg_trace_tail.py <dir>"""
import csv, glob, sys
import os
f = max(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f)) if "jpeg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
big = [i for i, r in enumerate(rows) if "jpeg_idct<true>" in r["Kernel_Name"] and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 200000]
end = big[-1]; start = big[-2] + 1 if len(big) > 1 else 0
for r in rows[start:end + 1]:
    print("%-28s %9.1f us" % (r["Kernel_Name"].split("(")[0][:28], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
