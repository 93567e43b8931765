#!/usr/bin/env python3
"""Per-kernel launch-duration statistics (count, median, mean, min, max) from a
rocprofv3 --kernel-trace CSV (one row per dispatch), written as a small CSV.
rocprofv3's own --stats summary has no median.

    python tools/kernel_trace_median.py <dir with *_kernel_trace.csv> <out.csv> [name filter, default "pa::"]
"""
import collections
import csv
import glob
import os
import sys

src, out = sys.argv[1], sys.argv[2]
needle = sys.argv[3] if len(sys.argv) > 3 else "pa::"
files = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
assert files, f"no *_kernel_trace.csv under {src}"
durs = collections.defaultdict(list)
for path in files:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"]
            if needle in name:
                durs[name.split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out, "w", newline="") as g:
    w = csv.writer(g)
    w.writerow(["Name", "Calls", "MedianNs", "AverageNs", "MinNs", "MaxNs"])
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
        v.sort()
        w.writerow([name, len(v), v[len(v) // 2], round(sum(v) / len(v), 1), v[0], v[-1]])
print(open(out).read())
