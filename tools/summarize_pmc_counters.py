#!/usr/bin/env python3
"""Median per counter over the dispatches of one kernel, from rocprofv3 --pmc csv output.
usage: summarize_pmc_counters.py <dir holding one sub-directory per pass> <kernel name substring>"""
import collections
import csv
import glob
import os
import statistics
import sys


def main():
    src, kernel = sys.argv[1], sys.argv[2]
    for sub in sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d))):
        values = collections.defaultdict(list)
        durations = {}
        for path in glob.glob(os.path.join(src, sub, "**", "*_counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for r in csv.DictReader(f):
                    if kernel not in r["Kernel_Name"]:
                        continue
                    values[r["Counter_Name"]].append(float(r["Counter_Value"]))
                    if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                        durations[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if not values:
            print(f"# pass {sub}: no dispatch of {kernel}")
            continue
        n = max(len(v) for v in values.values())
        dur = f", profiled duration median {statistics.median(durations.values()):.0f} us" if durations else ""
        print(f"# pass {sub}: {n} launches{dur}")
        for name in sorted(values):
            print(f"{name:32s} median {statistics.median(values[name]):.4e}")


if __name__ == "__main__":
    main()
