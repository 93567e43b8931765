#!/usr/bin/env python3
"""Turns a gpurun_out/prof_* directory (rocprofv3 --kernel-trace --stats run of
bench.py plus one --pmc FETCH_SIZE and one --pmc WRITE_SIZE run) into the
small summaries kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/prof_r01c r01c
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
with open(stats) as f, open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w") as g:
    for line in f:
        if line.startswith('"Name"') or "pa::" in line:
            g.write(line)
shutil.copy(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), os.path.join(dst, "bench_kernel_stats_latest.csv"))
for name in ("bench.json", "bench_trace.json", "decode_isolated_bench.json", "decode_isolated_kernel_stats.csv",
             "bench_kernel_medians.csv"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
        if name.endswith(".csv"):  # stable names for the files the documents cite
            shutil.copy(p, os.path.join(dst, name.replace(".csv", "_latest.csv")))

out = {}
for sub, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(src, sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("void pa::"):
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        out.setdefault(k, {})[counter + "_KiB_mean_per_launch"] = sum(v) / len(v)
        out[k][counter + "_launches"] = len(v)
for k, d in list(out.items()):
    f = d.get("FETCH_SIZE_KiB_mean_per_launch", 0) * 1024
    w = d.get("WRITE_SIZE_KiB_mean_per_launch", 0) * 1024
    # gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads
    # at 64 bytes, so the read side is doubled; WRITE_SIZE is exact
    d["read_bytes_corrected_x2"] = 2 * f
    d["write_bytes"] = w
    d["hbm_traffic_bytes_per_launch"] = 2 * f + w
# what the counted kernels were built from: bench.py compares these with the tree it runs from and says so in its line
out["_kernel_sources_sha16"] = {
    name: hashlib.sha256(open(os.path.join(root, "vllm-v0.1.0_amd", "csrc", name), "rb").read()).hexdigest()[:16]
    for name in ("pa_decode.hip", "pa_cache.hip", "pa_common.h")}
with open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w") as g:
    json.dump(out, g, indent=1)
shutil.copy(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), os.path.join(dst, "pmc_hbm_traffic_latest.json"))
print(json.dumps(out, indent=1))
