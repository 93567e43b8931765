#!/usr/bin/env python3
"""Groups a rocprofv3 *_kernel_stats.csv by operator class (this package's
kernels by name, GEMMs, everything else) and prints total device time per
class, optionally divided by a number of steps.

    python tools/summarize_kernel_classes.py kernel_stats.csv [steps]
"""
import csv
import sys

CLASSES = (
    ("pa_decode", "paged attention, generation tokens (pa_decode_*)"),
    ("pa_prompt", "prompt attention (pa_prompt_kernel)"),
    ("pa_token", "reshape_and_cache (pa_token_*)"),
    ("pa_rms_norm", "rms_norm"), ("pa_rotary", "rotary_embedding_neox"), ("pa_silu", "silu_and_mul"),
    ("Cijk", "GEMMs (hipBLASLt / rocBLAS)"),
)


def main():
    path = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    tot = {}
    calls = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"]
        key = next((label for sub, label in CLASSES if sub in name), "other (adds, embedding, sampling, copies)")
        tot[key] = tot.get(key, 0.0) + float(r["TotalDurationNs"])
        calls[key] = calls.get(key, 0) + int(r["Calls"])
    total = sum(tot.values())
    print(f"device time by operator class ({path}): total {total / 1e9:.3f} s")
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
        per = f"  {v / 1e6 / steps:7.3f} ms/step" if steps else ""
        print(f"  {k:52s} {v / 1e9:8.3f} s  {100 * v / total:5.1f} %  {calls[k]:8d} launches{per}")


main()
