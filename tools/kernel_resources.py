#!/usr/bin/env python3
"""Registers, spills and scratch of every kernel in a HIP source, from hipcc's own remarks.

    python tools/kernel_resources.py vllm-v0.1.0_amd/csrc/pa_prefill.hip [filter] [-- extra hipcc flags]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    extra = []
    if "--" in argv:
        k = argv.index("--")
        argv, extra = argv[:k], argv[k + 1:]
    src = argv[0]
    flt = argv[1] if len(argv) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-ffast-math",
           "-fno-finite-math-only", "-I" + os.path.join(ROOT, "include"), "-Rpass-analysis=kernel-resource-usage",
           "-c", src, "-o", "/dev/null"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for line in err.splitlines():
        m = re.search(r"remark: .*?:\d+:\d+:\s+(.*?) \[-Rpass-analysis", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:") or text.startswith("Name:"):
            cur = text.split(":", 1)[1].strip()
            rows[cur] = {}
        elif cur and ":" in text:
            k, v = text.split(":", 1)
            rows[cur][k.strip()] = v.strip()
    for name, r in rows.items():
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0]
        if flt and flt not in dem:
            continue
        print(f"{dem:60s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} spill {r.get('VGPRs Spill', '?'):>4} "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>5} occ {r.get('Occupancy [waves/SIMD]', '?')}")


main()
