#!/bin/bash
# Uniform partitions forced on batches of one to three residencies of rows (`make variant NAME=tune`; tools/ab_decode.py --planned):
# does a partition count that fills whole rounds of the chip's 768 workgroups beat the shipped rule?
cd "$(dirname "$0")/.."
SH=bs12,bs24,bs32,bs48,bs64,bs16_ctx4096,bs32_ctx2048,bs40_ctx4096
echo "== shipped rule"; timeout -k 10 200 python -u tools/ab_decode.py --libs tune --planned --shapes $SH --rounds 5 2>&1 | grep -v amdgpu
for p in 2 3 4 6; do echo "== PA_DECODE_FORCE_PARTS=$p"; PA_DECODE_FORCE_PARTS=$p timeout -k 10 200 python -u tools/ab_decode.py --libs tune --planned --shapes $SH --rounds 5 2>&1 | grep -v amdgpu || exit 1; done
