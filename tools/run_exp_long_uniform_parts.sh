#!/bin/bash
# Equal long contexts on batches of one to five residencies of rows: the item list's partitions of 1024 tokens (shipped) against two or three
# uniform partitions (`make variant NAME=tune`; tools/ab_decode.py --planned).
cd "$(dirname "$0")/.."
SH=bs24_ctx4096,bs32_ctx4096,bs40_ctx4096,bs64_ctx4096,bs32_ctx2048,bs48_ctx2048,bs64_ctx2048,bs128_ctx2048,bs32_ctx8192,bs40_ctx4096_ragged
echo "== shipped rule"; timeout -k 10 300 python -u tools/ab_decode.py --libs tune --planned --shapes $SH --rounds 5 2>&1 | grep -v amdgpu
for p in 2 3; do echo "== PA_DECODE_FORCE_PARTS=$p"; PA_DECODE_FORCE_PARTS=$p timeout -k 10 300 python -u tools/ab_decode.py --libs tune --planned --shapes $SH --rounds 5 2>&1 | grep -v amdgpu || exit 1; done
