#!/bin/bash
# Cache policy of the whole-tile value update's loads and stores (variant builds ld<nt>st<nt>), tools/exp_cache_append.py
cd "$(dirname "$0")/.."
for v in tune ld0st1 ld1st0 ld0st0; do echo "== $v"; timeout -k 10 120 python -u tools/exp_cache_append.py $v 2>&1 | grep -v amdgpu; done
