#!/bin/bash
# Launch forms of small per-call decode batches (tools/exp_small_batch_forms.py) on a `make variant NAME=tune` build.
set -e
cd "$(dirname "$0")/.."
timeout -k 10 100 python -u tools/exp_small_batch_forms.py default
timeout -k 10 100 python -u tools/exp_small_batch_forms.py tune
PA_DECODE_SPLIT_TOKENS=131072 timeout -k 10 100 python -u tools/exp_small_batch_forms.py tune
PA_DECODE_SPLIT_TOKENS=65536 timeout -k 10 100 python -u tools/exp_small_batch_forms.py tune
PA_DECODE_SPLIT_TOKENS=131072 PA_DECODE_TARGET_WGS=512 timeout -k 10 100 python -u tools/exp_small_batch_forms.py tune
PA_DECODE_SPLIT_TOKENS=131072 PA_DECODE_TARGET_WGS=1024 timeout -k 10 100 python -u tools/exp_small_batch_forms.py tune
PA_DECODE_SPLIT_ROWS=128 timeout -k 10 100 python -u tools/exp_small_batch_forms.py tune
