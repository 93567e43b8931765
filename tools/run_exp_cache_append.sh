#!/bin/bash
# Sweeps the launch shape of the whole-tile value update (tools/exp_cache_append.py) on a `make variant NAME=tune` build.
set -e
cd "$(dirname "$0")/.."
for cfg in "4 2048" "8 2048" "8 1024" "16 1024" "16 512" "4 1024" "16 2048" "8 4096"; do
  set -- $cfg
  PA_CACHE_APPEND_IN_FLIGHT=$1 PA_CACHE_APPEND_WORKGROUPS=$2 timeout -k 10 120 python -u tools/exp_cache_append.py tune
done
