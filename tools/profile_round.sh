#!/bin/bash
# The rocprofv3 passes behind roofline.* of the bench line, for a box with one MI355X:
#   tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/{trace,iso,pmc_fetch,pmc_write}/...
# then, where profiles/ is kept:    python tools/summarize_profiles.py gpurun_out/prof_TAG TAG
# Counters are collected in runs of their own (one counter each), the kernel traces without counters.
set -u
TAG=${1:?tag}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# 1. kernel trace of the default command (without the CPU baseline and the secondary measurements: same kernels, shorter)
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { echo "kernel trace failed"; tail -5 "$OUT/trace.err"; exit 1; }
# 2. isolated decode launches: the per-dispatch median of this trace is what roofline.frac is checked against
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/iso" -- python3 "$ROOT/bench.py" --roofline-only > "$OUT/decode_isolated_bench.json" 2> "$OUT/iso.err" || { echo "isolated trace failed"; tail -5 "$OUT/iso.err"; exit 1; }
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE, each in its own run, on 4 layers
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --layers 4 > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { echo "FETCH_SIZE pass failed"; tail -5 "$OUT/pmc_fetch.err"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --layers 4 > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || { echo "WRITE_SIZE pass failed"; tail -5 "$OUT/pmc_write.err"; exit 1; }
# keep what is small: the stats, and the counter rows of this library's kernels
find "$OUT" -name "*_kernel_trace.csv" -size +20M -delete
for f in $(find "$OUT" -name "*_counter_collection.csv"); do grep -E '^"?Correlation_Id|pa::' "$f" > "$f.pa" && mv "$f.pa" "$f"; done
du -sh "$OUT"
python3 "$ROOT/tools/kernel_trace_median.py" "$OUT/iso" "$OUT/decode_isolated_kernel_stats.csv" > /dev/null
python3 "$ROOT/tools/kernel_trace_median.py" "$OUT/trace" "$OUT/bench_kernel_medians.csv" > /dev/null
head -4 "$OUT/decode_isolated_kernel_stats.csv"
