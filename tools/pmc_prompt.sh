#!/bin/bash
# rocprofv3 counter passes over the prompt-attention kernel (tools/prof_prompt.py bfloat16 1 16384): one counter group
# per run, each within the hardware's slots, each under its own time limit.  usage: tools/pmc_prompt.sh <out dir> [dtype n len]
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/pmc_prompt}
DT=${2:-bfloat16}; NS=${3:-1}; LEN=${4:-16384}
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/prof_prompt.py "$DT" "$NS" "$LEN" > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; exit 1; }
  echo "pass $name done"
}
mkdir -p "$OUT"
pass p1 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES && \
pass p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS && \
pass p3 SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && \
python3 tools/summarize_pmc_counters.py "$OUT" pa_prompt_kernel
