#!/bin/bash
# Prompt attention: 128-row workgroups with one or two workgroups per CU (tuning build `tune`).
cd "$(dirname "$0")/.."
S=5x512,1x512,2x1024,1x2560,16x1024,8x2048,1x16384
echo "== product build (its own choice of 4 or 8 waves) vs tuning build, 4 waves forced, two workgroups per CU"
PA_PROMPT_WAVES=4 timeout -k 10 200 python -u tools/ab_prompt.py --libs default,tune --shapes $S
echo "== tuning build, 4 waves forced, ONE workgroup per CU (LDS padded)"
PA_PROMPT_WAVES=4 PA_PROMPT_LDS_PAD=24576 timeout -k 10 200 python -u tools/ab_prompt.py --libs default,tune --shapes $S
