#!/usr/bin/env bash
# The commands behind the measured numbers of this repository, for a box with one MI355X.
# Each writes its output under gpurun_out/; the copies kept in profiles/ were
# made from those files.
set -euo pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()"
python -u bench.py                                  | tee gpurun_out/bench.json            # profiles/r01r_bench.json
python -u tools/sweep_decode.py --graph             | tee gpurun_out/sweep_decode.log      # profiles/r01g_decode_sweep.txt
python -u tools/read_bw.py --build && python -u tools/read_bw.py | tee gpurun_out/read_bw.log   # profiles/r01j_hbm_read_ceiling.txt
python -u tools/bench_cache_ops.py                  | tee gpurun_out/bench_cache.log       # profiles/r01f_cache_ops_bench.txt
python -u tools/bench_swap.py                       | tee gpurun_out/bench_swap.log
python -u tools/bench_elementwise.py                | tee gpurun_out/bench_elementwise.log # profiles/r01d_elementwise_bench.txt
python -u tools/bench_prompt.py                     | tee gpurun_out/bench_prompt.log      # profiles/r01e_prompt_attention_bench.txt
python -u tools/bench_decoder_step.py               | tee gpurun_out/decoder_step.log      # profiles/r01h_decoder_step.txt
python -u tools/exp_step.py                         | tee gpurun_out/exp_step.log          # reshape_and_cache -> decode interaction
python -u tools/exp_ragged_balance.py               | tee gpurun_out/exp_balance.log       # XCD imbalance of ragged batches (add --lognormal)
for m in "PA_DECODE_BALANCE=0" "PA_DECODE_BALANCE=4" "X=1 --planned"; do                   # profiles/r01n_decode_balance.txt
  set -- $m; env $1 python -u tools/sweep_decode.py ${2:-} | tee -a gpurun_out/decode_balance.log
done
python -u tools/bench_throughput.py --json gpurun_out/throughput_7b.json                   # profiles/r01m_throughput_llama7b.json
#   rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tp -o tp -- python3 tools/bench_throughput.py
#   python tools/summarize_kernel_classes.py /tmp/prof_tp/tp_kernel_stats.csv 2326          # profiles/r01m_throughput_kernel_classes.txt
# rocprofv3 passes (kernel trace, then FETCH_SIZE and WRITE_SIZE each in a run of its own):
#   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline
#   rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers 4
#   rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --layers 4
#   python tools/summarize_profiles.py gpurun_out/prof <tag>
# Later measurements:
python -u tools/sweep_decode.py --planned --graph   | tee gpurun_out/sweep_decode_planned.log   # profiles/r02p_decode_sweep.txt
python -u tools/bench_append_decode.py              | tee gpurun_out/append_decode.log          # profiles/r02u_append_form_experiments.txt
python -u tools/bench_rotary_cache.py               | tee gpurun_out/rotary_cache.log           # profiles/r02n_rotary_cache.txt
python -u tools/exp_cache_placement.py              | tee gpurun_out/cache_placement.log        # profiles/r02z_cache_placement.txt
python -u tools/exp_all_reduce_beside_decode.py     | tee gpurun_out/all_reduce_beside.log      # profiles/r02z_all_reduce_beside_decode.txt
python -u tools/bench_prompt.py                     | tee gpurun_out/bench_prompt_r02.log       # profiles/r02zz_prompt_attention_bench.txt
python -u bench.py --roofline-only                  | tee gpurun_out/bench_roofline_only.json   # isolated decode launches (rocprofv3 median: tools/kernel_trace_median.py)
#   several ranks sharing the one card (rendezvous over gloo, peers through IPC):
#   python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 tools/stress_push_all_reduce.py
#   A/B of library variants built with `make -C vllm-v0.1.0_amd/csrc variant NAME=x EXTRA_HIPFLAGS=-D...`:
#   python tools/ab_decode.py --libs default,x --planned ; python tools/ab_prompt.py --libs default,x
#   registers / spills / scratch of every kernel of a source file: python tools/kernel_resources.py vllm-v0.1.0_amd/csrc/pa_prefill.hip
#   PMC passes over the prompt kernel (profiles/r02zz_prompt_pmc_cycles.txt):
#   rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d /tmp/p1 -- python3 tools/prof_prompt.py bfloat16 1 16384
# Third batch of measurements:
bash tools/profile_round.sh r03x                                                                # the four rocprofv3 passes + summaries (profiles/r03a_*, *_latest)
bash tools/run_exp_cache_append.sh                  | tee gpurun_out/cache_append_shape.log     # profiles/r03_cache_append_shape.txt
bash tools/run_exp_cache_policy.sh                  | tee gpurun_out/cache_tile_policy.log      # profiles/r03_cache_tile_policy.txt (variant builds ld0st1, ld1st0, ld0st0)
python -u tools/sweep_small_batch.py                | tee gpurun_out/small_batch.log            # reference signature against the shared plan, batch 1-16
bash tools/run_exp_small_batch_forms.sh             | tee gpurun_out/small_batch_forms.log      # profiles/r03_small_batch_forms.txt
python -u tools/k_fragments_from_global.py          | tee gpurun_out/k_fragments.log            # profiles/r03_k_fragments_from_global_microbench.txt
bash tools/run_exp_prompt_occupancy.sh              | tee gpurun_out/prompt_occupancy.log       # profiles/r03_prompt_one_workgroup_per_cu_experiment.txt
python -u tools/prompt_phase_cycles.py              | tee gpurun_out/prompt_phase_cycles.log    # profiles/r03_prompt_phase_cycles.txt (variant build `timing`: -DPA_PROMPT_TIMING)
python -u tools/ab_prompt.py --libs default,ahead1  | tee gpurun_out/prompt_tiles_ahead.log     # profiles/r03_prompt_tiles_ahead_experiment.txt (variant -DPA_PROMPT_TILES_AHEAD=1)
python -u tools/ab_prompt.py --libs default --dtype float32 | tee gpurun_out/prompt_fp32.log    # profiles/r03_prompt_fp32_long_shapes.txt
python -u tools/bench_throughput.py --model 13b-tp2 --json gpurun_out/throughput_13b_tp2.json   # profiles/r03_throughput_llama13b_tp2_shard.json
python -u tools/bench_throughput.py --model neox20b-tp4 --n 3 --num-prompts 600 --num-blocks 9000 --num-cpu-blocks 3000 --json gpurun_out/throughput_neox_n3.json                                  # ..._n3_no_preemption.json
python -u tools/bench_throughput.py --model neox20b-tp4 --n 3 --num-prompts 600 --num-blocks 3500 --num-cpu-blocks 3000 --json gpurun_out/throughput_neox_swap.json  # ..._n3_swapping.json
python -u tests/soak_randomized.py                  > gpurun_out/soak.log                       # profiles/r03_soak_randomized.txt
bash tools/pmc_prompt.sh gpurun_out/pmc_prompt      | tee gpurun_out/pmc_prompt.log             # profiles/r03z_prompt_pmc_cycles.txt
#   work order and several-units-per-workgroup A/B (variant builds: -DPA_PROMPT_STAGGER=0, -DPA_PROMPT_PERSISTENT=1):
python -u tools/ab_prompt.py --libs default,nostagger,persist --shapes 1x16384,4x4096,8x2048,16x1024,32x512  # profiles/r03_prompt_work_order.txt, r03_prompt_persistent_units_experiment.txt
bash tools/run_exp_mid_batch_parts.sh; bash tools/run_exp_long_uniform_parts.sh    # profiles/r03_mid_batch_forced_partitions.txt, r03_equal_long_contexts_two_partitions.txt
