#!/bin/bash
# Opening GPU call of the round after a round that ended without GPU minutes: everything round 3 staged on the simulator (tests/sim) and
# every default it switched on from isolated evidence gets its hardware answer in ONE box (outputs under gpurun_out/r4a_*):
#   1. the staged tile variants (ops.STAGED_TILES: 27 / 28 / 29) through the kernel checks on the GPU
#   2. race screen + isolated timing of every tile variant incl. the staged ones on the step's problems (tools/ring_check.py)
#   3. in-step A/B of the GEGLU tile rule (variant 26 by default since 812a155, never timed inside the step)
#   4. in-step A/B of the table-driven skinny launches (Runtime.batch_small, on by default since the end of round 3: 176 -> 5 launches per
#      step, bit-identical on the simulator, never run on the GPU) -- preceded by their kernel checks and the GPU bit-equality test
#   5. the in-situ sweep with the staged candidates (bench.py --tune): per-problem winners inside the real step
# usage (about 20 GPU-minutes): gpurun --timeout 1500 -- 'bash tools/r4_first_call.sh'
O=gpurun_out; mkdir -p $O
SVDX_STAGED=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "v27 or v28 or v29 or tn_v12 or tn_v13 or tn_v21" > $O/r4a_staged_checks.txt 2>&1; tail -n 3 $O/r4a_staged_checks.txt
timeout 540 python tools/ring_check.py race time tn > $O/r4a_ring_race_time.txt 2>&1; grep -E "SUMMARY|L0 conv 320|L0 dx K2560|L0 ff2|L0 proj|L0 qkv|tn R= 35840" $O/r4a_ring_race_time.txt | cut -c1-260
bash tools/ab_env.sh SVDX_GEGLU_TILE=default SVDX_GEGLU_TILE=sweep > $O/r4a_ab_geglu_tile.txt 2>&1; cat $O/r4a_ab_geglu_tile.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -q -k "small or layernorm or batched_skinny" > $O/r4a_batch_small_checks.txt 2>&1; tail -n 3 $O/r4a_batch_small_checks.txt
bash tools/ab_env.sh SVDX_BATCH_SMALL=1 SVDX_BATCH_SMALL=0 > $O/r4a_ab_batch_small.txt 2>&1; cat $O/r4a_ab_batch_small.txt
bash tools/ab_env.sh SVDX_DVEC_FROM_DW=1 SVDX_DVEC_FROM_DW=0 > $O/r4a_ab_dvec_from_dw.txt 2>&1; cat $O/r4a_ab_dvec_from_dw.txt
bash tools/ab_env.sh SVDX_LORA_STACK_DA=1 SVDX_LORA_STACK_DA=0 --lora-rank 64 --dtype bf16 > $O/r4a_ab_lora_stack_da.txt 2>&1; cat $O/r4a_ab_lora_stack_da.txt
bash tools/ab_env.sh SVDX_BATCH_SMALL=1 SVDX_BATCH_SMALL=0 --lora-rank 64 --dtype bf16 > $O/r4a_ab_batch_small_c5.txt 2>&1; cat $O/r4a_ab_batch_small_c5.txt
timeout 600 python bench.py --tune --tune-rounds 2 --steps 40 --warmup 3 --no-cpu-baseline > $O/r4a_bench_tuned.json 2> $O/r4a_bench_tuned.err; grep -o '"ms_per_step": [0-9.]*' $O/r4a_bench_tuned.json | head -1
cp $O/gemm_tuned.json $O/r4a_gemm_tuned.json 2>/dev/null
python bench.py --steps 40 --warmup 3 --no-cpu-baseline | grep -o '"ms_per_step": [0-9.]*' | head -1
