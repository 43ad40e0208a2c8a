#!/bin/bash
# Round 4, first GPU call: hardware answers for everything round 3 shipped or staged blind -- in ONE process per configuration family
# (tools/ab_inproc.py: model built once, one hipGraph per setting, alternating replays).
O=gpurun_out; mkdir -p $O
SVDX_STAGED=1 timeout 420 python -m pytest tests/test_kernels_gpu.py -q -x -k "v27 or v28 or v29 or tn_v12 or tn_v13 or tn_v21 or small or batched_skinny" > $O/r4a_staged_checks.txt 2>&1; tail -n 3 $O/r4a_staged_checks.txt
timeout 600 python tools/ab_inproc.py --out $O/r4a_ab_c2.json -- base batch_small=0 dvec_from_dw=0 SVDX_GEGLU_TILE=sweep fuse_tsa=0 tuned > $O/r4a_ab_c2.txt 2>&1; grep -v "^\[" $O/r4a_ab_c2.txt | tail -n 16
timeout 420 python tools/ab_inproc.py --dtype bf16 --lora-rank 64 -- base lora_stack_da=0 batch_small=0 > $O/r4a_ab_c5.txt 2>&1; grep -v "^\[" $O/r4a_ab_c5.txt | tail -n 8
