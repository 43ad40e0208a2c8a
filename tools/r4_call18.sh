O=gpurun_out; mkdir -p $O
export SVDX_GRAPH_KEEP_LOSS=0
timeout 900 python tools/ab_inproc.py --out $O/r4e_ab_c2.json -- base batch_small=0 dvec_from_dw=0 SVDX_GEGLU_TILE=sweep fuse_tsa=0 fuse_gn_stats=0 tuned > $O/r4e_ab_c2.txt 2>&1; grep -v "^\[" $O/r4e_ab_c2.txt | tail -n 20
timeout 600 python tools/ab_inproc.py --dtype bf16 --lora-rank 64 -- base lora_stack_da=0 batch_small=0 > $O/r4e_ab_c5.txt 2>&1; grep -v "^\[" $O/r4e_ab_c5.txt | tail -n 9
