#!/bin/bash
# The round's bench lines from one box: default (with the CPU baseline legs), bf16, config 4 (25 x 1024 x 576, grad-accum 2), config 5
# (LoRA r = 64, bf16), c2 with the VAE encode beside the step, and north_star's schedule on one rank.  usage: tools/bench_lines.sh <tag>
tag=${1:-rX}
mkdir -p gpurun_out
O=gpurun_out
python bench.py > $O/${tag}_bench_default_run.json 2> $O/${tag}_bench_default_run.err
python bench.py --steps 40 --no-cpu-baseline --dtype bf16 > $O/${tag}_bench_bf16.json 2>> $O/${tag}_bench.err
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --frames 25 --height 576 --width 1024 --grad-accum 2 > $O/${tag}_bench_c4.json 2>> $O/${tag}_bench.err
python bench.py --steps 40 --no-cpu-baseline --lora-rank 64 --dtype bf16 > $O/${tag}_bench_c5.json 2>> $O/${tag}_bench.err
python bench.py --steps 40 --no-cpu-baseline --with-vae > $O/${tag}_bench_with_vae.json 2>> $O/${tag}_bench.err
python bench.py --steps 40 --no-cpu-baseline --overlap vae > $O/${tag}_bench_overlap_vae.json 2>> $O/${tag}_bench.err
for f in default_run bf16 c4 c5 with_vae overlap_vae; do echo "$f $(grep -o '"ms_per_step": [0-9.]*' $O/${tag}_bench_$f.json | head -2 | tr '\n' ' ')"; done
tail -n 3 $O/${tag}_bench.err
