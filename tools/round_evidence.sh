#!/bin/bash
# Everything a round's DESIGN.md section 6 quotes, from ONE GPU box (run through gpurun; outputs under gpurun_out/<tag>_*; copy what is
# to be judged into profiles/).  About 8 GPU-minutes (the real loop -- new clip, VAE + CLIP + EDM prep, step -- is the `real_loop` object of every line): the default bench line (CPU-baseline legs included), the other configurations of
# BASELINE.json on one GPU, the in-step A/B of the default-on paths, then the profiler passes of tools/collect_evidence.sh.
# usage: gpurun --timeout 2400 -- 'bash tools/round_evidence.sh r4'
tag=${1:-rX}; O=gpurun_out; mkdir -p $O
python bench.py > $O/${tag}_bench_default_run.json 2> $O/${tag}_bench_default_run.err; tail -c 600 $O/${tag}_bench_default_run.json | head -c 300; echo
python bench.py --dtype bf16 --no-cpu-baseline > $O/${tag}_bench_bf16.json 2>/dev/null
python bench.py --lora-rank 64 --dtype bf16 --no-cpu-baseline > $O/${tag}_bench_c5.json 2>/dev/null
python bench.py --frames 25 --height 576 --width 1024 --grad-accum 2 --steps 20 --no-cpu-baseline > $O/${tag}_bench_c4.json 2>/dev/null
for f in bf16 c5 c4; do python - <<PY
import json
d = json.loads(open("$O/${tag}_bench_$f.json").read().strip().splitlines()[-1]); c = d["config"]; r = d.get("roofline") or {}
print("$f", round(d["ms_per_step"], 2), "ms", round(d["value"], 3), d["unit"], "loss", c["loss"], "steps", c["opt_steps"], "frac", r.get("frac"), (d.get("with_vae") or {}).get("ms_per_step"))
PY
done
timeout 900 python tools/ab_inproc.py -- base fold_finite=0 batch_small=0 dvec_from_dw=0 fuse_tsa=0 fuse_gn_stats=0 defer_grad_finalize=0 > $O/${tag}_ab_c2.txt 2>&1; grep -v "^\[" $O/${tag}_ab_c2.txt | tail -n 10
bash tools/collect_evidence.sh $tag > $O/${tag}_collect.log 2>&1; tail -n 4 $O/${tag}_collect.log
