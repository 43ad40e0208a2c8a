#!/bin/bash
# Two ranks sharing ONE GPU over gloo (RCCL refuses two ranks on a device): rehearses the multi-rank paths of bench.py -- graph chain,
# buckets / single collective / the collective under the next clip's VAE encode -- end to end on the box this round has.
# usage: tools/rehearse_2rank.sh <out-prefix> [extra bench flags]
out=$1; shift
for mode in buckets single vae; do
  SVDX_DIST_BACKEND=gloo SVDX_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port $((29600 + RANDOM % 300)) bench.py --gpus 2 --steps 5 --warmup 2 --tiny --no-cpu-baseline --no-roofline --overlap $mode "$@" \
    2> ${out}_${mode}.err | grep '^{' > ${out}_${mode}.json
  python - "$mode" "${out}_${mode}.json" <<'PY'
import json, sys
r = json.load(open(sys.argv[2]))
c = r["config"]
print(sys.argv[1], "ms/step", round(r["ms_per_step"], 2), "| ranks", c["ranks_seen"], "|", c["grad_allreduce"], "|", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in (r.get("with_vae") or {}).items() if k in ("ms_per_step", "allreduce_ms_exposed", "vae_ms_alone")})
PY
done
