#!/bin/bash
# config 4 (25 x 1024 x 576, grad-accum 2) alternately with two settings of a Runtime switch, separate processes (two c4 graphs do not fit one):
#   gpurun -- 'AB_FLAG="--rt big_m_rules=0" bash tools/ab_c4_tune.sh'        (default AB_FLAG: --tune, the in-situ tuner's table)
cd ${GRAFT_REPO_ROOT:-.}
for i in 1 2; do for f in "" "${AB_FLAG:---tune}"; do
python bench.py --frames 25 --height 576 --width 1024 --grad-accum 2 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-real-loop $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$f]', round(d['ms_per_step'],2), d['config'].get('gemm_tuning_sweeps'), d['config']['loss'])"
done; done
