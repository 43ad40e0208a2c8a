#!/bin/bash
# round 5, GPU call 3: the TN kernels with loads and MFMAs overlapped (hidden LDS-DMA) + L2 prefetch -- kernel checks, in-step A/B, bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5c; mkdir -p $O
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_loop.py -m gpu -x -q -k "gemm_tn or example_script" > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -4 $O/tests.log)
timeout 900 python tools/ab_inproc.py --steps 20 --reps 3 --out $O/ab.json -- base tn_flat=1 tn_prefetch=0 tuned > $O/ab.log 2>&1; echo "ab rc $?"; grep -v "^\[" $O/ab.log | tail -12
timeout 600 python bench.py --steps 30 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5c/bench.json').read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"]["kernel_ms_per_step"], "tsa", d["roofline"]["temporal_self_attention"]["op_frac_of_mfma_peak"], "real_loop", d["real_loop"])
PY
