#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5h; mkdir -p $O
(timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -x -q -k "gemm_tn or optim or matches_oracle_tiny or graphed_step or same_seed or resume" > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -4 $O/tests.log)
timeout 900 python tools/ab_inproc.py --steps 20 --reps 3 -- base fold_finite=0 > $O/ab.log 2>&1; echo "ab rc $?"; grep -v "^\[" $O/ab.log | tail -6
