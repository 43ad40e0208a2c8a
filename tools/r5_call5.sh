#!/bin/bash
# round 5, GPU call 5: epilogue operand preloads (GEGLU backward, residual adds) -- kernel checks, then old / new library alternately in one box
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5e; mkdir -p $O
(timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log)
for rep in 1 2 3; do
  for lib in gpurun_ab/libsvdx_before_epilogue_preload.so svd_xtend_amd/csrc/libsvdx.so; do
    ms=$(SVDX_LIB=$PWD/$lib timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-roofline --no-real-loop 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$lib] $ms" | tee -a $O/ab_lib.txt
  done
done
