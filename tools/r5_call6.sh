#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
for rep in 1 2; do
for lib in gpurun_ab/libsvdx_before_epilogue_preload.so svd_xtend_amd/csrc/libsvdx.so; do
  echo "== $lib" | tee -a $O/time.txt
  SVDX_LIB=$PWD/$lib timeout 300 python tools/stall_pmc.py time 2>/dev/null | tee -a $O/time.txt
done
done
