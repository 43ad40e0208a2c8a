#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python tools/ab_inproc.py --steps 20 --reps 3 --out $O/ab.json -- base tuned tn_flat=1 > $O/ab.log 2>&1; echo "ab rc $?"; grep -v "^\[" $O/ab.log | tail -8
