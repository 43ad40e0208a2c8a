#!/bin/bash
# usage: tools/ab_commits.sh <dirA> <dirB> [bench flags]   -- alternate `python bench.py` of two checkouts of this repo on one box
# (box-to-box spread is +-3 %: every step-time comparison has to come from one gpurun call); prints ms/step per run
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for d in "$A" "$B"; do
    ms=$(cd "$d" && python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$d] $ms"
  done
done
