#!/bin/bash
# usage: tools/ab_env.sh VAR=a VAR=b [bench flags]  -- alternate two settings of one environment knob on one box (3 x 40 steps each)
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for kv in "$A" "$B"; do
    ms=$(env "$kv" python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "[$kv] $ms"
  done
done
