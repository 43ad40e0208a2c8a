#!/bin/bash
# usage: tools/ab_power.sh "<bench flags A>" "<bench flags B>"  -- alternate two bench configurations, sampling clocks/power meanwhile
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for mode in "$1" "$2"; do
    python bench.py $mode --steps 60 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/ab.log 2>&1 &
    pid=$!
    sleep 14
    for i in 1 2 3; do
      /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power \(W\)|Average Graphics|Socket" | tr -s ' ' | tr '\n' ';'
      echo
      sleep 1
    done
    wait $pid
    tail -1 /tmp/ab.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('[$mode]', round(r['ms_per_step'],3), 'ms')"
  done
done
