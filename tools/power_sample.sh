#!/bin/bash
# Samples the shader clock and socket power while the default bench step replays (is the steady state power-limited?).
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-roofline > /tmp/ps_bench.json 2>/dev/null &
pid=$!
sleep 9
for i in $(seq 1 12); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket|Average Graphics" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
wait $pid
grep -o '"ms_per_step": [0-9.]*' /tmp/ps_bench.json
/opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket|Average Graphics" | tr -s ' ' | tr '\n' ';'; echo " (idle)"
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -2
