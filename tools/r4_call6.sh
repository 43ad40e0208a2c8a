one() { python bench.py "$@" --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py', d['ms_per_step'], d['config'].get('gpu_clock'))"; }
one --steps 30 --warmup 3
python tools/dbg_timing.py bench 2>&1 | tail -2
timeout 300 python tools/ab_inproc.py --reps 2 -- base 2>&1 | grep -v "^\[" | tail -n 1
python tools/dbg_timing.py ab 2>&1 | tail -2
one --steps 30 --warmup 3
