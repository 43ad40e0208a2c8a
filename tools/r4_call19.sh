export SVDX_GRAPH_KEEP_LOSS=0
for c in noread noread none none; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('bench 4 steps', {k:c[k] for k in ('loss','loss_scale','opt_steps','exec')})"
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-roofline --no-graph 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print('bench 4 steps eager', {k:c[k] for k in ('loss','loss_scale','opt_steps','exec')})"
