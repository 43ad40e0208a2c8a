one() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(round(d['ms_per_step'],2), {k:c[k] for k in ('loss','loss_scale','opt_steps','exec')})"; }
echo default; one
echo no-gn; SVDX_FUSE_GN_STATS=0 one
echo no-keep-loss; SVDX_GRAPH_KEEP_LOSS=0 one
echo no-graph; one --no-graph
echo no-batch-small; SVDX_BATCH_SMALL=0 one
