export SVDX_GRAPH_KEEP_LOSS=0
for c in noread sync_only read_other read_pflat read_batch h2d_other read_pinned none; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
