export SVDX_GRAPH_KEEP_LOSS=0
for c in none none none plus1 plus1 plus1_sync_before plus1_sync_before plus1_big mm; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
echo "--- fuse_gn_stats=0"
export SVDX_FUSE_GN_STATS=0
for c in none none plus1 plus1_sync_before plus1_big; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
