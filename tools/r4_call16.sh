export SVDX_GRAPH_KEEP_LOSS=0
for c in none plus1 div_out plus1_out mul2 exp inplace svdx_add mm plus1_side plus1_big plus1_sync_before item; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
