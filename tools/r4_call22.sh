export SVDX_GRAPH_KEEP_LOSS=0
echo "--- zero = kernel"
for c in noread read_other h2d_other plus1_big plus1_sync_before mm; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
echo "--- zero = hipMemsetAsync"
export SVDX_ZERO_MEMSET=1
for c in noread read_other; do python tools/dbg_corrupt.py $c 2>&1 | tail -1; done
