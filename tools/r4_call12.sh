export SVDX_NO_PRIME=1
for p in kernel inplace alloc alloc_big svdx svdx_alloc; do python tools/dbg_timing.py --warm 5 --chunk 20 --nchunks 2 --post $p 2>&1 | tail -1 | sed 's/.*post.: //'; done
