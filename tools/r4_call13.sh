export SVDX_NO_PRIME=1
for p in zeros_new other_plus1 clone last_loss last_loss_nosync big_plus1 each_step; do python tools/dbg_timing.py --warm 5 --chunk 20 --nchunks 2 --post $p 2>&1 | tail -1 | sed 's/.*post.: //'; done
