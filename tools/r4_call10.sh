for p in sync_stream kernel d2h d2h_nb h2d event; do python tools/dbg_timing.py --warm 5 --chunk 20 --nchunks 2 --post $p 2>&1 | tail -1 | sed 's/.*post.: //'; done
python tools/dbg_timing.py --warm 5 --chunk 20 --nchunks 2 --loss-print --long 300 2>&1 | tail -1 | sed 's/.*post.: //'
