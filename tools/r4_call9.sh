python tools/dbg_timing.py --warm 10 --chunk 30 --nchunks 2 2>&1 | tail -1
python tools/dbg_timing.py --warm 10 --chunk 30 --nchunks 2 --loss-print 2>&1 | tail -1
python tools/dbg_timing.py --warm 10 --chunk 30 --nchunks 2 --apply 2>&1 | tail -1
python tools/ab_inproc.py --reps 2 -- base 2>&1 | grep -v "^#" | tail -n 1
