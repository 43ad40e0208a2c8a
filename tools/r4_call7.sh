python tools/dbg_timing.py --warm 0 --chunk 10 2>&1 | tail -1
python tools/dbg_timing.py --warm 10 --chunk 30 --nchunks 3 2>&1 | tail -1
python tools/dbg_timing.py --warm 0 --chunk 30 --nchunks 3 2>&1 | tail -1
python tools/dbg_timing.py --warm 10 --chunk 10 2>&1 | tail -1
timeout 300 python tools/ab_inproc.py --reps 2 -- base 2>&1 | grep -v "^#" | tail -n 3
timeout 300 python tools/ab_inproc.py --reps 2 --steps 10 -- base 2>&1 | grep -v "^#" | tail -n 3
