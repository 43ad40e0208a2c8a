#!/bin/bash
# (as run, the profiler line carried --stats: rocprofv3 wrote its database and then sat in post-processing until the 600 s limit; fixed below)
# round 5, call 9: the matrix-pipe CLIP attention on the GPU: kernel check, tower parity, per-kernel split of the tower, the real loop.
O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_clip.py tests/test_kernels_gpu.py tests/test_train_loop.py -m gpu -q -k "encode_image or encoders or loop or example" > $O/r5_call9_tests.txt 2>&1; tail -n 3 $O/r5_call9_tests.txt
timeout 300 python tools/clip_bench.py > $O/r5_clip_bench.txt 2>&1; tail -n 1 $O/r5_clip_bench.txt
timeout 600 rocprofv3 --kernel-trace -d $O/ev_clip -- python tools/clip_bench.py --iters 5 > /dev/null 2> $O/ev_clip.err
python tools/prof_summary.py $(ls $O/ev_clip/*/*_results.db | head -1) | head -12 >> $O/r5_clip_bench.txt; rm -rf $O/ev_clip; tail -n 12 $O/r5_clip_bench.txt
timeout 900 python bench.py --no-cpu-baseline --no-roofline --steps 100 > $O/r5_call9_bench.json 2> $O/r5_call9_bench.err; tail -c 900 $O/r5_call9_bench.json | head -c 500; echo
