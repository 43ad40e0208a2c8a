O=gpurun_out; mkdir -p $O
SVDX_ZERO_MEMSET=1 timeout 1200 python -m pytest tests/test_e2e_gpu.py -q -x -k "survive" 2>&1 | tail -n 6 | cut -c1-400
timeout 1200 python -m pytest tests/test_e2e_gpu.py -q -x -k "survive" 2>&1 | tail -n 3 | cut -c1-300
timeout 600 python tools/ab_inproc.py --reps 3 -- base defer_grad_finalize=0 fuse_gn_stats=0 > $O/r4f_ab.txt 2>&1; grep -v "^\[" $O/r4f_ab.txt | tail -n 8
