O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_gn or gemm_tn" 2>&1 | tail -n 3
timeout 600 python tools/ab_inproc.py --reps 3 -- base fuse_gn_bwd_stats=0 fuse_gn_stats=0 > $O/r4g_ab.txt 2>&1; grep -v "^\[" $O/r4g_ab.txt | tail -n 8
