O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "survive or graphed_step" 2>&1 | tail -n 4
SVDX_ZERO_MEMSET=1 timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "survive" 2>&1 | tail -n 6 | cut -c1-300
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "elementwise or groupnorm or gemm_gn" 2>&1 | tail -n 3
bash tools/r4_call18.sh
