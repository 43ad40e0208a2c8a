#!/bin/bash
# round 5, call 8: the CLIP attention change on the GPU (kernel check + tower parity), then the whole evidence set again from one box
# with the profiler passes kept clear of the real loop's conditioners.
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_clip.py tests/test_kernels_gpu.py -m gpu -q -k "encode_image or encoders" > $O/r5_call8_tests.txt 2>&1; tail -n 3 $O/r5_call8_tests.txt
timeout 1500 bash tools/round_evidence.sh r5b
