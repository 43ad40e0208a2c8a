#!/bin/bash
# round 5, last call: the driver's own checks on the closing commit -- smoke(), and the GPU tests that touch what changed last.
O=gpurun_out; mkdir -p $O
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r5_final_smoke.txt 2>&1; tail -n 2 $O/r5_final_smoke.txt
timeout 200 python -m pytest tests -m gpu -q -x -k "encoders or optim or small or capi or smoke or encode_image or tiny" > $O/r5_final_gpu_subset.txt 2>&1; tail -n 3 $O/r5_final_gpu_subset.txt
