#!/bin/bash
# round 5, GPU call 2: cross-process direct all-reduce tests, allreduce kernel check, stall counters of the main GEMM kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5b; mkdir -p $O
(timeout 600 python -m pytest tests/test_direct_allreduce.py tests/test_kernels_gpu.py -m gpu -x -q -k "direct or optim" > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -4 $O/tests.log)
timeout 300 python tools/stall_pmc.py run > $O/stall_plain.log 2>&1; echo "plain run rc $?"
PA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
PB="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
timeout 400 rocprofv3 --pmc $PA --kernel-trace --output-format csv -d $O/passA -- python tools/stall_pmc.py run > $O/passA.log 2>&1; echo "pass A rc $?"
timeout 400 rocprofv3 --pmc $PB --kernel-trace --output-format csv -d $O/passB -- python tools/stall_pmc.py run > $O/passB.log 2>&1; echo "pass B rc $?"
python tools/stall_pmc.py report $O/passA $O/passB > $O/stall_report.txt 2> $O/stall_report.err
rm -rf $O/passA/*/*.db 2>/dev/null
find $O -name "*.csv" -size +20M -delete
tail -60 $O/stall_report.txt
