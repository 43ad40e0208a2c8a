export SVDX_GRAPH_KEEP_LOSS=0 TMPDIR=/tmp
O=gpurun_out
rocprofv3 --kernel-trace -d $O/dbgtr -- python tools/dbg_trace.py > $O/dbgtr.log 2>&1; tail -n 2 $O/dbgtr.log
DB=$(ls $O/dbgtr/*/*_results.db | head -1)
python tools/dbg_trace_cmp.py $DB
rm -rf $O/dbgtr
