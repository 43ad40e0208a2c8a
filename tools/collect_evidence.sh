#!/bin/bash
# Measurement artefacts of one round, from one GPU box (run through gpurun; copies land in gpurun_out/<tag>_*):
#   kernel stats of the default bench (rocprofv3 --kernel-trace), the per-call step trace, HBM traffic per kernel family (separate
#   FETCH_SIZE / WRITE_SIZE counter passes, MI355X_MICROARCH.md HBM section) and the in-step MFMA utilisation per kernel symbol.
# usage: tools/collect_evidence.sh <tag> [stats-only]      (counter passes never combine --pmc with sys / hip / hsa traces)
tag=${1:-rX}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
commit=$(git rev-parse --short HEAD 2>/dev/null || echo worktree)
# 1. kernel stats + step trace of the graph-replayed default step
rocprofv3 --kernel-trace -d $O/ev_tr -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-real-loop --launch-log $O/launch_log.json > $O/${tag}_bench_under_profiler.json 2> $O/ev_tr.err
DB=$(ls $O/ev_tr/*/*_results.db | head -1)
{ echo "# rocprofv3 --kernel-trace -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-real-loop   (last 40 % of the dispatches = graph-replayed steps only; tools/prof_summary.py)"; python tools/prof_summary.py $DB --last-fraction=0.40; } > $O/${tag}_rocprofv3_kernel_stats.txt 2>> $O/ev_tr.err
python tools/step_trace.py $DB $O/launch_log.json $O/step_trace.json > $O/${tag}_step_trace.txt 2> $O/step_trace.err
python tools/step_categories.py $O/step_trace.json > $O/${tag}_step_categories.txt
rm -rf $O/ev_tr
if [ "$2" = "stats-only" ]; then head -8 $O/${tag}_rocprofv3_kernel_stats.txt; exit 0; fi
# 2. HBM traffic: two counter passes over two eager steps
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/ev_$c -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-real-loop > /dev/null 2> $O/ev_$c.err
done
F=$(ls $O/ev_FETCH_SIZE/*/*counter_collection.csv | head -1); W=$(ls $O/ev_WRITE_SIZE/*/*counter_collection.csv | head -1)
python tools/pmc_traffic.py $F $W $O/${tag}_pmc_traffic.json > $O/${tag}_pmc_traffic.txt 2> $O/pmc_traffic.err
python tools/pmc_traffic.py $F $W --by-grid --by-shape $O/launch_log.json > $O/${tag}_pmc_traffic_by_grid.txt 2>> $O/pmc_traffic.err
rm -rf $O/ev_FETCH_SIZE $O/ev_WRITE_SIZE
# 3. MFMA-pipe busy cycles per kernel symbol inside the step
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/ev_mfma -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-real-loop > /dev/null 2> $O/ev_mfma.err
M=$(ls $O/ev_mfma/*/*counter_collection.csv | head -1)
{ echo "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-real-loop"; python tools/pmc_mfma.py $M $O/${tag}_pmc_mfma_in_step.json; } > $O/${tag}_pmc_mfma_in_step.txt 2> $O/pmc_mfma.err
rm -rf $O/ev_mfma
head -12 $O/${tag}_step_categories.txt; head -8 $O/${tag}_pmc_traffic.txt; head -12 $O/${tag}_pmc_mfma_in_step.txt; for f in step_trace pmc_traffic pmc_mfma; do tail -n 2 $O/$f.err; done
