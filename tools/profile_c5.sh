cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6y; mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr -- python bench.py --lora-rank 64 --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-real-loop --launch-log $O/launch_log.json > $O/bench.json 2> $O/err.txt
DB=$(ls $O/tr/*/*_results.db | head -1)
python tools/prof_summary.py $DB --last-fraction=0.40 > $O/kernel_stats.txt
python tools/step_trace.py $DB $O/launch_log.json $O/step_trace.json > $O/step_trace.txt 2> $O/step_trace.err
python tools/step_categories.py $O/step_trace.json > $O/step_categories.txt
rm -rf $O/tr
head -45 $O/step_categories.txt
