export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "
import importlib.util
for m in ('diffusers','peft','accelerate','transformers'):
    s = importlib.util.find_spec(m)
    print(m, s is not None)
" > gpurun_out/r3_probe_imports.txt 2>&1
rocprofv3 --kernel-trace -d gpurun_out/tr1 -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --launch-log gpurun_out/launch_log.json > gpurun_out/tr1_bench.json 2> gpurun_out/tr1.err
DB=$(ls gpurun_out/tr1/*/*_results.db | head -1)
python tools/step_trace.py $DB gpurun_out/launch_log.json gpurun_out/step_trace.json > gpurun_out/step_trace.txt 2> gpurun_out/step_trace.err
rm -rf gpurun_out/tr1
cat gpurun_out/r3_probe_imports.txt; cat gpurun_out/step_trace.err; head -30 gpurun_out/step_trace.txt
