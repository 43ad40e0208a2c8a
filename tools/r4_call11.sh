O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_gn" 2>&1 | tail -n 3
timeout 600 python tools/ab_inproc.py --reps 3 -- base fuse_gn_stats=0 > $O/r4c_ab_gn.txt 2>&1; grep -v "^\[" $O/r4c_ab_gn.txt | tail -n 5
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-roofline --launch-log $O/launch_log_gn.json 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('gpu_clock'))"
python - <<'PY'
import json, collections
l = json.load(open('gpurun_out/launch_log_gn.json'))
c = collections.Counter(x[0] for x in l)
print({k: c[k] for k in ('svdx_gn_stats', 'svdx_gemm_gn', 'svdx_gemm', 'svdx_gemm_finalize', 'svdx_gn_apply')}, len(l))
PY
timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x 2>&1 | tail -n 3
python bench.py --steps 100 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench 100 steps', d['ms_per_step'], d['config'].get('gpu_clock'))"
python tools/dbg_timing.py --warm 5 --chunk 20 --nchunks 2 2>&1 | tail -1 | sed 's/.*post.: //'
