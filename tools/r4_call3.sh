O=gpurun_out; mkdir -p $O
SVDX_STAGED=1 timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "tn_v32 or tn_v41 or tn_v13" 2>&1 | tail -n 3
timeout 600 python tools/ab_inproc.py --reps 2 --out $O/r4b_ab_tn.json -- base tuned > $O/r4b_ab_tn.txt 2>&1; grep -v "^\[" $O/r4b_ab_tn.txt | tail -n 6
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('gpu_clock'))"
