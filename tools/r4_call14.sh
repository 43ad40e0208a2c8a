O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_gn or gemm_plain_v1 or gemm_plain_v4 or gemm_tn" 2>&1 | tail -n 3
python bench.py --steps 100 --warmup 3 --no-cpu-baseline --launch-log $O/launch_log_gn.json > $O/r4d_bench.json 2>$O/r4d_bench.err; python - <<'PY'
import json, collections
d = json.loads(open('gpurun_out/r4d_bench.json').read().strip().splitlines()[-1])
print('bench', d['ms_per_step'], d['config'].get('gpu_clock'), 'roofline', d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])
t = d['roofline'].get('temporal_self_attention')
if t:
    print('tsa op', t['fwd_frac_of_mfma_peak'], t['bwd_frac_of_mfma_peak'], t['op_frac_of_mfma_peak'], t['ms_per_step'])
    for r in t['levels']: print('  ', r['rows'], r['channels'], {w: (r[w]['blocks'], round(r[w]['ms'], 3), round(r[w]['frac_of_mfma_peak'], 4)) for w in ('fwd', 'bwd')})
l = json.load(open('gpurun_out/launch_log_gn.json'))
c = collections.Counter(x[0] for x in l)
print({k: c[k] for k in ('svdx_gn_stats', 'svdx_gemm_gn', 'svdx_gemm_finalize_gn', 'svdx_gemm', 'svdx_gemm_finalize', 'svdx_gn_apply')}, len(l))
PY
timeout 600 python tools/ab_inproc.py --reps 3 -- base fuse_gn_stats=0 > $O/r4d_ab_gn.txt 2>&1; grep -v "^\[" $O/r4d_ab_gn.txt | tail -n 3
bash tools/collect_evidence.sh r4b > $O/r4b_collect.log 2>&1; tail -n 30 $O/r4b_collect.log
