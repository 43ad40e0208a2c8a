O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_tn" 2>&1 | tail -n 3
python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -n 1 > $O/r4h_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4h_bench.json')); c=d['config']; r=d['roofline']; print('bench', d['ms_per_step'], c['loss'], c['opt_steps'], c['gpu_clock']); print('roofline', r['frac'], r['kernel_ms_per_step'], r['kernel_ms_per_step_events'], r['clock'][:40]); t=r['temporal_self_attention']; print('tsa', t['op_frac_of_mfma_peak'], t['ms_per_step'])"
bash tools/collect_evidence.sh r4d > $O/r4d_collect.log 2>&1; grep -E "weight-grad|joined" $O/r4d_step_categories.txt; grep -E "^\('tn'" $O/r4d_pmc_traffic_by_grid.txt | head -8; tail -n 2 $O/r4d_pmc_traffic_by_grid.txt
