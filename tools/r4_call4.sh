O=gpurun_out; mkdir -p $O
ls /sys/class/drm/ | tr '\n' ' '; echo
python - <<'PY'
import torch
p=torch.cuda.get_device_properties(0)
print(p.name, getattr(p,'pci_bus_id',None), getattr(p,'pci_device_id',None), getattr(p,'pci_domain_id',None), p.multi_processor_count)
import glob
for f in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk'))[:10]:
    import os
    print(f, os.path.realpath(os.path.dirname(f)), [l.strip() for l in open(f)][:3])
PY
one() { python bench.py "$@" --no-cpu-baseline --no-roofline 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config'].get('gpu_clock'))"; }
echo "bench 30 sampler on"; one --steps 30 --warmup 3
echo "bench 30 sampler off"; SVDX_NO_CLOCK_SAMPLER=1 one --steps 30 --warmup 3
echo "bench 200 sampler off"; SVDX_NO_CLOCK_SAMPLER=1 one --steps 200 --warmup 5
echo "bench 200 sampler on"; one --steps 200 --warmup 5
timeout 300 python tools/ab_inproc.py --reps 2 -- base 2>&1 | grep -v "^\[" | tail -n 3
