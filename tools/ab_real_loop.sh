cd $GRAFT_REPO_ROOT
for i in 1 2; do for f in "" "${AB_FLAG:---serial-optimizer}"; do
python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline $f 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['real_loop']; print('[$f]', round(d['ms_per_step'],2), 'real', round(r['ms_per_step'],2), 'cond alone', round(r['conditioners_ms_alone'],2), r['loss_last'])"
done; done
