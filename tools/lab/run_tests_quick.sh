cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do for v in 0 1; do
VV_FWD_LEAD=$v timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lead $v', round(d['ms_per_step'],4), d['kernels_ms'], d['final_loss'])"
done; done
