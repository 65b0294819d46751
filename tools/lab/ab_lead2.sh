cd $GRAFT_REPO_ROOT
run() { # label, args, env...
  env "${@:3}" timeout 400 python bench.py $2 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'])")"
}
for r in 1 2; do for v in 0 2; do
  run "dense 192-row tiles (3 rounds) lead $v" "--dedup off --steps 200 --warmup 20" VV_FWD_LEAD=$v VV_PH_MQ=3
  run "dense 256-row tiles (2 rounds) lead $v" "--dedup off --steps 200 --warmup 20" VV_FWD_LEAD=$v
done; done
