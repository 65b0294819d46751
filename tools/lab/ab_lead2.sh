cd $GRAFT_REPO_ROOT
run() { # label, args, env...
  env "${@:3}" timeout 400 python bench.py $2 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'])")"
}
for r in 1 2; do for v in 0 1; do
  run "dense lead $v  " "--dedup off --steps 200 --warmup 20" VV_FWD_LEAD=$v
  run "cfg5 lead $v   " "--workload cfg5 --steps 40 --warmup 5" VV_FWD_LEAD=$v
  run "shipped lead $v" "--workload shipped --steps 200 --warmup 20" VV_FWD_LEAD=$v
  run "bf16 lead $v   " "--prec bf16 --steps 200 --warmup 20" VV_FWD_LEAD=$v
done; done
