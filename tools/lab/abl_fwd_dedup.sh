cd $GRAFT_REPO_ROOT
run() { # label, env...
  env "${@:2}" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms']['fwd_gemm'])")"
}
for lead in 0 1; do
  run "lead $lead full              " VV_FWD_LEAD=$lead
  run "lead $lead no stream (1)     " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=1
  run "lead $lead no MFMA (2)       " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=2
  run "lead $lead stream only (6)   " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=6
  run "lead $lead L2-hot rows (8)   " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=8
  run "lead $lead stream only L2-hot" VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=14
done
