cd $GRAFT_REPO_ROOT
# needs the lab build of the library (make -C videovector_amd/csrc lab): the ablated kernels are not in the product library
export VV_LIB=${GRAFT_REPO_ROOT:-/root/repo}/videovector_amd/lib/libvideovec_lab.so
run() { # label, env...
  env "${@:2}" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms']['fwd_gemm'])")"
}
VV_FWD_LEAD=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py tests/test_gpu_cfg5.py tests/test_gpu_shipped.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do
  run "lead 0 full" VV_FWD_LEAD=0
  run "lead 1 full" VV_FWD_LEAD=1
done
for lead in 0 1; do
  run "lead $lead no stream (1)     " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=1
  run "lead $lead stream only (6)   " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=6
  run "lead $lead L2-hot rows (8)   " VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=8
  run "lead $lead stream only L2-hot" VV_FWD_LEAD=$lead VV_LAB_FWD_ABL=14
done
