cd $GRAFT_REPO_ROOT
# needs the lab build of the library (make -C videovector_amd/csrc lab): the ablated kernels are not in the product library
export VV_LIB=${GRAFT_REPO_ROOT:-/root/repo}/videovector_amd/lib/libvideovec_lab.so
run() { # label, env...
  env "${@:2}" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), 'fwd', d['kernels_ms']['fwd_gemm'], 'wgrad', d['kernels_ms']['wgrad_gemm'])")"
}
for r in 1 2; do
run "full                           " A=1
run "fwd: no stores (64)            " VV_LAB_FWD_ABL=64
run "fwd: no stream no MFMA (3)     " VV_LAB_FWD_ABL=3
run "fwd: + no stores (67)          " VV_LAB_FWD_ABL=67
run "wgrad: no stores (64)          " VV_LAB_WG_ABL=64
run "wgrad: nothing, no stores (71) " VV_LAB_WG_ABL=71
done
