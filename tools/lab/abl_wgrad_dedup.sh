# what the weight-gradient kernel is made of at the de-duplicated size (VV_LAB_WG_ABL: 1 no LDS-DMA stream, 2 no MFMA, 4 no fragment reads, 8 every row the zero row)
# needs the lab build of the library (make -C videovector_amd/csrc lab): the ablated kernels are not in the product library
export VV_LIB=${GRAFT_REPO_ROOT:-/root/repo}/videovector_amd/lib/libvideovec_lab.so
cd $GRAFT_REPO_ROOT
run() { # label, env...
  env "${@:2}" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms']['wgrad_gemm'])")"
}
run "full                      " A=1
run "no stream (1)             " VV_LAB_WG_ABL=1
run "no MFMA (2)               " VV_LAB_WG_ABL=2
run "no fragment reads (4)     " VV_LAB_WG_ABL=4
run "stream only (6)           " VV_LAB_WG_ABL=6
run "rows L2-hot (8)           " VV_LAB_WG_ABL=8
run "no stream no MFMA (3)     " VV_LAB_WG_ABL=3
run "nothing (7)               " VV_LAB_WG_ABL=7
run "full                      " A=1
