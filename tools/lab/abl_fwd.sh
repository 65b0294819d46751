cd $GRAFT_REPO_ROOT
# needs the lab build of the library (make -C videovector_amd/csrc lab): the ablated kernels are not in the product library
export VV_LIB=${GRAFT_REPO_ROOT:-/root/repo}/videovector_amd/lib/libvideovec_lab.so
for ab in 0 6 14 8 2; do
  VV_GEMM_VARIANT=5 VV_ABLATE=$ab timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/abl.log 2>&1
  echo "ablate $ab: $(python3 -c "
import json
l=[x for x in open('gpurun_out/abl.log') if x.startswith('{')]
d=json.loads(l[-1]); print('fwd', d['kernels_ms']['fwd_gemm'], 'wgrad', d['kernels_ms']['wgrad_gemm'])")"
done
