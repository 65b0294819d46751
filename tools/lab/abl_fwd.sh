cd $GRAFT_REPO_ROOT
for ab in 0 6 14 8 2; do
  VV_GEMM_VARIANT=5 VV_ABLATE=$ab timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/abl.log 2>&1
  echo "ablate $ab: $(python3 -c "
import json
l=[x for x in open('gpurun_out/abl.log') if x.startswith('{')]
d=json.loads(l[-1]); print('fwd', d['kernels_ms']['fwd_gemm'], 'wgrad', d['kernels_ms']['wgrad_gemm'])")"
done
