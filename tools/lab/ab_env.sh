# A/B of environment switches on one box: bash tools/lab/ab_env.sh "LABEL1:ENV1=.. ENV2=.." "LABEL2:..." ...
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py tests/test_gpu_comm.py tests/test_gpu_cfg5.py tests/test_gpu_segbwd.py tests/test_gpu_shipped.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do
  for spec in "$@"; do
    label="${spec%%:*}"; envs="${spec#*:}"
    env $envs timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
    echo "$label: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'])")"
  done
done
