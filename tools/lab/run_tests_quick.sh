cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fused_update.py -m gpu -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py tests/test_gpu_comm.py tests/test_gpu_cfg5.py tests/test_gpu_shipped.py tests/test_gpu_facade.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2 3; do for v in "VV_FUSE_UPDATE=0" "VV_FUSE_KEEP_GRADS=1" "A=1"; do
env $v timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), d['kernels_ms'], d['final_loss'])"
done; done
