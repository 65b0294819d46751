cd $GRAFT_REPO_ROOT
summ() { python3 - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    d = json.loads([l for l in open(f).read().splitlines() if l.startswith("{")][-1])
    print(f.split("/")[-1], "ms %.5f" % d["ms_per_step"], "box %.0f" % d["box"]["gemm_tflops"], {k: round(v, 4) for k, v in d["kernels_ms"].items()})
PY
}
for i in 1 2 3; do for l in v2 v3; do
  if [ $l = v2 ]; then export VV_LIB=$PWD/videovector_amd/lib/libvideovec_v2.so; else unset VV_LIB; fi
  timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --steps 400 > gpurun_out/r06_s23_${l}_$i.json 2>> gpurun_out/r06_s23.err
done; done
summ gpurun_out/r06_s23_v*.json
unset VV_LIB
timeout 1500 python -m pytest tests/test_gpu_wgrad_lean.py tests/test_gpu_fused_update.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_cfg5.py -x -q -m gpu 2>&1 | tail -3
