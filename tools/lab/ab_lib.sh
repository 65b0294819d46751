# A/B of two builds of the library on one box: videovector_amd/lib/libvideovec_prev.so against libvideovec.so
cd $GRAFT_REPO_ROOT
L=videovector_amd/lib
cp $L/libvideovec.so $L/libvideovec_new.so
run() { # label, lib, env...
  cp $L/$2 $L/libvideovec.so
  env "${@:3}" timeout 300 python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-extra-legs > gpurun_out/ab.log 2>&1
  echo "$1: $(python3 -c "
import json
l=[x for x in open('gpurun_out/ab.log') if x.startswith('{')]
d=json.loads(l[-1]); print(round(d['ms_per_step'],4), d['kernels_ms'])")"
}
cp $L/libvideovec_new.so $L/libvideovec.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dedup.py tests/test_gpu_comm.py -m gpu -x -q 2>&1 | tail -2
for r in 1 2 3; do
  run "prev        " libvideovec_prev.so A=1
  run "new, lead 0 " libvideovec_new.so VV_FWD_LEAD=0
  run "new, lead 1 " libvideovec_new.so VV_FWD_LEAD=1
done
run "prev dense  " libvideovec_prev.so VV_DEDUP=0
run "new dense l0" libvideovec_new.so VV_DEDUP=0 VV_FWD_LEAD=0
run "new dense l1" libvideovec_new.so VV_DEDUP=0 VV_FWD_LEAD=1
cp $L/libvideovec_new.so $L/libvideovec.so
