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
for r in 1 2 3; do
  run "prev" libvideovec_prev.so A=1
  run "new " libvideovec_new.so A=1
done
cp $L/libvideovec_new.so $L/libvideovec.so
