#!/bin/bash
# Round-5 session 38: slab pitch D_p F_p (pad 0) against D_p F_p + 4 KiB, alternating on one box: cfg 2, dense, cfg 5.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
P0=$PWD/videovector_amd/lib/libvideovec_pad0.so
run() { # label lib args...
  local label=$1 lib=$2; shift 2
  VV_LIB=$lib timeout 600 python bench.py "$@" --no-cpu-baseline --no-extra-legs 2> /dev/null | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); k=d['kernels_ms']; print('%-22s step %.4f  wgrad %.4f  reduce_sgd %.4f  fwd %.4f' % ('$label', d['ms_per_step'], k['wgrad_gemm'], k['reduce_sgd'], k['fwd_gemm']))"
}
{
for i in 1 2 3; do
  run "cfg2  pad 0" $P0 --steps 300 --warmup 30
  run "cfg2  pad 4 KiB" "" --steps 300 --warmup 30
done
for i in 1 2; do
  run "dense pad 0" $P0 --dedup off
  run "dense pad 4 KiB" "" --dedup off
  run "cfg5  pad 0" $P0 --workload cfg5 --steps 40 --warmup 5
  run "cfg5  pad 4 KiB" "" --workload cfg5 --steps 40 --warmup 5
done
} > gpurun_out/r05_s38_slab_pad_ab.txt 2>&1
cat gpurun_out/r05_s38_slab_pad_ab.txt
