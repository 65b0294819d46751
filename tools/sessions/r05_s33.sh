#!/bin/bash
# Round-5 session 33 (lab): does the speed of the update applied in the weight-gradient GEMM's epilogue (shipped shape, wgrad_update = 1) depend on
# where W, its history and the 16-bit copy lie relative to each other?  One arena, the second / third array a 2-MiB multiple + a skew behind the first.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
export VV_LIB=$PWD/videovector_amd/lib/libvideovec_lab.so
run() {
  timeout 300 python bench.py --workload shipped --steps 150 --warmup 20 --no-cpu-baseline --no-extra-legs 2> $O/r05_s33.err | python3 -c "
import sys, json
d=json.loads([x for x in sys.stdin if x.startswith('{')][-1]); k=d['kernels_ms']; print('%-28s step %.4f  wgrad %.4f  reduce_sgd %.4f  fwd %.4f' % ('$1', d['ms_per_step'], k['wgrad_gemm'], k['reduce_sgd'], k['fwd_gemm']))"
  grep -h "vv lab" $O/r05_s33.err | head -1
}
{
export VV_WGRAD_UPDATE=1
for i in 1 2 3 4; do unset VV_LAB_PARAM_ARENA; run "separate allocations #$i"; done
for sk in "0,0" "4096,0" "8192,0" "16384,0" "65536,0" "262144,0" "1048576,0" "0,4096" "0,65536" "4096,8192" "69632,139264" "0,0"; do export VV_LAB_PARAM_ARENA=$sk; run "arena skew $sk"; done
unset VV_LAB_PARAM_ARENA
export VV_WGRAD_UPDATE=0
run "update as its own launch"
} > $O/r05_s33_param_placement.txt 2>&1
cat $O/r05_s33_param_placement.txt
