#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 tools/samp_time.py gpurun_out/r2_s4_samp.json > gpurun_out/r2_s4_samp.txt 2>&1
for i in 1 2 3; do VV_TRACE_HOST=0.5 VV_BENCH_DIAG=1 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 100 --warmup 5 > gpurun_out/r2_s4_$i.json 2> gpurun_out/r2_s4_$i.err; grep "vv host" gpurun_out/r2_s4_$i.err | head; done
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r2_s4_short.json 2> gpurun_out/r2_s4_short.err
python3 bench.py --no-cpu-baseline > gpurun_out/r2_s4_bench.json 2> gpurun_out/r2_s4_bench.err
for f in 1 2 3 short bench; do python3 -c "
import json,sys;d=json.load(open('gpurun_out/r2_s4_$f.json'));print('$f',d['ms_per_step'],d['value'],d.get('gpu_path_only',{}).get('ms_per_step'),d.get('step_ms_stats'))"; done
cat gpurun_out/r2_s4_samp.txt
