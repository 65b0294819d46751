#!/bin/bash
# Round-5 session 14: host sampler on the GPU box's host (EPYC 9575F): walk / slot / frame stages alone for both vector widths,
# pipeline rates with spread, per-stage waiting.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -I include tools/lab/walk_bench.cc -o /tmp/walk_bench -lrt 2> /dev/null
g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -DVV_WALK_PROF -I include tools/lab/walk_bench.cc -o /tmp/walk_bench_p -lrt 2> /dev/null
{
grep -m1 "model name" /proc/cpuinfo
echo "== stages alone, 512-bit forms"; timeout 120 /tmp/walk_bench 2000000
echo "== walk with the stream from a helper thread, 512"; HELPER=1 timeout 120 /tmp/walk_bench 2000000 | head -3
echo "== stages alone, 256-bit forms (VV_SAMPLER_AVX512=0)"; VV_SAMPLER_AVX512=0 timeout 120 /tmp/walk_bench 2000000
echo "== sections (rdtsc, ~25 ticks each) 512"; timeout 120 /tmp/walk_bench_p 1000000 | head -2
echo "== sections 256"; VV_SAMPLER_AVX512=0 timeout 120 /tmp/walk_bench_p 1000000 | head -2
} > $O/r05_s14_walk.txt 2>&1
timeout 600 python3 tools/samp_rates.py 10 $O/r05_s14_sampler_rates.json > $O/r05_s14_sampler_rates.txt 2>&1
timeout 300 python3 tools/lab/samp_stages.py >> $O/r05_s14_sampler_rates.txt 2>&1
echo "== 256-bit forms" >> $O/r05_s14_sampler_rates.txt
VV_SAMPLER_AVX512=0 timeout 600 python3 tools/samp_rates.py 5 >> $O/r05_s14_sampler_rates.txt 2>&1
cat $O/r05_s14_walk.txt; cat $O/r05_s14_sampler_rates.txt
