#!/bin/bash
# Round-5 session 16: sampler on the EPYC host: generation cost per word, stage waiting incl. the stream thread, rates.
cd $GRAFT_REPO_ROOT
O=gpurun_out
g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -I include tools/lab/walk_bench.cc -o /tmp/walk_bench -lrt 2> /dev/null
g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -DVV_WALK_PROF -I include tools/lab/walk_bench.cc -o /tmp/walk_bench_p -lrt 2> /dev/null
{
echo "== 512"; timeout 120 /tmp/walk_bench 2000000 | tail -5; timeout 120 /tmp/walk_bench_p 1000000 | head -2
echo "== 256"; VV_SAMPLER_AVX512=0 timeout 120 /tmp/walk_bench 2000000 | tail -5; VV_SAMPLER_AVX512=0 timeout 120 /tmp/walk_bench_p 1000000 | head -2
timeout 300 python3 tools/lab/samp_stages.py
timeout 600 python3 tools/samp_rates.py 10
} > $O/r05_s16_sampler.txt 2>&1
cat $O/r05_s16_sampler.txt
