#!/bin/bash
# Round-5 session 15: the walk's apply loop on the EPYC host, ablated (lab build of walk_bench: wrong results by design).
cd $GRAFT_REPO_ROOT
O=gpurun_out
g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -DVV_WALK_LAB -I include tools/lab/walk_bench.cc -o /tmp/walk_bench_l -lrt 2> /dev/null
{
for l in 0 1 2 16 0 1 2 16; do echo "LAB=$l: $(LAB=$l /tmp/walk_bench_l 2000000 | sed -n 3p)"; done
} > $O/r05_s15_walk_lab.txt 2>&1
cat $O/r05_s15_walk_lab.txt
