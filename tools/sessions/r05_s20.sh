#!/bin/bash
cd $GRAFT_REPO_ROOT
g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -I include tools/lab/walk_bench.cc -o /tmp/walk_bench -lrt 2> /dev/null
{
echo "serial:            $(PIN=9 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "serial + log:      $(PIN=9 LOGEV=1 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "helper on 10:      $(HELPER=1 PIN=9,10 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "helper on 10 + log:$(HELPER=1 PIN=9,10 LOGEV=1 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "helper on sibling: $(HELPER=1 PIN=9,137 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "helper other CCD:  $(HELPER=1 PIN=9,17 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "helper on 10, pf 2048: $(VV_SAMPLER_PREFETCH=2048 HELPER=1 PIN=9,10 /tmp/walk_bench 2000000 | sed -n 3p)"
echo "helper on 10, block 262144: $(VV_SAMPLER_BLOCK=262144 HELPER=1 PIN=9,10 /tmp/walk_bench 2000000 | sed -n 3p)"
} > gpurun_out/r05_s20_walk_helper.txt 2>&1
cat gpurun_out/r05_s20_walk_helper.txt
