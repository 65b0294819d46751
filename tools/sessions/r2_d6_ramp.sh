#!/bin/bash
# step time against steps since the device was last idle (no settle steps), after a short and after a long idle spell
cd $GRAFT_REPO_ROOT
for idle in 0 20 0 20; do
  sleep $idle
  VV_BENCH_DIAG=1 python3 bench.py --steps 3000 --warmup 0 --settle-ms 0 --no-cpu-baseline --no-extra-legs 2>&1 >/dev/null | grep "step ms" | python3 -c "
import sys, statistics
x=[float(t) for t in sys.stdin.read().split(':')[1].split()]
edges=[0,20,40,80,120,160,200,300,400,600,800,1200,1600,2000,2600]
print('idle $idle s:', ' '.join('%d:%.4f' % (a, statistics.mean(x[a:b])) for a,b in zip(edges[:-1], edges[1:])))"
done
