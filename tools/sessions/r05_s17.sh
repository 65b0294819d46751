#!/bin/bash
# Round-5 session 17: placement of the sampler's stage threads on the EPYC host.
cd $GRAFT_REPO_ROOT
O=gpurun_out
{ lscpu | grep -E "Model name|Thread|Core|Socket|NUMA|L2|L3"; cat /sys/devices/system/cpu/cpu8/topology/thread_siblings_list; python3 -c "import os; a=sorted(os.sched_getaffinity(0)); print(len(a), a[:4], a[-4:])"
timeout 500 python3 tools/lab/samp_place.py; } > $O/r05_s17_place.txt 2>&1
cat $O/r05_s17_place.txt
