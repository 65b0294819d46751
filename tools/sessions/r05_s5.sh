#!/bin/bash
# Round-5 session 5: bench.py's N > 1 line with its schedule / transport / sampler legs on the one-device hook, the exchange tests, the cost of
# the schedules on one rank over real RCCL (clean box), the CPU baseline by thread count and binding.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_comm.py -q -x --durations=8 > $O/r05_s5_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s5_pytest.log
ps aux | grep -c "bench.py" > $O/r05_s5_strays.txt
python3 - > $O/r05_s5_overlap_cost.txt 2>&1 <<'PY'
import os, subprocess, sys
for rep in range(2):
    for mode, env in (("none", {}), ("sync", {}), ("overlap", {"VV_COMM_FIRST_INLINE": "1"}), ("overlap", {"VV_COMM_FIRST_INLINE": "0"}), ("sharded", {}),
                      ("overlap", {"VV_COMM_FIRST_INLINE": "1", "VV_COMM_TEST_DELAY_US": "20"}), ("overlap", {"VV_COMM_FIRST_INLINE": "0", "VV_COMM_TEST_DELAY_US": "20"}),
                      ("sharded", {"VV_COMM_TEST_DELAY_US": "60"})):
        print("--", mode, env, flush=True)
        subprocess.run([sys.executable, "tools/lab/overlap_cost.py", mode, "600"], env=dict(os.environ, **env))
PY
timeout 900 python tools/lab/cpu_baseline_scaling.py > $O/r05_s5_cpu_scaling.txt 2>&1
tail -14 $O/r05_s5_pytest.log
grep -E "^--|ms/step" $O/r05_s5_overlap_cost.txt
cat $O/r05_s5_cpu_scaling.txt | cut -c1-700
