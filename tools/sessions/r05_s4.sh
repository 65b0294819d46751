#!/bin/bash
# Round-5 session 4: k_score_fwd's phase stamps (lab library); the cost of the data-parallel schedules on one rank over real RCCL with the
# overlapped update's first chunk in the compute stream / on the communication stream; the exchange tests.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
VV_LIB=$PWD/videovector_amd/lib/libvideovec_lab.so VV_LAB_SCORE_TS=1 timeout 300 python tools/lab/score_ts.py > $O/r05_s4_score_ts.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_comm.py tests/test_gpu_dist.py -q -x > $O/r05_s4_pytest.log 2>&1; echo "pytest exit $?" >> $O/r05_s4_pytest.log
python3 - > $O/r05_s4_overlap_cost.txt 2>&1 <<'PY'
import os, subprocess, sys
for rep in range(2):
    for mode, env in (("none", {}), ("sync", {}), ("overlap", {"VV_COMM_FIRST_INLINE": "1"}), ("overlap", {"VV_COMM_FIRST_INLINE": "0"}), ("sharded", {}),
                      ("overlap", {"VV_COMM_FIRST_INLINE": "1", "VV_COMM_TEST_DELAY_US": "20"}), ("overlap", {"VV_COMM_FIRST_INLINE": "0", "VV_COMM_TEST_DELAY_US": "20"}),
                      ("sharded", {"VV_COMM_TEST_DELAY_US": "60"})):
        print("--", mode, env, flush=True)
        subprocess.run([sys.executable, "tools/lab/overlap_cost.py", mode, "600"], env=dict(os.environ, **env))
PY
cat $O/r05_s4_score_ts.txt | tail -20
tail -3 $O/r05_s4_pytest.log
grep -E "^--|ms/step" $O/r05_s4_overlap_cost.txt
