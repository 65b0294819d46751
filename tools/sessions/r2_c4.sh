#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
( time caffe_facade/build/caffe train --solver=examples/videovec_cfg1_solver.prototxt --log_file=/tmp/c1.log > /dev/null 2>&1 ) 2>&1 | grep real
python3 - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle as orc
orc.build()
from videovector_amd.synth import SyntheticVideos, init_weights
ds = SyntheticVideos(seed=1701, n_videos=50)
table = ds.table(128)
W, b = init_weights(3, 32, 128, std=0.02)
idx = np.random.default_rng(0).integers(0, ds.n_rows, size=(32, 7)).astype(np.int32)
for nt in (0, 8, 1):
    orc.set_threads(nt)
    t0 = time.time()
    for _ in range(5):
        orc.forward_backward(table, idx, W, b, C_=5, Nn=2, want=("dW", "db"))
    print("oracle fb small, threads", orc.get_threads(), (time.time() - t0) / 5, "s per call")
PY
