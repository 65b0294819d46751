"""GPU test of `caffe train` reading a real `source:` database: VideoShots records in an
LMDB environment -> facade LmdbReader -> feature table in HBM -> fused step; the loss trajectory is checked
against the oracle run on the same records."""
import os
import re

import numpy as np
import pytest

from tests.test_facade_lmdb import make_shots_db, pb as lmdb_pb  # noqa: F401
from tests.test_facade_proto import pb, tool  # noqa: F401  (fixtures)
from tests.test_gpu_facade import read_caffemodel, run_caffe, write_caffemodel
from tests.test_gpu_parity import rel_fro, round_operand
from videovector_amd.prototxt import solver, train_net
from videovector_amd.synth import init_weights

pytestmark = pytest.mark.gpu


def test_caffe_train_from_lmdb_matches_oracle(tool, pb, lmdb_pb, oracle, tmp_path):
    B, C, Nn, F, D = 16, 5, 3, 96, 32
    vids = make_shots_db(lmdb_pb, str(tmp_path / "train_db"), n_videos=31, F=F, seed=9)
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net(str(tmp_path / "train_db"), B, C, Nn, D, max_buffer=300, w_std=0.02))
    sol_p.write_text(solver(str(net_p), base_lr=0.01, max_iter=8, display=1, snapshot_prefix=str(tmp_path / "snap")))
    W0, b0 = init_weights(4, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
                    str(tmp_path / "train.log"))
    assert "Opening lmdb" in log
    losses = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]
    assert len(losses) == 9

    video_id = np.array([v for v, _, _ in vids], np.int32)
    n_shots = np.array([len(i) for _, i, _ in vids], np.int32)
    row_base = np.concatenate([[0], np.cumsum(n_shots[:-1])]).astype(np.int64)
    shot_ids = np.concatenate([i for _, i, _ in vids]).astype(np.int32)
    table = np.concatenate([f for _, _, f in vids]).astype(np.float32)
    smp = oracle.Sampler(video_id, n_shots, row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=300, negative_swap_percentage=50, shot_ids=shot_ids)
    Wq, bq = W0.copy(), b0.copy()
    hW, hb = np.zeros_like(W0), np.zeros_like(b0)
    for it in range(8):
        idx = smp.next()[0]
        lr = oracle.learning_rate("inv", 0.01, 1e-3, 0.75, 0, it)
        r = oracle.forward_backward(table, idx, round_operand(Wq, "f16"), bq, C_=C, Nn=Nn, want=("dW", "db"))
        assert abs(losses[it] - r["loss"]) <= 1e-3 * r["loss"], (it, losses[it], r["loss"])
        oracle.sgd_update(Wq, r["dW"], hW, lr, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bq, r["db"], hb, lr, 2.0, 0.9, 5e-4, 0.0)
    Wg, bg, _ = read_caffemodel(pb, str(tmp_path / "snap_iter_8.caffemodel"))
    assert rel_fro(Wg, Wq) <= 1e-3 and rel_fro(bg, bq) <= 2e-3
