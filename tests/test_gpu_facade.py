"""GPU tests of the C++ facade: `caffe train` (the reference's train entry, tools/caffe.cpp:80-123)
driven by solver / net prototxts, checked against the oracle's trajectory; caffemodel / solverstate
files exchanged with the REAL protobuf runtime; snapshot -> restore."""
import os
import re
import subprocess

import numpy as np
import pytest

from tests.test_facade_proto import pb, tool  # noqa: F401  (fixtures)
from tests.test_gpu_parity import rel_fro, round_operand
from videovector_amd.prototxt import solver, train_net
from videovector_amd.synth import SyntheticVideos, init_weights

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAFFE = os.path.join(ROOT, "caffe_facade", "build", "caffe")


def write_caffemodel(pb, path, W, b):
    net = pb["NetParameter"](name="init")
    l = net.layers.add(name="fc7", type=14)
    l.blobs.add(num=1, channels=1, height=W.shape[0], width=W.shape[1]).data.extend(W.reshape(-1).tolist())
    l.blobs.add(num=1, channels=1, height=1, width=len(b)).data.extend(b.tolist())
    open(path, "wb").write(net.SerializeToString())


def read_caffemodel(pb, path):
    net = pb["NetParameter"]()
    net.ParseFromString(open(path, "rb").read())
    fc = [l for l in net.layers if l.name == "fc7"][0]
    W = np.array(fc.blobs[0].data, np.float32).reshape(fc.blobs[0].height, fc.blobs[0].width)
    return W, np.array(fc.blobs[1].data, np.float32), net


def run_caffe(args, log, env=None):
    r = subprocess.run([CAFFE] + args + ["--log_file=%s" % log], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return open(log).read()


@pytest.mark.parametrize("dedup", ["0", "1"])
def test_caffe_train_matches_oracle_trajectory(tool, pb, oracle, tmp_path, dedup):
    # dedup "0": dense path, tight bounds on the 12-iteration trajectory.  "1" (the default of the library):
    # the gradient sums are reassociated, and this small case is chaotic (tests/test_gpu_parity.py:
    # test_sgd_steps_match_oracle), so the weights after 12 free-running iterations get a loose bound while
    # every per-iteration loss keeps the 1e-3 bound.
    # (round 6: "0" also keeps the split-K partial products of dW as fp32 -- VV_SLAB16=0, the tight bounds are bounds on fixed arithmetic,
    # tests/conftest.py: fp32_slabs --; "1" is the library as it ships: de-duplication, f16 rows of ip2, f16 partial products)
    import videovector_amd as vv
    env = {"VV_DEDUP": dedup, "VV_SLAB16": dedup}
    loose = dedup == "1"
    B, C, Nn, F, D, V = 32, 5, 2, 128, 32, 50
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=500,
                               w_std=0.02))
    kw = dict(base_lr=0.01, max_iter=12, display=1, snapshot=6, snapshot_prefix=str(tmp_path / "snap"))
    sol_p.write_text(solver(str(net_p), **kw))
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
                    str(tmp_path / "train.log"), env)
    losses = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]
    lrs = [float(x) for x in re.findall(r"Iteration \d+, lr = ([0-9.eE+-]+)", log)]
    assert len(losses) == 13 and len(lrs) == 12          # 12 iterations + the final display pass
    # log lines in the exact form caffe_utils/plot_training_stats.py greps (solver.cpp:211-214)
    assert re.search(r"Train net output #0: loss_output = iter = 0 value = [0-9.e+-]+ \(\* 1 = [0-9.e+-]+ loss\)", log)
    assert re.search(r"Train net output #1: train_violations = iter = 0 value = \d+", log)

    ds = SyntheticVideos(seed=1701, n_videos=V)
    table = ds.table(F)
    smp = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=500, negative_swap_percentage=50)
    Wq, bq = W0.copy(), b0.copy()
    hW, hb = np.zeros_like(W0), np.zeros_like(b0)
    for it in range(12):
        idx = smp.next()[0]
        lr = oracle.learning_rate("inv", 0.01, 1e-3, 0.75, 0, it)
        assert abs(lrs[it] - lr) <= 1e-6 * lr
        r = oracle.forward_backward(table, idx, round_operand(Wq, "f16"), bq, C_=C, Nn=Nn, want=("dW", "db"))
        assert abs(losses[it] - r["loss"]) <= 1e-3 * r["loss"], (it, losses[it], r["loss"])
        oracle.sgd_update(Wq, r["dW"], hW, lr, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bq, r["db"], hb, lr, 2.0, 0.9, 5e-4, 0.0)
        if it == 5:
            W6, h6 = Wq.copy(), hW.copy()
    Wg, bg, net = read_caffemodel(pb, str(tmp_path / "snap_iter_12.caffemodel"))
    assert len(net.layers) == 20 and rel_fro(Wg, Wq) <= (5e-2 if loose else 1e-3) and rel_fro(bg, bq) <= (1e-1 if loose else 2e-3)
    # mid-run snapshot: weights and momentum history after 6 iterations
    Wm, _, _ = read_caffemodel(pb, str(tmp_path / "snap_iter_6.caffemodel"))
    st = pb["SolverState"]()
    st.ParseFromString(open(tmp_path / "snap_iter_6.solverstate", "rb").read())
    assert st.iter == 6 and st.learned_net.endswith("snap_iter_6.caffemodel") and len(st.history) == 2
    hist = np.array(st.history[0].data, np.float32).reshape(D, F)
    assert rel_fro(Wm, W6) <= (3e-2 if loose else 1e-3) and rel_fro(hist, h6) <= (1e-1 if loose else 4e-3)

    # resume from the snapshot with nothing left to do: restores iter, weights and history exactly
    sol2 = tmp_path / "solver2.prototxt"
    sol2.write_text(solver(str(net_p), **dict(kw, max_iter=6, snapshot=0, snapshot_prefix=str(tmp_path / "re"))))
    log2 = run_caffe(["train", "--solver=%s" % sol2, "--snapshot=%s" % (tmp_path / "snap_iter_6.solverstate")],
                     str(tmp_path / "resume.log"), env)
    assert "Restoring previous solver status" in log2
    Wr, br, _ = read_caffemodel(pb, str(tmp_path / "re_iter_6.caffemodel"))
    st2 = pb["SolverState"]()
    st2.ParseFromString(open(tmp_path / "re_iter_6.solverstate", "rb").read())
    assert np.array_equal(Wr, Wm) and st2.iter == 6
    assert np.array_equal(np.array(st2.history[0].data, np.float32), np.array(st.history[0].data, np.float32))


@pytest.mark.parametrize("stype,mom", [("NESTEROV", 0.9), ("ADAGRAD", 0.0)])
def test_caffe_train_nesterov_and_adagrad_solvers(tool, pb, oracle, tmp_path, stype, mom):
    # GetSolver (solver.hpp:128-143) -> NesterovSolver / AdaGradSolver; two iterations against the oracle
    # (short enough that the chaotic growth of rounding differences stays below the bound)
    B, C, Nn, F, D, V = 32, 5, 2, 128, 32, 50
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=500,
                               w_std=0.02))
    sol_p.write_text(solver(str(net_p), base_lr=0.01, momentum=mom, max_iter=2, display=1, lr_policy="fixed",
                            snapshot_prefix=str(tmp_path / "snap"), solver_type=stype, delta=1e-6))
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
              str(tmp_path / "train.log"), {"VV_DEDUP": "0"})
    ds = SyntheticVideos(seed=1701, n_videos=V)
    table = ds.table(F)
    smp = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=500, negative_swap_percentage=50)
    Wq, bq = W0.copy(), b0.copy()
    hW, hb = np.zeros_like(W0), np.zeros_like(b0)
    for it in range(2):
        idx = smp.next()[0]
        r = oracle.forward_backward(table, idx, round_operand(Wq, "f16"), bq, C_=C, Nn=Nn, want=("dW", "db"))
        oracle.sgd_update(Wq, r["dW"], hW, 0.01, 1.0, mom, 5e-4, 1.0, solver=stype, delta=1e-6)
        oracle.sgd_update(bq, r["db"], hb, 0.01, 2.0, mom, 5e-4, 0.0, solver=stype, delta=1e-6)
    Wg, bg, _ = read_caffemodel(pb, str(tmp_path / "snap_iter_2.caffemodel"))
    st = pb["SolverState"]()
    st.ParseFromString(open(tmp_path / "snap_iter_2.solverstate", "rb").read())
    hist = np.array(st.history[0].data, np.float32).reshape(D, F)
    assert rel_fro(Wg, Wq) <= 1e-3 and rel_fro(bg, bq) <= 2e-3 and rel_fro(hist, hW) <= 5e-3
    if stype == "ADAGRAD":       # AdaGradSolver's constructor check (solver.hpp:121-122)
        bad = tmp_path / "bad_solver.prototxt"
        bad.write_text(solver(str(net_p), momentum=0.9, max_iter=1, solver_type="ADAGRAD"))
        r = subprocess.run([CAFFE, "train", "--solver=%s" % bad], capture_output=True, text=True, timeout=120)
        assert r.returncode != 0 and "Momentum cannot be used with AdaGrad" in (r.stderr + r.stdout)


@pytest.mark.parametrize("direct", [False, True])
def test_caffe_train_weighted_loss_and_ip_regularization(tool, pb, oracle, tmp_path, direct):
    # third bottom of MAX_MARGIN_LOSS = the data layer's video ids replicated by a SUM layer; weights from
    # id_to_weight_file (ids missing from the file weigh 0) or the ids themselves (use_direct_weight);
    # InnerProduct regularization: 0.5 scales dW by 1.25
    B, C, Nn, F, D, V = 32, 5, 2, 128, 32, 50
    wfile = tmp_path / "id2w.txt"
    wmap = {v: 0.25 * (v % 5) + 0.5 for v in range(0, V, 2)}
    wfile.write_text("".join("%d,%g\n" % kv for kv in wmap.items()))
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=500, w_std=0.02,
                               id_to_weight_file=None if direct else str(wfile), use_direct_weight=direct,
                               ip_regularization=0.5))
    sol_p.write_text(solver(str(net_p), base_lr=0.002, max_iter=3, display=1, lr_policy="fixed",
                            snapshot_prefix=str(tmp_path / "snap")))
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
                    str(tmp_path / "train.log"), {"VV_DEDUP": "0"})
    losses = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]
    ds = SyntheticVideos(seed=1701, n_videos=V)
    table = ds.table(F)
    smp = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=500, negative_swap_percentage=50)
    Wq, bq = W0.copy(), b0.copy()
    hW, hb = np.zeros_like(W0), np.zeros_like(b0)
    for it in range(3):
        idx, _, label = smp.next()
        w = label.astype(np.float32) if direct else np.array([wmap.get(int(v), 0.0) for v in label], np.float32)
        r = oracle.forward_backward(table, idx, round_operand(Wq, "f16"), bq, C_=C, Nn=Nn, item_weight=w,
                                    ip_regularization=0.5, want=("dW", "db"))
        assert abs(losses[it] - r["loss"]) <= 1e-3 * r["loss"], (it, losses[it], r["loss"])
        oracle.sgd_update(Wq, r["dW"], hW, 0.002, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bq, r["db"], hb, 0.002, 2.0, 0.9, 5e-4, 0.0)
    Wg, bg, _ = read_caffemodel(pb, str(tmp_path / "snap_iter_3.caffemodel"))
    assert rel_fro(Wg, Wq) <= 2e-3 and rel_fro(bg, bq) <= 4e-3


@pytest.mark.parametrize("ctype,C", [("PAST", 4), ("PAST_CONTINUOUS", 3), ("PAST_CONTINUOUS_FIXED", 5)])
def test_caffe_train_past_context_types(tool, pb, oracle, tmp_path, ctype, C):
    # context_type PAST* (video_sampled_shots_data_layer.cpp:510-757) through the data layer, with same-video
    # negatives (quirk Q1 slots) and an even context size
    B, Nn, F, D, V = 32, 3, 128, 32, 50
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=500, w_std=0.02,
                               context_type=ctype, max_same=2))
    sol_p.write_text(solver(str(net_p), base_lr=0.002, max_iter=3, display=1, lr_policy="fixed",
                            snapshot_prefix=str(tmp_path / "snap")))
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
                    str(tmp_path / "train.log"), {"VV_DEDUP": "0"})
    losses = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]
    ds = SyntheticVideos(seed=1701, n_videos=V)
    table = ds.table(F)
    smp = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=500, negative_swap_percentage=50, max_same_video_negs=2, context_type=ctype)
    Wq, bq = W0.copy(), b0.copy()
    hW, hb = np.zeros_like(W0), np.zeros_like(b0)
    saw_q1 = False
    for it in range(3):
        idx, last, _ = smp.next()
        saw_q1 |= bool((idx != last).any())
        r = oracle.forward_backward(table, idx, round_operand(Wq, "f16"), bq, C_=C, Nn=Nn, last_src=last, want=("dW", "db"))
        assert abs(losses[it] - r["loss"]) <= 1e-3 * r["loss"], (it, losses[it], r["loss"])
        oracle.sgd_update(Wq, r["dW"], hW, 0.002, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bq, r["db"], hb, 0.002, 2.0, 0.9, 5e-4, 0.0)
    assert saw_q1
    Wg, bg, _ = read_caffemodel(pb, str(tmp_path / "snap_iter_3.caffemodel"))
    assert rel_fro(Wg, Wq) <= 2e-3 and rel_fro(bg, bq) <= 4e-3


def test_every_named_blob_of_the_train_graph_is_materialisable(tool, pb, oracle, tmp_path):
    # Net::blob_by_name (net.cpp:846-857) on a graph that runs as a fused plan: every intermediate blob is rebuilt on
    # demand; values against the oracle's layer-by-layer forward of the same batch
    B, C, Nn, F, D, V = 16, 5, 3, 128, 32, 50
    net_p = tmp_path / "net.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=500, w_std=0.05))
    W0, b0 = init_weights(3, D, F, std=0.05)
    b0 = (np.random.default_rng(1).standard_normal(D) * 0.05).astype(np.float32)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    names = ["data", "target_datum", "context_datum_2", "negative_datum_3", "concat_input_datums", "original_feature",
             "ip1_nonorm", "ip2", "target_emb_nonorm", "context_window_emb_3_nonorm", "negative_emb_2_nonorm",
             "context_feature_nonorm", "context_feature", "pos_neg_nonorm", "pos_neg", "target_emb", "negative_emb_1",
             "target_prod", "neg_prod_2", "target_score", "neg_score_3", "negative_scores", "loss_output", "train_violations"]
    out = tmp_path / "blobs"
    r = subprocess.run([os.path.join(ROOT, "caffe_facade", "build", "dump_blobs"), str(net_p), str(tmp_path / "init.caffemodel"),
                        str(out), ",".join(names)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]

    def load(n):
        raw = open(out / (n + ".bin"), "rb").read()
        shape = np.frombuffer(raw[:16], np.int32)
        return shape, np.frombuffer(raw[16:], np.float32)
    ds = SyntheticVideos(seed=1701, n_videos=V)
    table = ds.table(F)
    smp = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=500, negative_swap_percentage=50)
    idx = smp.next()[0]
    ref = oracle.forward_backward(table, idx, W0, b0, C_=C, Nn=Nn, want=("Y", "H", "ctx", "posneg", "s_true", "s_bogus"))
    X = table[idx]                                            # [B][CN][F]
    Xcm = X.transpose(1, 0, 2).reshape(-1, F)                 # channel-major rows
    H = ref["H"].reshape(C + Nn, B, D)
    PN = ref["posneg"].reshape(1 + Nn, B, D)
    ctx_mean = H[1:C].mean(0)
    pn_nonorm = np.concatenate([H[0:1], H[C:]], 0)
    want = {
        "data": ((B, C + Nn, F, 1), X), "target_datum": ((B, 1, F, 1), X[:, 0]), "context_datum_2": ((B, 1, F, 1), X[:, 2]),
        "negative_datum_3": ((B, 1, F, 1), X[:, C + 2]), "concat_input_datums": (((C + Nn) * B, 1, F, 1), Xcm),
        "original_feature": (((C + Nn) * B, F, 1, 1), Xcm), "ip1_nonorm": (((C + Nn) * B, D, 1, 1), ref["Y"]),
        "ip2": (((C + Nn) * B, D, 1, 1), ref["H"]), "target_emb_nonorm": ((B, D, 1, 1), H[0]),
        "context_window_emb_3_nonorm": ((B, D, 1, 1), H[3]), "negative_emb_2_nonorm": ((B, D, 1, 1), H[C + 1]),
        "context_feature_nonorm": ((B, D, 1, 1), ctx_mean), "context_feature": ((B, D, 1, 1), ref["ctx"]),
        "pos_neg_nonorm": (((1 + Nn) * B, D, 1, 1), pn_nonorm), "pos_neg": (((1 + Nn) * B, D, 1, 1), ref["posneg"]),
        "target_emb": ((B, D, 1, 1), PN[0]), "negative_emb_1": ((B, D, 1, 1), PN[1]),
        "target_prod": ((B, D, 1, 1), ref["ctx"] * PN[0]), "neg_prod_2": ((B, D, 1, 1), ref["ctx"] * PN[2]),
        "target_score": ((B, Nn, 1, 1), ref["s_true"]), "neg_score_3": ((B, 1, 1, 1), ref["s_bogus"][:, 2]),
        "negative_scores": ((B, Nn, 1, 1), ref["s_bogus"]),
        "loss_output": ((1, 1, 1, 1), np.array([ref["loss"]])), "train_violations": ((1, 1, 1, 1), np.array([ref["violations"]])),
    }
    for n, (shape, val) in want.items():
        got_shape, got = load(n)
        assert tuple(got_shape) == shape, (n, tuple(got_shape), shape)
        val = np.asarray(val, np.float32).reshape(-1)
        err = np.abs(got - val).max() / max(np.abs(val).max(), 1e-12)
        assert err <= 2e-3, (n, err)


def test_caffe_test_time_and_device_query_commands(tool, pb, oracle, tmp_path):
    # the other three commands of tools/caffe.cpp (:68-76, :127-189, :193-265)
    from videovector_amd.synth import synthetic_windows
    B, C, Nn, F, D, V, NW = 16, 5, 3, 128, 64, 60, 90
    ds = SyntheticVideos(seed=9, n_videos=V)
    cls = {int(v): int(v % 5) + 1 for v in range(V)}
    (tmp_path / "id2class.txt").write_text("".join("%d,%d\n" % kv for kv in cls.items()))
    src = "synthetic://videos=%d;seed=9;features=%d" % (V, F)
    wsrc = "synthetic-windows://videos=%d;seed=9;features=%d;windows=%d;context=4;wseed=5" % (V, F, NW)
    net_p = tmp_path / "net.prototxt"
    net_p.write_text(train_net(src, B, C, Nn, D, max_buffer=300, w_std=0.02, test_source=wsrc, test_batch=NW,
                               test_frames=4, id_to_class_file=str(tmp_path / "id2class.txt")))
    W0, b0 = init_weights(4, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    # caffe test: two iterations of the TEST-phase net, per-batch lines and the mean of every output
    log = run_caffe(["test", "--model=%s" % net_p, "--weights=%s" % (tmp_path / "init.caffemodel"), "--iterations=2"],
                    str(tmp_path / "test.log"))
    assert "Running for 2 iterations." in log and len(re.findall(r"Batch \d+, test_map = ", log)) == 2
    mean_map = float(re.findall(r"\] test_map = ([0-9.eE+-]+)", log)[-1])
    rows, vids = synthetic_windows(ds, NW, 4, 5)
    emb = oracle.embed(ds.table(F)[rows].mean(1).astype(np.float32), None, W0, b0, relu=True, l2norm=True)
    assert abs(mean_map - oracle.retrieval_stats(emb, vids, cls)[0]) <= 5e-3
    # caffe time: the fused plan's kernels and the whole iteration
    log = run_caffe(["time", "--model=%s" % net_p, "--iterations=20"], str(tmp_path / "time.log"))
    assert "*** Benchmark begins ***" in log and "*** Benchmark ends ***" in log
    for k in ("fwd_gemm", "score_loss", "wgrad_gemm"):
        assert re.search(r"%s\tkernel: [0-9.eE+-]+ milliseconds" % k, log), k
    # the reduction and the update: one launch (the default) or two (VV_FUSE_UPDATE=0)
    assert re.search(r"reduce_sgd\tkernel: [0-9.eE+-]+ milliseconds", log) or \
        (re.search(r"reduce\tkernel: [0-9.eE+-]+ milliseconds", log) and re.search(r"sgd\tkernel: [0-9.eE+-]+ milliseconds", log))
    assert re.search(r"Forward-backward-update iteration: [0-9.eE+-]+ milliseconds", log)
    # caffe device_query
    log = run_caffe(["device_query", "--gpu=0"], str(tmp_path / "dq.log"))
    assert "Querying device ID = 0" in log and "gfx950" in log and re.search(r"Compute units:\s+256", log)


def mt19937_first(seed):
    """first 32-bit output of the standard mt19937 seeded with `seed` (what boost::mt19937 / std::mt19937 give)"""
    mt = [0] * 624
    mt[0] = seed & 0xFFFFFFFF
    for i in range(1, 624):
        mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
    for k in range(624):
        y = (mt[k] & 0x80000000) | (mt[(k + 1) % 624] & 0x7FFFFFFF)
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
    y = mt[0]
    y ^= y >> 11; y ^= (y << 7) & 0x9D2C5680; y ^= (y << 15) & 0xEFC60000; y ^= y >> 18
    return y & 0xFFFFFFFF


def test_caffe_train_rand_skip(tool, pb, oracle, tmp_path):
    # rand_skip: the data layer skips caffe_rng_rand() % rand_skip records, the first draw of the seeded mt19937
    assert mt19937_first(5489) == 3499211612                      # the generator's published first output
    B, C, Nn, F, D, V = 16, 5, 2, 128, 32, 50
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=300, w_std=0.02,
                               rand_skip=37))
    sol_p.write_text(solver(str(net_p), base_lr=0.002, max_iter=2, display=1, lr_policy="fixed", random_seed=99,
                            snapshot_prefix=str(tmp_path / "snap")))
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")], str(tmp_path / "t.log"))
    skip = mt19937_first(99) % 37
    assert "Skipping first %d data points." % skip in log
    losses = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]
    ds = SyntheticVideos(seed=1701, n_videos=V)
    smp = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                         max_buffer_size=300, negative_swap_percentage=50, initial_cursor=skip)
    r = oracle.forward_backward(ds.table(F), smp.next()[0], round_operand(W0, "f16"), b0, C_=C, Nn=Nn)
    assert abs(losses[0] - r["loss"]) <= 1e-3 * r["loss"]


def test_caffe_train_shipped_configuration(tool, tmp_path):
    # The shipped project settings (mednet_embedding_train.prototxt:13-23,200,226 and its solver):
    # batch 128, window 5, 10 negatives of which up to 6 from the same video (quirk Q1), 4096 -> 4096,
    # dropout 0.9 -- on a synthetic source.  Checks that the graph is accepted and iterates (finite loss
    # in the hinge's range; with dropout 0.9 and lr 1e-3 thirty iterations are far too few to see a trend).
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net("synthetic://videos=400;seed=5;features=4096", 128, 5, 10, 4096, max_same=6, dropout=0.9))
    sol_p.write_text(solver(str(net_p), base_lr=0.001, max_iter=30, display=5, snapshot_prefix=str(tmp_path / "m"),
                            random_seed=7))
    log = run_caffe(["train", "--solver=%s" % sol_p], str(tmp_path / "t.log"))
    losses = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]
    assert len(losses) == 7 and all(np.isfinite(losses)) and all(0 < l < 16 for l in losses)
    assert "Fused videovec plan: B=128 C=5 Nn=10 F=4096 D=4096" in log


def _losses(log):
    return [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log)]


def test_layer_by_layer_executor_matches_the_fused_plan(tool, pb, tmp_path):
    """Layer<Dtype>::Forward_gpu / Backward_gpu + Net::ForwardFromTo / BackwardFromTo (layer.hpp:308-337, net.cpp:501-578):
    (a) the shipped graph forced through the layer-by-layer executor (VV_FACADE_SEQUENTIAL=1) trains like the fused plan;
    (b) a graph the fused-plan matcher does NOT accept -- two context embeddings wired out of order, which round 1
        refused with a fatal error -- now trains layer by layer, and (the context average being symmetric) lands on the
        same numbers; (c) with dropout and the weighted loss the two executors still agree on the first iteration;
    (d) a layer type outside the path stays fatal."""
    B, C, Nn, F, D, V, IT = 16, 5, 3, 128, 64, 60, 4
    W0, b0 = init_weights(4, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    src = "synthetic://videos=%d;seed=1701;features=%d" % (V, F)
    txt = train_net(src, B, C, Nn, D, max_buffer=300, w_std=0.02)

    def run(tag, text, env=None, iters=IT):
        net_p, sol_p = tmp_path / ("net_%s.prototxt" % tag), tmp_path / ("sol_%s.prototxt" % tag)
        net_p.write_text(text)
        sol_p.write_text(solver(str(net_p), base_lr=0.01, max_iter=iters, display=1, snapshot_prefix=str(tmp_path / ("s_" + tag))))
        log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
                        str(tmp_path / (tag + ".log")), dict(env or {}, VV_DEDUP="0"))
        Wn, bn, _ = read_caffemodel(pb, str(tmp_path / ("s_%s_iter_%d.caffemodel" % (tag, iters))))
        return log, Wn, bn

    log_f, Wf, bf = run("fused", txt)
    assert "Fused videovec plan" in log_f
    log_s, Ws, bs = run("seq", txt, {"VV_FACADE_SEQUENTIAL": "1"})
    assert "Layer-by-layer plan" in log_s and "Fused videovec plan" not in log_s
    lf, ls = _losses(log_f), _losses(log_s)
    assert len(lf) == len(ls) == IT + 1
    print("SEQUENTIAL losses fused %s / layer-by-layer %s ; dW step rel %.3e" % (lf, ls, rel_fro(Ws - W0, Wf - W0)))
    assert max(abs(a - b) / a for a, b in zip(lf, ls)) <= 1e-3
    assert rel_fro(Ws - W0, Wf - W0) <= 2e-2 and rel_fro(bs - b0, bf - b0) <= 2e-2      # free-running small case, f16 gradient rounding differs
    viol_f = re.findall(r"train_violations = iter = 0 value = (\d+)", log_f)
    viol_s = re.findall(r"train_violations = iter = 0 value = (\d+)", log_s)
    assert viol_f == viol_s and len(viol_f) == 1

    a, b = '  bottom: "context_window_emb_1_nonorm"\n', '  bottom: "context_window_emb_2_nonorm"\n'
    assert a + b in txt
    log_r, Wr, _ = run("reordered", txt.replace(a + b, b + a, 1))                    # context embeddings out of order
    assert "Not the fused videovec pattern" in log_r and "Running this net layer by layer" in log_r
    lr_ = _losses(log_r)
    assert max(abs(x - y) / x for x, y in zip(ls, lr_)) <= 1e-5 and rel_fro(Wr - W0, Ws - W0) <= 1e-4

    (tmp_path / "id2w.txt").write_text("".join("%d,%g\n" % (v, 0.5 + (v % 4) * 0.5) for v in range(V)))
    txt_w = train_net(src, B, C, Nn, D, max_buffer=300, w_std=0.02, dropout=0.0, id_to_weight_file=str(tmp_path / "id2w.txt"))
    l_fw = _losses(run("fw", txt_w, iters=1)[0])
    l_sw = _losses(run("sw", txt_w, {"VV_FACADE_SEQUENTIAL": "1"}, iters=1)[0])
    assert abs(l_fw[0] - l_sw[0]) <= 1e-3 * l_fw[0]

    bad = txt.replace("type: RELU", "type: SIGMOID", 1)
    net_p, sol_p = tmp_path / "net_bad.prototxt", tmp_path / "sol_bad.prototxt"
    net_p.write_text(bad)
    sol_p.write_text(solver(str(net_p), max_iter=1, snapshot_prefix=str(tmp_path / "x")))
    r = subprocess.run([CAFFE, "train", "--solver=%s" % sol_p], capture_output=True, text=True)
    assert r.returncode != 0 and "outside the videovec training path" in r.stderr


@pytest.mark.parametrize("executor", ["fused", "layer-by-layer"])
def test_caffe_train_with_test_net_and_extract_features(tool, pb, oracle, tmp_path, executor):
    # the whole shipped net: TRAIN branch + TEST branch (retrieval statistics every test_interval
    # iterations through Solver::Test, solver.cpp:251-317), then extract_features on the snapshot
    from videovector_amd.prototxt import extraction_net
    from videovector_amd.synth import synthetic_windows
    B, C, Nn, F, D, V, NW = 16, 5, 3, 128, 64, 60, 90
    ds = SyntheticVideos(seed=9, n_videos=V)
    cls = {int(v): int(v % 5) + 1 for v in range(V)}
    (tmp_path / "id2class.txt").write_text("".join("%d,%d\n" % kv for kv in cls.items()))
    src = "synthetic://videos=%d;seed=9;features=%d" % (V, F)
    wsrc = "synthetic-windows://videos=%d;seed=9;features=%d;windows=%d;context=4;wseed=5" % (V, F, NW)
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net(src, B, C, Nn, D, max_buffer=300, w_std=0.02, test_source=wsrc, test_batch=NW,
                               test_frames=4, id_to_class_file=str(tmp_path / "id2class.txt")))
    sol_p.write_text(solver(str(net_p), base_lr=0.01, max_iter=4, display=2, snapshot_prefix=str(tmp_path / "s"),
                            test_iter=1, test_interval=2))
    W0, b0 = init_weights(4, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    env = {"VV_FACADE_SEQUENTIAL": "1"} if executor == "layer-by-layer" else {}
    log = run_caffe(["train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel")],
                    str(tmp_path / "t.log"), env)
    if executor == "fused":
        assert "Fused videovec TEST plan: B=%d frames=4 F=%d D=%d + retrieval stats" % (NW, F, D) in log
    else:
        assert log.count("Layer-by-layer plan") == 2          # the TRAIN net and the TEST net
    tests = re.findall(r"Iteration (\d+), Testing net \(#0\)", log)
    assert tests == ["0", "2", "4"]
    # net outputs are listed in name order (net.cpp:200-208 iterates a std::set)
    maps = [float(x) for x in re.findall(r"Test net output #2: test_map = ([0-9.eE+-]+)", log)]
    h1 = [float(x) for x in re.findall(r"Test net output #0: test_hit_at_1 = ([0-9.eE+-]+)", log)]
    assert len(maps) == 3 and len(h1) == 3
    # iteration-0 statistics are those of the initial weights: check against the oracle
    rows, vids = synthetic_windows(ds, NW, 4, 5)
    table = ds.table(F)
    emb = oracle.embed(table[rows].mean(1).astype(np.float32), None, W0, b0, relu=True, l2norm=True)
    ref = oracle.retrieval_stats(emb, vids, cls)
    assert abs(maps[0] - ref[0]) <= 5e-3 and abs(h1[0] - ref[1]) <= 2e-2, (maps[0], h1[0], ref)

    # extract_features: ip2 of single frames with the trained snapshot
    ex_p = tmp_path / "extract.prototxt"
    xsrc = "synthetic-windows://videos=%d;seed=9;features=%d;windows=25;context=1;wseed=11" % (V, F)
    ex_p.write_text(extraction_net(xsrc, 10, D))
    r = subprocess.run([os.path.join(ROOT, "caffe_facade", "build", "extract_features"),
                        str(tmp_path / "s_iter_4.caffemodel"), "none", str(ex_p), "ip2", str(tmp_path / "feat"), "2", "GPU", "0"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = (tmp_path / "feat" / "text_output.txt").read_text().strip().split("\n")
    assert lines[0] == "#features" and len(lines) == 21
    got = np.array([[float(x) for x in l.rstrip(",").split(",")] for l in lines[1:]], np.float32)
    Wt, bt, _ = read_caffemodel(pb, str(tmp_path / "s_iter_4.caffemodel"))
    xrows, _ = synthetic_windows(ds, 25, 1, 11)
    ref = oracle.embed(table, xrows[:20, 0], Wt, bt, relu=True, l2norm=False)
    assert got.shape == (20, D)
    assert (np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-20)).max() <= 1e-3


@pytest.mark.parametrize("overlap", ["0", "1", "sharded", "peer-sharded"])
def test_caffe_train_data_parallel_two_ranks(tool, pb, tmp_path, overlap):
    """`caffe train` as a data-parallel job: two processes (WORLD_SIZE / RANK / LOCAL_RANK as torch.distributed.run sets
    them) on the one visible GPU, gradients over the shared-memory test transport.  Rank 0's sampler draws the global
    batch into a shared-memory ring, each rank trains on its half with the global loss count, the gradients are summed
    before every update.  The result must equal ONE process training on the global batch (same indices, sums
    reassociated), only rank 0 may write snapshots, and both ranks must log every iteration."""
    B, C, Nn, F, D, V, IT = 32, 5, 6, 512, 512, 80, 3         # D = 512: two 256-row blocks of dW (the overlap's unit)
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    src = "synthetic://videos=%d;seed=1701;features=%d" % (V, F)

    def files(tag, batch):
        net_p, sol_p = tmp_path / ("net_%s.prototxt" % tag), tmp_path / ("solver_%s.prototxt" % tag)
        net_p.write_text(train_net(src, batch, C, Nn, D, max_buffer=500, w_std=0.02))
        sol_p.write_text(solver(str(net_p), base_lr=0.01, max_iter=IT, display=1, snapshot=2,
                                snapshot_prefix=str(tmp_path / ("snap_" + tag))))
        return sol_p

    one = files("one", 2 * B)
    run_caffe(["train", "--solver=%s" % one, "--weights=%s" % (tmp_path / "init.caffemodel")], str(tmp_path / "one.log"))
    W1, b1, _ = read_caffemodel(pb, str(tmp_path / ("snap_one_iter_%d.caffemodel" % IT)))

    two = files("two", B)
    procs = []
    transport = "peer" if overlap.startswith("peer-") else "shm"     # peer: the one-shot direct exchange over hipIpc mappings (VV_COMM_PEER)
    overlap = overlap.replace("peer-", "")
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK="0", VV_COMM=transport, VV_COMM_OVERLAP=overlap if overlap != "sharded" else "0",
                   VV_JOB_ID="t%d_%s_%s" % (os.getpid(), overlap, transport), VV_SAMPLER_MODE="node")
        if overlap == "sharded":
            env["VV_COMM_SCHEDULE"] = "sharded"        # (snapshots at iterations 2 and 3: every rank gathers, rank 0 writes)
        procs.append(subprocess.Popen([CAFFE, "train", "--solver=%s" % two, "--weights=%s" % (tmp_path / "init.caffemodel"),
                                       "--gpu=0", "--log_file=%s" % (tmp_path / "two.log")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[1][-3000:]
    W2, b2, _ = read_caffemodel(pb, str(tmp_path / ("snap_two_iter_%d.caffemodel" % IT)))
    e = rel_fro(W2 - W0, W1 - W0)
    print("FACADE-DP overlap=%s: %d-iteration parameter change, 2 ranks vs 1 process on the global batch: %.3e" % (overlap, IT, e))
    # (free-running iterations of this small case amplify the reassociation of the gradient sums, as in
    # test_caffe_train_matches_oracle_trajectory; three iterations keep it near the one-step figure of ~2e-4)
    assert e <= 3e-3 and rel_fro(b2 - b0, b1 - b0) <= 3e-3
    log0, log1 = open(tmp_path / "two.log").read(), open(str(tmp_path / "two.log") + ".rank1").read()
    assert "Data-parallel rank 0 of 2" in log0 and "Data-parallel rank 1 of 2" in log1
    assert len(re.findall(r"Iteration \d+, lr = ", log0)) == IT and len(re.findall(r"Iteration \d+, lr = ", log1)) == IT
    assert "Snapshotting to" in log0 and "Snapshotting to" not in log1
    # each rank reports the loss of its own shard; their mean is the global batch's loss
    l0 = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log0)]
    l1 = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", log1)]
    lg = [float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", open(tmp_path / "one.log").read())]
    assert abs(0.5 * (l0[0] + l1[0]) - lg[0]) <= 1e-4 * lg[0]


def test_caffe_train_data_parallel_per_rank_samplers(tool, pb, oracle, tmp_path):
    """The default data-parallel form: every rank runs the reference's sampler for its own batch (srand(1 + rank), first
    record rank * records / world).  Checked against the oracle driven the same way: each rank's logged loss of iterations
    0 and 1 (iteration 1 sees the update made from the SUM of both ranks' gradients under the global loss count)."""
    B, C, Nn, F, D, V, IT = 32, 5, 6, 256, 64, 80, 2
    W0, b0 = init_weights(3, D, F, std=0.02)
    write_caffemodel(pb, str(tmp_path / "init.caffemodel"), W0, b0)
    src = "synthetic://videos=%d;seed=1701;features=%d" % (V, F)
    net_p, sol_p = tmp_path / "net.prototxt", tmp_path / "solver.prototxt"
    net_p.write_text(train_net(src, B, C, Nn, D, max_buffer=500, w_std=0.02))
    sol_p.write_text(solver(str(net_p), base_lr=0.01, max_iter=IT, display=1, snapshot=0, snapshot_prefix=str(tmp_path / "snap")))
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK="0", VV_COMM="shm", VV_JOB_ID="pr%d" % os.getpid(), VV_DEDUP="0")
        procs.append(subprocess.Popen([CAFFE, "train", "--solver=%s" % sol_p, "--weights=%s" % (tmp_path / "init.caffemodel"),
                                       "--gpu=0", "--log_file=%s" % (tmp_path / "dp.log")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[1][-3000:]
    logs = [open(tmp_path / "dp.log").read(), open(str(tmp_path / "dp.log") + ".rank1").read()]
    assert "own sampler, srand(1), first record 0" in logs[0] and "own sampler, srand(2), first record %d" % (V // 2) in logs[1]
    got = [[float(x) for x in re.findall(r"Iteration \d+, loss = ([0-9.eE+-]+)", l)] for l in logs]

    ds = SyntheticVideos(seed=1701, n_videos=V)
    table = ds.table(F)
    smp = [oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                          max_buffer_size=500, negative_swap_percentage=50, seed=1 + r, initial_cursor=(r * V) // 2) for r in range(2)]
    Wq, bq = W0.copy(), b0.copy()
    hW, hb = np.zeros_like(W0), np.zeros_like(b0)
    for it in range(IT):
        res = [oracle.forward_backward(table, smp[r].next()[0], round_operand(Wq, "f16"), bq, C_=C, Nn=Nn, want=("dW", "db")) for r in range(2)]
        for r in range(2):
            assert abs(got[r][it] - res[r]["loss"]) <= 1e-3 * res[r]["loss"], (it, r, got[r][it], res[r]["loss"])
        lr = oracle.learning_rate("inv", 0.01, 1e-3, 0.75, 0, it)
        oracle.sgd_update(Wq, 0.5 * (res[0]["dW"] + res[1]["dW"]), hW, lr, 1.0, 0.9, 5e-4, 1.0)
        oracle.sgd_update(bq, 0.5 * (res[0]["db"] + res[1]["db"]), hb, lr, 2.0, 0.9, 5e-4, 0.0)


def test_caffe_train_snapshot_diff_at_the_shape_whose_update_rides_in_the_gemm(tool, pb, tmp_path):
    """with `snapshot_diff: true` (caffe.proto SolverParameter field 16; Net::ToProto writes the diffs, net.cpp:784-800)
    Solver::Snapshot reads the parameter gradient back -- which a hinted step at the shipped 4096 x 4096 shape (one split of K: the solver's
    rule applied in the weight-gradient GEMM's epilogue) never stores.  Such a solver now keeps the update as its own launch: the job runs,
    the snapshot carries a diff of the blob's size, and the weights are bit for bit those of the same job without diffs (hinted)."""
    B, C, Nn, F, D, V = 128, 5, 10, 4096, 4096, 300
    net_p = tmp_path / "net.prototxt"
    net_p.write_text(train_net("synthetic://videos=%d;seed=1701;features=%d" % (V, F), B, C, Nn, D, max_buffer=2000, w_std=0.01))
    out = {}
    for tag, diff in (("plain", False), ("diff", True)):
        sol_p = tmp_path / ("solver_%s.prototxt" % tag)
        sol_p.write_text(solver(str(net_p), base_lr=0.01, max_iter=3, display=1, snapshot=2, random_seed=7,
                                snapshot_prefix=str(tmp_path / ("snap_" + tag)), snapshot_diff=diff))
        log = run_caffe(["train", "--solver=%s" % sol_p], str(tmp_path / (tag + ".log")))
        assert "Optimization Done." in log
        net = pb["NetParameter"]()
        net.ParseFromString(open(tmp_path / ("snap_%s_iter_2.caffemodel" % tag), "rb").read())
        fc = [l for l in net.layers if l.name == "fc7"][0]
        out[tag] = (np.array(fc.blobs[0].data, np.float32), np.array(fc.blobs[0].diff, np.float32), np.array(fc.blobs[1].diff, np.float32))
    assert np.array_equal(out["plain"][0], out["diff"][0])
    assert out["plain"][1].size == 0                                  # snapshot_diff false: no diff fields
    dW, db = out["diff"][1], out["diff"][2]
    assert dW.size == D * F and db.size == D and np.isfinite(dW).all() and np.abs(dW).max() > 0 and np.abs(db).max() > 0
