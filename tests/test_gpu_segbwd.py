"""GPU tests of the segment-wise backward of de-duplicated batches (kernels_elem.hip: k_score_fwd + k_seg_bwd): the
gradient of ip1_nonorm stays factored per instance and is summed per DISTINCT row; the per-instance 16-bit rows are never
written.  Checked against the path that writes them (VV_SEG_BWD=0: k_score_loss_reg + k_segsum), against the dense
path, and against the oracle.  The pair is built for the fc width of the hot path (D = 512)."""
import os

import numpy as np
import pytest

from tests.test_gpu_parity import TOL, check, make_case, rel_fro, run_both, vv  # noqa: F401

pytestmark = pytest.mark.gpu

D = 512


def run(vv, prec, table, idx, W, b, C, Nn, seg, dedup=True, **kw):
    return run_d(vv, prec, table, idx, W, b, C, Nn, seg, dedup, **kw)


def run_d(vv, prec, table, idx, W, b, C, Nn, seg, dedup=True, **kw):
    os.environ["VV_SEG_BWD"] = "1" if seg else "0"
    try:
        eng = vv.Engine(0, prec)
    finally:
        del os.environ["VV_SEG_BWD"]
    eng.set_dedup(dedup)
    eng.table_set(table)
    eng.params_set(W, b)
    cfg = vv.StepConfig(idx.shape[0], C, Nn, **kw)
    eng.forward_backward(cfg, idx)
    dW, db = eng.grads()                 # before the debug accessor below re-runs the row-writing kernel
    out = dict(loss=eng.loss(), stats=eng.dedup_stats(), dW=dW, db=db, **eng.blobs(cfg, ip1_diff=True))
    eng.close()
    return out


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("C,Nn", [(5, 10), (5, 20), (5, 50), (2, 3), (7, 55)])
def test_segment_wise_backward_matches_the_row_writing_path(vv, oracle, prec, C, Nn, fp32_ip2):
    B, F = 64, 256
    ds, table, idx, W, b = make_case(21, 8, B, C, Nn, F, D, wstd=0.02)      # ~300 table rows: heavy repeats
    idx[3, 1] = -1; idx[17, C + 1] = -1
    a = run(vv, prec, table, idx, W, b, C, Nn, seg=False)
    s = run(vv, prec, table, idx, W, b, C, Nn, seg=True)
    d = run(vv, prec, table, idx, W, b, C, Nn, seg=True, dedup=False)       # dense: the flag is ignored
    assert a["stats"] == s["stats"] and s["stats"][1] == len(np.unique(idx))
    # one forward, two kernels: the embeddings are the same array; the scores and the loss may differ in the last bit
    # (the compiler contracts the two kernels' dot products differently)
    assert np.array_equal(a["ip2"], s["ip2"])
    for k in ("target_score", "negative_scores"):
        assert np.abs(a[k] - s[k]).max() <= 2.5e-7, k
    assert abs(a["loss"][0] - s["loss"][0]) <= 1e-6 * a["loss"][0] and a["loss"][1] == s["loss"][1]
    assert rel_fro(s["ip1_diff"], a["ip1_diff"]) <= 1e-6                    # rebuilt on demand by the row-writing kernel
    # the sums over a row's instances are taken in fp32 before ONE rounding to 16 bits (the other path rounds every
    # instance row first): both are within the rounding of the dense result
    tol = 5e-4 if prec == "f16" else 4e-3
    assert rel_fro(s["dW"], d["dW"]) <= tol and rel_fro(a["dW"], d["dW"]) <= tol
    assert rel_fro(s["db"], d["db"]) <= 1e-5 and rel_fro(s["db"], a["db"]) <= 1e-5
    print("SEGBWD %s C=%d Nn=%d: dW vs dense seg %.3e rows %.3e; db %.3e" % (prec, C, Nn, rel_fro(s["dW"], d["dW"]),
                                                                             rel_fro(a["dW"], d["dW"]), rel_fro(s["db"], d["db"])))


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_segment_wise_backward_against_the_oracle(vv, oracle, prec):
    B, C, Nn, F = 96, 5, 20, 384
    ds, table, idx, W, b = make_case(3, 6, B, C, Nn, F, D, wstd=0.02)
    w = (1 + np.arange(B) % 4).astype(np.float32)
    for kw in (dict(), dict(norm=1, margin=0.5), dict(item_weight=w, loss_weight=2.0), dict(global_count=4 * B * Nn),
               dict(ctx_coeff=[0.1, 0.2, 0.3, 0.4])):
        eng, cfg, got, ref = run_both(vv, oracle, prec, table, idx, W, b, C, Nn, **kw)
        assert eng.dedup_stats()[1] < eng.dedup_stats()[0]
        check(got, ref, TOL[prec], "segbwd-%s-%s" % (prec, ",".join(kw) or "default"))
        eng.close()


def test_segment_wise_backward_degenerate_batches(vv, fp32_ip2):
    B, C, Nn, F = 40, 3, 6, 256
    ds, table, idx, W, b = make_case(5, 30, B, C, Nn, F, D, wstd=0.05)
    # (a) one segment holding every instance (360 records on one wave); (b) no repeats; (c) a batch of one item
    same = np.full_like(idx, 7)
    uniq = np.arange(B * (C + Nn), dtype=np.int32).reshape(B, C + Nn) % len(table)
    for name, ix in (("same", same), ("uniq", uniq), ("one", idx[:1])):
        d = run(vv, "f16", table, ix, W, b, C, Nn, seg=True, dedup=False)
        s = run(vv, "f16", table, ix, W, b, C, Nn, seg=True)
        assert np.array_equal(d["ip2"], s["ip2"]) and abs(d["loss"][0] - s["loss"][0]) <= 1e-6 * d["loss"][0]
        assert np.isfinite(s["dW"]).all()
        if name == "same":
            # every instance's gradient is zero in exact arithmetic (all rows equal: s Ah - t x = 0); what either path
            # leaves is rounding noise of terms of size ~1, far below a real gradient (~1e-3 here)
            assert np.abs(s["dW"]).max() <= 1e-6 and np.abs(d["dW"]).max() <= 1e-6 and np.abs(s["db"]).max() <= 1e-5
            continue
        assert np.abs(s["dW"] - d["dW"]).max() <= 2e-3 * np.abs(d["dW"]).max() + 1e-12, name
        assert np.abs(s["db"] - d["db"]).max() <= 1e-5 * np.abs(d["db"]).max() + 1e-12, name


def test_segment_wise_backward_sgd_trajectory(vv):
    """Free-running SGD on the same batches: the factored backward stays with the row-writing one."""
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F = 128, 5, 10, 512
    ds = SyntheticVideos(seed=4, n_videos=40)
    W0, b0 = init_weights(3, D, F, std=0.02)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                     max_buffer_size=400)
    batches = [smp.next() for _ in range(12)]
    res = []
    for seg in (False, True):
        os.environ["VV_SEG_BWD"] = "1" if seg else "0"
        eng = vv.Engine(0, "f16")
        del os.environ["VV_SEG_BWD"]
        eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W0, b0)
        cfg = vv.StepConfig(B, C, Nn, lr=0.01)
        losses = []
        for ix in batches:
            eng.step(cfg, ix)
            losses.append(eng.loss()[0])
        res.append((losses, eng.params_get()[0]))
        eng.close()
    print("SEGBWD trajectory: loss diff %.2e, W diff %.2e" % (max(abs(x - y) / y for x, y in zip(*[r[0] for r in res])),
                                                              rel_fro(res[1][1], res[0][1])))
    assert all(abs(x - y) <= 1e-4 * y for x, y in zip(res[0][0], res[1][0]))
    print("SEGBWD trajectory step diff %.3e" % rel_fro(res[1][1] - W0, res[0][1] - W0))
    assert rel_fro(res[1][1] - W0, res[0][1] - W0) <= 1e-2      # free-running: the two roundings of dYu drift apart slowly


def test_segment_wise_backward_is_bit_reproducible(vv, fp32_ip2):
    """The records of a distinct row arrive in a different order in every run (an atomic counter hands out the
    positions); the sums must not depend on it: segments up to 64 instances are summed in instance order, longer ones
    with order-independent (exact) sums."""
    B, C, Nn, F = 128, 5, 50, 256
    ds, table, idx, W, b = make_case(11, 8, B, C, Nn, F, D, wstd=0.02)      # ~300 rows: segments of ~20 instances
    idx[:, C + 7] = 5                                                       # one row 128+ times: a long segment
    idx[:40, C + 9] = 6                                                     # and one of 40+
    outs = [run(vv, "f16", table, idx, W, b, C, Nn, seg=True) for _ in range(4)]
    for o in outs[1:]:
        assert np.array_equal(o["dW"], outs[0]["dW"]) and np.array_equal(o["db"], outs[0]["db"])
        assert o["loss"] == outs[0]["loss"]
    d = run(vv, "f16", table, idx, W, b, C, Nn, seg=True, dedup=False)
    assert rel_fro(outs[0]["dW"], d["dW"]) <= 5e-4 and rel_fro(outs[0]["db"], d["db"]) <= 1e-5


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("Dx,C,Nn", [(1024, 5, 20), (1024, 4, 200), (512, 5, 70), (512, 9, 10)])
def test_segment_wise_backward_streaming_forward_shapes(vv, oracle, prec, Dx, C, Nn, fp32_ip2):
    """Shapes the register-resident forward does not hold (D = 1024; more than 56 target / negative rows; more than 6
    context rows) take the streaming kernel's segment-wise form and the two-chunk k_seg_bwd: the per-GPU shape of
    BASELINE configs[4] is one of them."""
    B, F = 48, 256
    ds, table, idx, W, b = make_case(31, 8, B, C, Nn, F, Dx, wstd=0.02)
    idx[5, 1] = -1
    a = run_d(vv, prec, table, idx, W, b, C, Nn, seg=False)
    s = run_d(vv, prec, table, idx, W, b, C, Nn, seg=True)
    d = run_d(vv, prec, table, idx, W, b, C, Nn, seg=True, dedup=False)
    assert np.array_equal(a["ip2"], s["ip2"])
    for k in ("target_score", "negative_scores"):
        assert np.abs(a[k] - s[k]).max() <= 2.5e-7, k
    assert abs(a["loss"][0] - s["loss"][0]) <= 1e-6 * a["loss"][0] and a["loss"][1] == s["loss"][1]
    tol = 5e-4 if prec == "f16" else 4e-3
    assert rel_fro(s["dW"], d["dW"]) <= tol and rel_fro(s["db"], d["db"]) <= 1e-5
    s2 = run_d(vv, prec, table, idx, W, b, C, Nn, seg=True)
    assert np.array_equal(s["dW"], s2["dW"]) and np.array_equal(s["db"], s2["db"])        # bit-reproducible
    print("SEGBWD streaming %s D=%d C=%d Nn=%d: dW vs dense %.3e (rows path %.3e)" % (prec, Dx, C, Nn, rel_fro(s["dW"], d["dW"]),
                                                                                     rel_fro(a["dW"], d["dW"])))
    if prec == "f16":
        eng, cfg, got, ref = run_both(vv, oracle, prec, table, idx, W, b, C, Nn)
        check(got, ref, TOL[prec], "segbwd-streaming-D%d-C%d-Nn%d" % (Dx, C, Nn))
        eng.close()
