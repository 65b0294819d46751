"""GPU tests of the row de-duplication path (videovector_amd/csrc/kernels_dedup.hip): the projection runs
once per distinct table row of the batch and the instances' gradient rows are summed before the
weight-gradient GEMM.  Checked against the dense path of the same library (bit-exact where the arithmetic
is the same, tight where a sum is reassociated) and against the oracle."""
import numpy as np
import pytest

from tests.test_gpu_parity import TOL, check, make_case, rel_fro, round_operand, round_table, run_both, vv  # noqa: F401
from videovector_amd.synth import SyntheticVideos, init_weights

pytestmark = pytest.mark.gpu


def run_mode(vv, prec, table, idx, W, b, C, Nn, dedup, **kw):
    eng = vv.Engine(0, prec)
    eng.set_dedup(dedup)
    eng.table_set(table)
    eng.params_set(W, b)
    cfg = vv.StepConfig(idx.shape[0], C, Nn, **kw)
    eng.forward_backward(cfg, idx)
    out = dict(loss=eng.loss(), stats=eng.dedup_stats(), **eng.blobs(cfg, ip1_diff=True))
    out["dW"], out["db"] = eng.grads()
    return eng, cfg, out


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_dedup_equals_dense_on_a_heavily_repeated_batch(vv, oracle, prec):
    B, C, Nn, F, D = 96, 5, 20, 384, 320
    ds, table, idx, W, b = make_case(21, 6, B, C, Nn, F, D, wstd=0.02)      # ~240 table rows, 2400 batch rows
    idx[3, 2] = -1; idx[17, 9] = -1                                         # empty slots de-duplicate too
    _, _, dn = run_mode(vv, prec, table, idx, W, b, C, Nn, False)
    _, _, dd = run_mode(vv, prec, table, idx, W, b, C, Nn, True)
    n_unique = len(np.unique(idx))
    assert dn["stats"] == (B * (C + Nn), B * (C + Nn))
    assert dd["stats"] == (B * (C + Nn), n_unique) and n_unique < 300
    # same arithmetic, shared: embeddings, scores, loss and per-instance gradients are bit-identical
    for k in ("ip2", "target_score", "negative_scores", "ip1_diff", "db"):
        assert np.array_equal(dn[k], dd[k]), k
    assert dn["loss"] == dd["loss"]
    # dW: the sum over instances is taken before the 16-bit rounding and the GEMM instead of inside it
    assert rel_fro(dd["dW"], dn["dW"]) <= (5e-4 if prec == "f16" else 4e-3)
    # and against the oracle, like every other parity case
    eng, cfg, got, ref = run_both(vv, oracle, prec, table, idx, W, b, C, Nn)
    assert eng.dedup_stats()[1] == n_unique
    check(got, ref, TOL[prec], "dedup-" + prec)


def test_dedup_degenerate_batches(vv, oracle):
    B, C, Nn, F, D = 40, 3, 6, 256, 64
    ds, table, idx, W, b = make_case(5, 30, B, C, Nn, F, D, wstd=0.05)
    # (a) every slot names the same row: one segment holding all B*(C+Nn) instances
    same = np.full_like(idx, 7)
    _, _, dn = run_mode(vv, "f16", table, same, W, b, C, Nn, False)
    _, _, dd = run_mode(vv, "f16", table, same, W, b, C, Nn, True)
    assert dd["stats"] == (B * (C + Nn), 1)
    assert np.array_equal(dn["ip2"], dd["ip2"]) and dn["loss"] == dd["loss"]
    assert np.allclose(dd["dW"], dn["dW"], rtol=0, atol=2e-3 * np.abs(dn["dW"]).max() + 1e-12)
    # (b) no repeats at all: identity mapping
    uniq = np.arange(B * (C + Nn), dtype=np.int32).reshape(B, C + Nn) % len(table)
    _, _, dn = run_mode(vv, "f16", table, uniq, W, b, C, Nn, False)
    _, _, dd = run_mode(vv, "f16", table, uniq, W, b, C, Nn, True)
    assert dd["stats"][1] == len(np.unique(uniq))
    assert np.array_equal(dn["ip2"], dd["ip2"]) and rel_fro(dd["dW"], dn["dW"]) <= 5e-4


def test_dedup_is_deterministic_and_survives_shape_changes(vv):
    F, D = 512, 128
    ds = SyntheticVideos(seed=4, n_videos=12)
    table = ds.table(F)
    W, b = init_weights(9, D, F, std=0.02)
    eng = vv.Engine(0, "f16")
    eng.table_set(table); eng.params_set(W, b)
    seen = {}
    for rep in range(3):
        for (B, C, Nn) in [(64, 5, 12), (33, 3, 7), (64, 5, 12)]:
            idx = np.random.default_rng(B + rep * 0).integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
            cfg = vv.StepConfig(B, C, Nn)
            eng.forward_backward(cfg, idx)
            dW = eng.grads()[0].copy()
            key = (B, C, Nn)
            if key in seen:
                assert np.array_equal(seen[key], dW), key       # f64 segment sums: arrival order cannot matter
            seen[key] = dW


def test_dedup_sgd_trajectory_matches_dense(vv):
    B, C, Nn, F, D = 128, 5, 10, 512, 256
    ds = SyntheticVideos(seed=1701, n_videos=30)
    table = ds.table(F)
    W, b = init_weights(3, D, F, std=0.02)
    rng = np.random.default_rng(0)
    batches = [rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32) for _ in range(10)]
    res = []
    for mode in (False, True):
        eng = vv.Engine(0, "f16")
        eng.set_dedup(mode)
        eng.table_set(table); eng.params_set(W, b)
        cfg = vv.StepConfig(B, C, Nn, lr=0.01)
        losses = []
        for idx in batches:
            eng.step(cfg, idx)
            losses.append(eng.loss()[0])
        res.append((np.array(losses), eng.params_get()[0]))
    assert np.allclose(res[0][0], res[1][0], rtol=2e-5)
    assert rel_fro(res[1][1], res[0][1]) <= 5e-4


@pytest.mark.parametrize("dedup", [False, True])
def test_out_of_range_device_indices_read_the_zero_row(vv, dedup):
    # a device-resident index array cannot be validated on the host: entries outside the table behave like -1
    import torch
    B, C, Nn, F, D = 16, 3, 4, 256, 64
    ds, table, idx, W, b = make_case(2, 10, B, C, Nn, F, D, wstd=0.05)
    bad = idx.copy()
    bad[1, 3] = 10 ** 9; bad[5, 0] = -7; bad[9, 6] = len(table)
    ok = bad.copy()
    ok[1, 3] = ok[5, 0] = ok[9, 6] = -1
    outs = []
    for arr in (bad, ok):
        eng = vv.Engine(0, "f16")
        eng.set_dedup(dedup)
        eng.table_set(table); eng.params_set(W, b)
        t = torch.from_numpy(arr).to("cuda:0")
        torch.cuda.synchronize()
        cfg = vv.StepConfig(B, C, Nn)
        eng.forward_backward(cfg, idx_dev_ptr=t.data_ptr())
        outs.append((eng.loss(), eng.grads()[0].copy()))
    assert outs[0][0] == outs[1][0] and np.array_equal(outs[0][1], outs[1][1])


def test_heavily_repeated_row_with_tiny_norm_does_not_poison_the_gradient():
    """The per-distinct-row gradient SUM is stored as 16 bits.  A row repeated thousands of times in a batch whose
    embedding norm is tiny (gradients ~ 1 / |x|) can push that sum past 65504 in f16 at the usual scale.  Such a value
    is never applied clipped: the kernel that rounds records its maxima, and its conditional repeat produces the sums
    again at a smaller power-of-two scale BEFORE the weight-gradient product runs (vv_internal.h: GradGuard) -- the
    very first step is already right; the host then lowers the scale so that later steps need no repeat."""
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F, D = 64, 5, 40, 128, 64
    ds = SyntheticVideos(seed=2, n_videos=30)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
    idx[:, C:] = 7                                   # ONE negative row, B * Nn = 2560 instances
    for wstd in (1e-3, 1e-7):                        # ordinary embeddings / embeddings with a tiny norm
        W, b = init_weights(2, D, F, std=wstd)
        cfg = vv.StepConfig(B, C, Nn)
        dense = vv.Engine(0, "f16")
        dense.table_synth(ds.seed, ds.n_rows, F); dense.params_set(W, b); dense.set_dedup(False)
        eng = vv.Engine(0, "f16")
        eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W, b); eng.set_dedup(True)
        errs = []
        for step in range(10):                       # the same batch again and again
            dense.forward_backward(cfg, idx)
            dWd, _ = dense.grads()
            eng.forward_backward(cfg, idx)
            dW, db = eng.grads()
            rows, uniq = eng.dedup_stats()
            assert uniq < rows // 5 and np.isfinite(dW).all() and np.isfinite(db).all() and np.isfinite(dWd).all()
            errs.append(np.linalg.norm(dW - dWd) / max(np.linalg.norm(dWd), 1e-30))
        print("DEDUP heavy-repeat wstd %g: |dW| %.3e, dedup vs dense per step %s; repeats %s" % (
            wstd, np.linalg.norm(dWd), " ".join("%.1e" % e for e in errs), eng.grad_scale_stats()))
        assert max(errs) <= 2e-3, errs               # the first step included
        dense.close(); eng.close()


@pytest.mark.parametrize("path", ["seg", "segsum", "dense"])
def test_f16_gradient_never_applied_clipped(path, monkeypatch):
    """Gradients 3e5 times their usual size (loss_weight) leave f16's range at the usual scale.  Every path that rounds
    gradients to 16 bits -- the segment-wise backward, the row-writing kernels + k_segsum, the dense rows -- produces
    them again at a smaller scale inside the SAME step: the first step is already the unit-weight gradient times 3e5,
    the device reports the repeats, and the host's scale follows so that they stop."""
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    if path == "segsum":
        monkeypatch.setenv("VV_SEG_BWD", "0")
    dedup = path != "dense"
    B, C, Nn, F, D = 64, 5, 10, 256, 512
    ds = SyntheticVideos(seed=6, n_videos=20)
    idx = np.random.default_rng(1).integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
    W, b = init_weights(4, D, F, std=0.02)
    ref = vv.Engine(0, "f16")
    ref.table_synth(ds.seed, ds.n_rows, F); ref.params_set(W, b); ref.set_dedup(dedup)
    ref.forward_backward(vv.StepConfig(B, C, Nn), idx)
    dW1, db1 = ref.grads()
    assert ref.grad_scale_stats()[0] == 0
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W, b); eng.set_dedup(dedup)
    cfg = vv.StepConfig(B, C, Nn, loss_weight=3e5)
    errs, errs_b = [], []
    for _ in range(12):
        eng.forward_backward(cfg, idx)
        dW, db = eng.grads()
        assert np.isfinite(dW).all() and np.isfinite(db).all()
        errs.append(float(np.linalg.norm(dW - 3e5 * dW1) / np.linalg.norm(3e5 * dW1)))
        errs_b.append(float(np.linalg.norm(db - 3e5 * db1) / np.linalg.norm(3e5 * db1)))
    repeats, scale = eng.grad_scale_stats()
    print("F16 GUARD %s: dW error per step %s; repeats reported %d, scale now %g" % (path, " ".join("%.1e" % e for e in errs), repeats, scale))
    assert max(errs) <= 2e-3 and max(errs_b) <= 2e-3, (errs, errs_b)      # the first step included
    if path == "seg":
        assert repeats == 0                  # proactive: the score kernel bounds the gradients, k_seg_bwd settles the scale before it rounds
    else:
        assert 1 <= repeats <= 6             # the first steps repeated (reports arrive four steps late), then the scale had followed
    ip1 = eng.blobs(cfg, ip2=False, scores=False, ip1_diff=True)["ip1_diff"]
    ip1_ref = ref.blobs(cfg, ip2=False, scores=False, ip1_diff=True)["ip1_diff"]
    assert np.linalg.norm(ip1 - 3e5 * ip1_ref) <= 4e-3 * np.linalg.norm(3e5 * ip1_ref)
    ref.close(); eng.close()


def test_f16_guard_tiny_gradients_are_scaled_up():
    """The other side of the range: gradients 1e-7 times their usual size would sink into f16's subnormals at the
    count-based default scale; the segment-wise backward settles its scale on the device from a bound on the step's own
    gradients, in both directions, before it rounds anything: the first step is already right."""
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F, D = 64, 5, 10, 256, 512
    ds = SyntheticVideos(seed=6, n_videos=20)
    idx = np.random.default_rng(1).integers(0, ds.n_rows, size=(B, C + Nn)).astype(np.int32)
    W, b = init_weights(4, D, F, std=0.02)
    ref = vv.Engine(0, "f16")
    ref.table_synth(ds.seed, ds.n_rows, F); ref.params_set(W, b)
    ref.forward_backward(vv.StepConfig(B, C, Nn), idx)
    dW1, _ = ref.grads()
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, loss_weight=1e-7)
    errs = []
    for _ in range(8):
        eng.forward_backward(cfg, idx)
        dW, _ = eng.grads()
        errs.append(float(np.linalg.norm(dW - 1e-7 * dW1) / np.linalg.norm(1e-7 * dW1)))
    print("F16 GUARD tiny: dW error per step %s; scale now %g" % (" ".join("%.1e" % e for e in errs), eng.grad_scale_stats()[1]))
    assert max(errs) <= 2e-3 and eng.grad_scale_stats()[0] == 0       # the first step included: the scale is settled on the device
    ref.close(); eng.close()


def test_kernel_timing_hook_samples_the_nth_steps_and_honours_the_selection(vv):
    """vv_profile_enable(N): the N-th, 2N-th, ... step after the call carry timing events; vv_profile_select restricts
    them to the named kernels (bench.py times the two GEMMs inside its timed region)."""
    B, C, Nn, F, D = 64, 5, 10, 256, 512
    ds, table, idx, W, b = make_case(5, 30, B, C, Nn, F, D, wstd=0.02)
    eng = vv.Engine(0, "f16")
    eng.table_set(table)
    eng.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn)
    for _ in range(3):
        eng.step(cfg, idx)
    eng.profile_select(("fwd_gemm", "wgrad_gemm"))
    eng.profile_enable(4)
    for _ in range(10):
        eng.step(cfg, idx)
    got = {k: eng.profile_get(k) for k in ("dedup", "fwd_gemm", "score_loss", "segsum", "wgrad_gemm", "reduce", "sgd", "reduce_sgd")}
    assert got["fwd_gemm"][1] == 2 and got["wgrad_gemm"][1] == 2          # steps 4 and 8 of the 10
    assert got["fwd_gemm"][0] > 0 and got["wgrad_gemm"][0] > 0
    assert all(got[k][1] == 0 for k in ("dedup", "score_loss", "segsum", "reduce", "sgd", "reduce_sgd"))
    eng.profile_select(None)
    eng.profile_enable(1)
    eng.step(cfg, idx)
    assert all(eng.profile_get(k)[1] == 1 for k in ("dedup", "fwd_gemm", "score_loss", "segsum", "wgrad_gemm"))
    # the reduction and the update: one launch (k_reduce_sgd, the default) or two (VV_FUSE_UPDATE=0)
    n = {k: eng.profile_get(k)[1] for k in ("reduce", "sgd", "reduce_sgd")}
    assert n in ({"reduce": 0, "sgd": 0, "reduce_sgd": 1}, {"reduce": 1, "sgd": 1, "reduce_sgd": 0})
    eng.profile_enable(False)
    eng.close()


def test_pending_scale_update_is_flushed_by_whatever_comes_next(vv):
    """The W -> half scale update of an SGD step rides in the NEXT step's slab reduction; two updates in a row, a
    parameter read-back or new parameters in between must give the same weights as the plain sequence."""
    B, C, Nn, F, D = 64, 5, 10, 256, 512
    ds, table, idx, W, b = make_case(6, 30, B, C, Nn, F, D, wstd=0.02)
    cfg = vv.StepConfig(B, C, Nn, lr=0.5)                      # large steps: max |W| moves, the f16 scale with it

    def run(variant):
        eng = vv.Engine(0, "f16")
        eng.table_set(table)
        eng.params_set(W, b)
        for it in range(6):
            eng.forward_backward(cfg, idx)
            eng.apply_update(cfg)
            if variant == "double" and it == 2:
                eng.apply_update(cfg)                          # same gradients applied twice: the pending update is flushed first
            if variant == "readback" and it == 2:
                eng.params_get()
        out = eng.params_get()
        emb = eng.embed(idx[:8, 0].copy())
        eng.close()
        return out, emb

    (Wa, ba, _, _), ea = run("plain")
    (Wr, br, _, _), er = run("readback")
    assert np.array_equal(Wa, Wr) and np.array_equal(ba, br) and np.array_equal(ea, er)
    (Wd, bd, _, _), ed = run("double")
    assert np.isfinite(Wd).all() and np.isfinite(ed).all() and not np.array_equal(Wd, Wa)


def test_guard_long_run_on_the_benchmark_stream():
    """3000 training steps of the benchmark's own workload (BASELINE configs[1]: batch 1024, C5, Nn50, 4096 -> 512, the
    reference sampler's stream, shipped solver schedule): no step needs a repeat (the scale follows the reported maxima
    well before anything leaves f16's range), and every 250th step's dW agrees with a dense engine evaluated at the same
    parameters on the same batch."""
    import videovector_amd as vv
    from videovector_amd.synth import SyntheticVideos, init_weights
    B, C, Nn, F, D = 1024, 5, 50, 4096, 512
    ds = SyntheticVideos(seed=1701, n_videos=2048)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                     max_buffer_size=5000, negative_swap_percentage=50)
    W, b = init_weights(1701, D, F)
    eng = vv.Engine(0, "f16"); eng.table_synth(ds.seed, ds.n_rows, F); eng.params_set(W, b)
    dense = vv.Engine(0, "f16"); dense.table_synth(ds.seed, ds.n_rows, F); dense.params_set(W, b); dense.set_dedup(False)
    cfg = vv.StepConfig(B, C, Nn)
    errs, scales = [], set()
    for it in range(3000):
        idx = smp.next()
        cfg.set("lr", 1e-3 * (1.0 + 1e-3 * it) ** -0.75)
        eng.forward_backward(cfg, idx)
        if it % 250 == 249:
            Wc, bc, hW, hb = eng.params_get()
            dense.params_set(Wc, bc)
            dense.forward_backward(cfg, idx)
            dW, db = eng.grads()
            dWd, dbd = dense.grads()
            errs.append(float(np.linalg.norm(dW - dWd) / np.linalg.norm(dWd)))
            assert np.linalg.norm(db - dbd) <= 2e-3 * np.linalg.norm(dbd)
            scales.add(eng.grad_scale_stats()[1])
        eng.apply_update(cfg)
    repeats, scale = eng.grad_scale_stats()
    print("GUARD long run: dW dedup vs dense every 250th step %s; repeats %d; scales seen %s" % (
        " ".join("%.1e" % e for e in errs), repeats, sorted(scales)))
    assert repeats == 0 and max(errs) <= 2e-3
    smp.close(); eng.close(); dense.close()
