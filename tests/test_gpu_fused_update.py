"""The reduction and the update in one launch (k_reduce_sgd, the lazy reduction of api.hip): bit for bit the parameters,
gradients and losses of the two-launch form (VV_FUSE_UPDATE=0), whoever asks for what in between."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights
B, C, Nn, F, D = 256, 5, 50, 4096, int(sys.argv[4])
ds = SyntheticVideos(seed=7, n_videos=512)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                 max_buffer_size=5000, negative_swap_percentage=50)
W, b = init_weights(7, D, F)
eng = vv.Engine(0, sys.argv[1])
eng.table_synth(ds.seed, ds.n_rows, F)
eng.params_set(W, b)
solver = int(sys.argv[2])
out = {}
losses = []
for it in range(12):
    idx = smp.next()
    idx = idx[0] if isinstance(idx, tuple) else idx
    cfg = vv.StepConfig(B, C, Nn, lr=0.05, momentum=0.0 if solver == 2 else 0.9, weight_decay=5e-4)     # (AdaGrad takes no momentum, solver.hpp:121-122)
    cfg.set("solver_type", solver)
    if solver == 2: cfg.set("delta", 1e-8)
    eng.forward_backward(cfg, idx)
    if it %% 4 == 1: losses.append(eng.loss())              # loss asked for BEFORE the update: the plain reduction runs
    if it %% 4 == 2: out["dW_before_%%d" %% it] = eng.grads()[0]
    eng.apply_update(cfg)
    if it %% 4 == 3: losses.append(eng.loss())              # ... and after it
    if it == 7: out["dW_after_7"], out["db_after_7"] = eng.grads()
    if it == 5: eng.apply_update(cfg)                       # two updates in a row
Wn, bn, hW, hb = eng.params_get()
out.update(W=Wn, b=bn, hW=hW, hb=hb, losses=np.array(losses, dtype=np.float64), stats=np.array(eng.grad_scale_stats(), dtype=np.float64))
np.savez(sys.argv[3], **out)
"""


def _run(tmp_path, prec, solver, fuse, D=512, slab16=0):
    out = tmp_path / ("o_%s_%d_%d_%d_%d.npz" % (prec, solver, fuse, D, slab16))
    env = dict(os.environ, VV_FUSE_UPDATE=str(fuse), VV_SLAB16=str(slab16))
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT, prec, str(solver), str(out), str(D)], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out)


@pytest.mark.parametrize("prec,solver,D,slab16", [("f16", 0, 512, 0), ("f16", 1, 512, 0), ("f16", 2, 512, 0), ("bf16", 0, 512, 0), ("f16", 0, 1024, 0),
                                                  ("f16", 0, 512, 1), ("bf16", 1, 1024, 1)])
def test_fused_update_is_bit_identical(tmp_path, prec, solver, D, slab16):
    """(D = 1024: more 16-byte elements than the fused launch has threads -- its loop.  slab16 = 1, round 6: the split-K partial products as f16 x
    a power of two per tile -- both forms read the same halves and the same factors in the same order)"""
    a, b = _run(tmp_path, prec, solver, 0, D, slab16), _run(tmp_path, prec, solver, 1, D, slab16)
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["W"]).all() and np.abs(a["hW"]).max() > 0


CHILD_HINT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights
B, C, Nn, F, D = 128, 5, 10, 4096, int(sys.argv[4])
hint, drop = int(sys.argv[5]), float(sys.argv[6])
ds = SyntheticVideos(seed=7, n_videos=512)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                 max_buffer_size=5000, negative_swap_percentage=50)
W, b = init_weights(7, D, F)
eng = vv.Engine(0, sys.argv[1])
eng.set_option("wgrad_update", 1)                        # (the default since the rotated epilogue; set explicitly)
eng.table_synth(ds.seed, ds.n_rows, F)
eng.params_set(W, b)
solver = int(sys.argv[2])
out, losses, refused = {}, [], 0
for it in range(10):
    idx = smp.next()
    idx = idx[0] if isinstance(idx, tuple) else idx
    cfg = vv.StepConfig(B, C, Nn, lr=0.05 / (1 + it), momentum=0.0 if solver == 2 else 0.9, weight_decay=5e-4, dropout_ratio=drop, dropout_seed=5)
    cfg.set("solver_type", solver)
    if solver == 2: cfg.set("delta", 1e-8)
    if hint and it != 6: eng.update_hint(cfg)                # (iteration 6: an un-hinted step between hinted ones)
    eng.forward_backward(cfg, idx)
    if it %% 3 == 1: losses.append(eng.loss())              # the loss may be asked for between the two calls
    if hint and it == 4:
        for f in (eng.grads, eng.params_get):               # ... nothing else may
            try:
                f()
            except vv.VVError:
                refused += 1
    eng.apply_update(cfg)
    if it %% 3 == 2: losses.append(eng.loss())
    if hint and it == 5:
        try:
            eng.grads()                                      # the gradient of a hinted, fused step was never stored
        except vv.VVError:
            refused += 1
    if it == 6: out["dW_6"] = eng.grads()[0]                 # (the un-hinted step's gradient is there as ever)
Wn, bn, hW, hb = eng.params_get()
out.update(W=Wn, b=bn, hW=hW, hb=hb, losses=np.array(losses, dtype=np.float64), refused=np.array([refused]))
np.savez(sys.argv[3], **out)
"""


@pytest.mark.parametrize("prec,solver,D,drop", [("f16", 0, 4096, 0.9), ("f16", 1, 4096, 0.0), ("f16", 2, 4096, 0.0), ("bf16", 0, 4096, 0.0), ("f16", 0, 512, 0.0)])
def test_update_in_the_weight_gradient_epilogue_is_bit_identical(tmp_path, prec, solver, D, drop):
    """vv_update_hint: with one split of K -- the shipped 4096 x 4096 matrix at batch 128 -- the weight-gradient GEMM applies
    the solver's rule to the tile in its accumulators and dW is never written.  Parameters, bias, both histories and every loss are bit for bit
    those of the same calls without the hint; between the hinted backward pass and its update only the loss may be read; D = 512 at this batch
    has several splits: the hint changes nothing there (and refuses nothing)."""
    res = {}
    for hint in (0, 1):
        out = tmp_path / ("h_%s_%d_%d_%d.npz" % (prec, solver, D, hint))
        r = subprocess.run([sys.executable, "-c", CHILD_HINT % ROOT, prec, str(solver), str(out), str(D), str(hint), str(drop)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res[hint] = np.load(out)
    a, b = res[0], res[1]
    for k in ("W", "b", "hW", "hb", "losses", "dW_6"):
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["W"]).all() and np.abs(a["hW"]).max() > 0 and np.abs(a["W"] - b["W"]).max() == 0
    assert int(b["refused"][0]) == (3 if D == 4096 else 0), int(b["refused"][0])


CHILD_HINT_EDGES = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import videovector_amd as vv
from videovector_amd.synth import SyntheticVideos, init_weights
B, C, Nn, F, D = 128, 5, 10, 4096, 4096
ds = SyntheticVideos(seed=7, n_videos=512)
smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=B, context_size=C, num_negative_samples=Nn,
                 max_buffer_size=5000, negative_swap_percentage=50)
W, b = init_weights(7, D, F)
eng = vv.Engine(0, "f16")
eng.table_synth(ds.seed, ds.n_rows, F)
eng.params_set(W, b)
cfg = vv.StepConfig(B, C, Nn, lr=0.05, momentum=0.9, weight_decay=5e-4)
idx = smp.next()
idx = idx[0] if isinstance(idx, tuple) else idx
res = {}
# 1. a hint whose backward call returns early (an index out of range) is consumed: the next, un-hinted step keeps its gradient and
#    leaves the parameters alone until vv_apply_update
bad = idx.copy(); bad[0, 0] = ds.n_rows + 5
eng.update_hint(cfg)
try:
    eng.forward_backward(cfg, bad); res["bad_refused"] = 0
except vv.VVError:
    res["bad_refused"] = 1
eng.forward_backward(cfg, idx)
dW = eng.grads()[0]
res["grad_nonzero"] = int(np.abs(dW).max() > 0)
res["params_untouched"] = int(np.array_equal(eng.params_get()[0], W))
eng.apply_update(cfg)
# 2. between a hinted, fused backward pass and its update: no embedding (new half copy, old bias); vv_params_set un-sticks the context
eng.update_hint(cfg)
eng.forward_backward(cfg, idx)
try:
    eng.embed(np.arange(4, dtype=np.int32)); res["embed_refused"] = 0
except vv.VVError:
    res["embed_refused"] = 1
eng.params_set(W, b)
eng.forward_backward(cfg, idx)                    # (was refused for ever: 'call vv_apply_update first')
eng.apply_update(cfg)
res["after_reset_finite"] = int(np.isfinite(eng.params_get()[0]).all())
res["embed_ok"] = int(np.isfinite(eng.embed(np.arange(4, dtype=np.int32))).all())
np.savez(sys.argv[1], **{k: np.array([v]) for k, v in res.items()})
"""


def test_update_hint_edge_cases(tmp_path):
    """a hint is consumed by the backward call that follows it even when that call returns early; vv_params_set clears a hinted
    step that was waiting for its update; vv_embed is refused (VV_ERR_STATE) while the parameters are half-way."""
    out = tmp_path / "edges.npz"
    r = subprocess.run([sys.executable, "-c", CHILD_HINT_EDGES % ROOT, str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    for k in ("bad_refused", "grad_nonzero", "params_untouched", "embed_refused", "after_reset_finite", "embed_ok"):
        assert int(z[k][0]) == 1, k
