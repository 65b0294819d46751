"""k_wgrad_gemm_ph's LEAN instantiations (round 6, option wgrad_lean: gathered rows addressed as scalar base + 32-bit lane offset, LDS addresses
as immediates, X_hi issued a segment early, a K loop without end-of-stream tests) against the 64-bit-address instantiations they replace:
the same products in the same order -- bit-identical dW, db, parameters, whatever the K-tile count of a split (odd, even, 1, 2, 3) and
whichever epilogue (fp32 slabs, f16 slabs, the update in the epilogue)."""
import numpy as np
import pytest

from tests.test_gpu_parity import make_case, vv  # noqa: F401

pytestmark = pytest.mark.gpu


def _grads(vv, prec, table, idx, W, b, C, Nn, lean, dedup, slab16, steps=1, hint=False, **kw):
    eng = vv.Engine(0, prec)
    eng.set_option("wgrad_lean", lean)
    eng.set_option("slab16", slab16)
    eng.set_dedup(dedup)
    eng.table_set(table)
    eng.params_set(W, b)
    cfg = vv.StepConfig(idx.shape[0], C, Nn, lr=0.01, momentum=0.9, **kw)
    out = {}
    for it in range(steps):
        if hint:
            eng.update_hint(cfg)
        eng.forward_backward(cfg, idx)
        if not hint and it == steps - 1:
            out["dW"], out["db"] = eng.grads()
        eng.apply_update(cfg)
    out["W"], out["b"], out["hW"], out["hb"] = eng.params_get()
    out["loss"] = eng.loss()
    assert int(eng.get_option("wgrad_lean")) == lean
    eng.close()
    return out


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("B,Nn,dedup,slab16", [(8, 3, True, 1), (24, 10, True, 1), (64, 20, True, 0), (100, 50, True, 1), (128, 50, False, 1),
                                               (257, 7, False, 0), (512, 50, True, 1)])
def test_lean_weight_gradient_is_bit_identical(vv, prec, B, Nn, dedup, slab16):
    """Row counts from one K-tile per split (a handful of rows: most splits empty) to hundreds: every tail of the K loop."""
    C, F, D = 5, 512, 512
    ds, table, idx, W, b = make_case(17 + B, 40, B, C, Nn, F, D, wstd=0.03)
    idx[1, 0] = -1
    a = _grads(vv, prec, table, idx, W, b, C, Nn, 0, dedup, slab16)
    l = _grads(vv, prec, table, idx, W, b, C, Nn, 1, dedup, slab16)
    for k in ("dW", "db", "W", "b", "hW", "hb"):
        assert np.array_equal(a[k], l[k]), k
    assert a["loss"] == l["loss"] and np.abs(a["dW"]).max() > 0


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_lean_update_in_the_epilogue_is_bit_identical(vv, prec):
    """One split of K and vv_update_hint: the UPD instantiation (the shipped 4096 x 4096 shape's), three steps."""
    B, C, Nn, F, D = 128, 5, 10, 4096, 4096
    ds, table, idx, W, b = make_case(5, 60, B, C, Nn, F, D, wstd=0.01)
    a = _grads(vv, prec, table, idx, W, b, C, Nn, 0, False, 0, steps=3, hint=True)
    l = _grads(vv, prec, table, idx, W, b, C, Nn, 1, False, 0, steps=3, hint=True)
    for k in ("W", "b", "hW", "hb"):
        assert np.array_equal(a[k], l[k]), k
    assert a["loss"] == l["loss"]


def test_lean_offsets_above_2_gib_and_the_fallback_past_4_gib(vv):
    """The lean form's lane offsets are UNSIGNED 32-bit byte offsets from the table's base: rows between 2 and 4 GiB into a 3.85 GB table
    must come out as with 64-bit addresses; a table past 4 GiB takes the 64-bit instantiations by itself (the option stays on)."""
    B, C, Nn, F, D = 64, 5, 20, 4096, 512
    rng = np.random.default_rng(3)
    from videovector_amd.synth import init_weights
    W, b = init_weights(3, D, F, std=0.02)
    res = {}
    for n_rows, lo in ((470000, 300000), (560000, 300000)):      # 3.85 GB: lean; 4.59 GB: not lean
        idx = rng.integers(lo, n_rows, size=(B, C + Nn)).astype(np.int32)
        idx[:, 0] = n_rows - 1 - np.arange(B)                     # the very last rows of the table
        for lean in (0, 1):
            eng = vv.Engine(0, "f16")
            eng.set_option("wgrad_lean", lean)
            eng.table_synth(11, n_rows, F)
            eng.params_set(W, b)
            cfg = vv.StepConfig(B, C, Nn)
            eng.forward_backward(cfg, idx)
            res[(n_rows, lean)] = eng.grads() + (eng.loss(),)
            eng.close()
        a, l = res[(n_rows, 0)], res[(n_rows, 1)]
        assert np.array_equal(a[0], l[0]) and np.array_equal(a[1], l[1]) and a[2] == l[2], n_rows
        assert np.abs(a[0]).max() > 0 and np.isfinite(a[0]).all()
