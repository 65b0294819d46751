"""Round 6: the two 16-bit intermediates of the de-duplicated step and what they cost.

  h16 (default on)   ip2 between the forward GEMM and the segment-wise pair stored as f16 (scale 1, saturated at 65504): the SAME accumulators,
                     rounded once more -- so every stored value is EXACTLY the f16 rounding of the fp32 form's value; scores, loss and the
                     gradients move by what that rounding is worth (2^-12 relative per element).
  slab16 (default on) the weight gradient's split-K partial products as f16 x one power of two per (split, 256 x 256 tile): the forward
                     pass and db are untouched, dW moves by 2^-12 of each partial product.
Their bounds against the fp32 ORACLE are the whole-batch tests' (tests/test_gpu_fullsize.py, tests/test_gpu_cfg5.py); here: against the fp32
forms of the same engine, on the shapes of both score kernels (D = 512 register-resident, D = 1024 one sweep)."""
import numpy as np
import pytest

from tests.test_gpu_parity import make_case, rel_fro, vv  # noqa: F401

pytestmark = pytest.mark.gpu


def run(vv, prec, table, idx, W, b, C, Nn, **opts):
    eng = vv.Engine(0, prec)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.table_set(table)
    eng.params_set(W, b)
    cfg = vv.StepConfig(idx.shape[0], C, Nn)
    eng.forward_backward(cfg, idx)
    eng.forward_backward(cfg, idx)            # (the second call: the tile plan of a running job)
    dW, db = eng.grads()
    out = dict(loss=eng.loss(), dW=dW, db=db, **eng.blobs(cfg))
    eng.close()
    return out


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("D,B,C,Nn", [(512, 256, 5, 50), (1024, 64, 5, 200), (512, 96, 3, 7)])
def test_ip2_as_f16_is_the_rounding_of_the_fp32_form(vv, prec, D, B, C, Nn):
    ds, table, idx, W, b = make_case(23, 64, B, C, Nn, 1024, D, wstd=0.02)
    idx[:, C:] = idx[:, C:] % 900                                   # repeated negatives: a de-duplicated batch
    a = run(vv, prec, table, idx, W, b, C, Nn, h16=0, slab16=0)
    h = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=0)
    assert np.abs(a["ip2"]).max() < 6e4                              # (inside f16's range: nothing saturates here)
    assert np.array_equal(h["ip2"], a["ip2"].astype(np.float16).astype(np.float32))
    e_s = max(np.abs(h["target_score"] - a["target_score"]).max(), np.abs(h["negative_scores"] - a["negative_scores"]).max())
    print("H16 %s D %d: scores %.2e, loss %.7f / %.7f, dW %.2e db %.2e" % (prec, D, e_s, h["loss"][0], a["loss"][0], rel_fro(h["dW"], a["dW"]), rel_fro(h["db"], a["db"])))
    assert e_s <= 2e-4 and abs(h["loss"][0] - a["loss"][0]) <= 2e-5 * a["loss"][0]
    assert rel_fro(h["dW"], a["dW"]) <= 2e-3 and rel_fro(h["db"], a["db"]) <= 2e-3
    h2 = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=0)      # bit-reproducible, as the fp32 form is
    assert np.array_equal(h2["dW"], h["dW"]) and np.array_equal(h2["db"], h["db"]) and h2["loss"] == h["loss"] and np.array_equal(h2["ip2"], h["ip2"])


@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("D,B,C,Nn", [(512, 256, 5, 50), (1024, 64, 5, 200)])
def test_split_k_partials_as_f16(vv, prec, D, B, C, Nn):
    ds, table, idx, W, b = make_case(29, 64, B, C, Nn, 1024, D, wstd=0.02)
    idx[:, C:] = idx[:, C:] % 900
    a = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=0)
    s = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=1)
    assert np.array_equal(s["ip2"], a["ip2"]) and s["loss"] == a["loss"] and np.array_equal(s["db"], a["db"])     # the forward pass and the bias path are untouched
    e = rel_fro(s["dW"], a["dW"])
    # a partial product moves by at most 2^-11 of its tile's largest magnitude (f16 below the tile's power of two); the sum of S of them by S times that
    tile_max = np.abs(a["dW"]).reshape(D // 256, 256, 1024 // 256, 256).max(axis=(1, 3))
    slack = np.repeat(np.repeat(tile_max, 256, axis=0), 256, axis=1)
    print("SLAB16 %s D %d: dW %.2e (max elementwise %.2e of the tile's max)" % (prec, D, e, (np.abs(s["dW"] - a["dW"]) / np.maximum(slack, 1e-30)).max()))
    assert e <= 1e-3
    assert np.isfinite(s["dW"]).all() and np.all(np.abs(s["dW"] - a["dW"]) <= 16 * 2.0 ** -11 * 8 * slack + 1e-12)
    s2 = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=1)
    assert np.array_equal(s2["dW"], s["dW"])


@pytest.mark.parametrize("prec", ["f16", "bf16"])
def test_per_item_vectors_as_f16_on_the_one_sweep_path(vv, prec):
    """Option v16 (D = 1024, the one-sweep score kernel: the per-GPU shape of BASELINE configs[4]): Ah_b and dA_b leave as f16 -- dA_b times a
    power of two whose inverse rides in the context instances' alpha -- and k_seg_bwd gathers half the bytes.  The forward pass is untouched;
    the gradient moves by one more 2^-12."""
    D, B, C, Nn = 1024, 64, 5, 200
    ds, table, idx, W, b = make_case(31, 64, B, C, Nn, 1024, D, wstd=0.02)
    idx[:, C:] = idx[:, C:] % 900
    a = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=0, v16=0)
    v = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=0, v16=1)
    assert np.array_equal(v["ip2"], a["ip2"]) and v["loss"] == a["loss"]
    print("V16 %s: dW %.2e db %.2e" % (prec, rel_fro(v["dW"], a["dW"]), rel_fro(v["db"], a["db"])))
    tol = 1e-3 if prec == "f16" else 4e-3          # (bf16: dYu is rounded to 8 bits behind the sum -- a 2^-12 change of an input flips roundings of 2^-9)
    assert rel_fro(v["dW"], a["dW"]) <= tol and rel_fro(v["db"], a["db"]) <= 1e-3
    v2 = run(vv, prec, table, idx, W, b, C, Nn, h16=1, slab16=0, v16=1)
    assert np.array_equal(v2["dW"], v["dW"]) and np.array_equal(v2["db"], v["db"])
