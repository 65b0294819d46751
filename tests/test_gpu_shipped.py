"""The reference's SHIPPED training configuration (projects/videovec_embedding/mednet_embedding_train.prototxt:13-23, 200,
226: window 5, 10 negatives of which up to 6 come from the item's own video -- quirk Q1: copied without their last
feature --, fc7 4096 -> 4096, dropout 0.9) as a shard of 32 items against the oracle: explicit dropout mask (the
reference's masks come from boost / curand and are not reproducible across devices even there), indices
and last-feature sources from the bit-exact sampler.  The shapes the shipped graph runs: no row de-duplication (every
instance has its own mask), D = 4096 through the generic score kernel, the guard on the dense 16-bit gradient rows."""
import numpy as np
import pytest

from tests.test_gpu_parity import TOL, rel_fro, rel_rows, round_operand, round_table
from videovector_amd.synth import SyntheticVideos, init_weights

pytestmark = pytest.mark.gpu


def test_shipped_shape_shard_matches_the_oracle(oracle):
    import videovector_amd as vv
    B, C, Nn, F, D, ratio = 32, 5, 10, 4096, 4096, 0.9
    ds = SyntheticVideos(seed=41, n_videos=120)
    kw = dict(batch_size=B, context_size=C, num_negative_samples=Nn, max_buffer_size=1500, negative_swap_percentage=50,
              max_same_video_negs=6)
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    osm = oracle.Sampler(ds.video_id, ds.n_shots, ds.row_base, **kw)
    for _ in range(3):                                   # the third batch: its Q1 slots inherit last features from the second
        idx, last, _ = smp.next(want_last=True)
        o = osm.next()
    assert np.array_equal(idx, o[0]) and np.array_equal(last, o[1])          # triplet indices bit-exact
    assert (idx != last).any(), "no quirk-Q1 slot in this batch"
    W, b = init_weights(41, D, F, std=0.003)
    b = (np.random.default_rng(3).standard_normal(D) * 0.003).astype(np.float32)
    mask = (np.random.default_rng(2).random(((C + Nn) * B, D)) >= ratio).astype(np.uint8)     # keep with probability 1 - ratio
    eng = vv.Engine(0, "f16")
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    cfg = vv.StepConfig(B, C, Nn, dropout_ratio=ratio, dropout_mask=mask)
    eng.forward_backward_q1(cfg, idx, last)
    loss, viol = eng.loss()
    dW, db = eng.grads()
    rows, uniq = eng.dedup_stats()
    assert rows == uniq == B * (C + Nn)                  # dropout: the dense path
    assert eng.grad_scale_stats()[0] == 0
    # the oracle on the rows this batch names (table rows regenerated on the host from the same counter-based generator)
    used = np.unique(np.concatenate([idx.reshape(-1), last.reshape(-1)]))
    used = used[used >= 0]
    table = ds.table(F, used)
    remap = {int(r): i for i, r in enumerate(used)}
    f = np.vectorize(lambda r: remap[int(r)] if r >= 0 else -1)
    idx_l, last_l = f(idx).astype(np.int32), f(last).astype(np.int32)
    okw = dict(C_=C, Nn=Nn, dropout_ratio=ratio, dropout_mask=mask, last_src=last_l, want=("H", "s_true", "s_bogus", "dW", "db"))
    ref = oracle.forward_backward(table, idx_l, W, b, **okw)
    q = oracle.forward_backward(round_table(table, "f16"), idx_l, round_operand(W, "f16"), b, **dict(okw, want=("dW", "db")))
    blobs = eng.blobs(cfg, ip2=True, scores=True)
    m = dict(emb=rel_rows(blobs["ip2"], ref["H"]), loss=abs(loss - ref["loss"]) / abs(ref["loss"]),
             score=max(np.abs(blobs["target_score"] - ref["s_true"]).max(), np.abs(blobs["negative_scores"] - ref["s_bogus"]).max()),
             dw_q=rel_fro(dW, q["dW"]), db_q=rel_fro(db, q["db"]), dw=rel_fro(dW, ref["dW"]), db=rel_fro(db, ref["db"]))
    print("SHIPPED shard %s; violations %d vs %d" % (" ".join("%s=%.3e" % kv for kv in m.items()), viol, ref["violations"]))
    t = TOL["f16"]
    assert m["emb"] <= t["emb"] and m["loss"] <= t["loss"] and m["score"] <= t["score"]
    assert m["dw_q"] <= t["grad_q"] and m["db_q"] <= t["grad_q"] and m["dw"] <= t["grad"] and m["db"] <= t["grad"]
    assert viol == ref["violations"]
    smp.close(); eng.close()
