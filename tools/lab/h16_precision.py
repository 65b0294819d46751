"""What a 16-bit `H` (ip2 stored as f16 with a power-of-two scale instead of fp32) would cost in precision -- measured, VERDICT r3
item 4.  The engine's own fp32 `ip2` blob is rounded to f16 on the host (max |h| placed in [2^11, 2^12): exact scaling) and the rest
of the forward graph -- context mean, the two normalisations, the 1 + Nn dot products, the hinge loss -- is evaluated from it with
the oracle's layer functions, next to the same evaluation of the unrounded blob.  Both are compared with the fp32 oracle on fp32
operands.  Shapes: BASELINE configs[0] (plumbing), a shard of configs[1], a shard of configs[4].

Uses oracle/ as the checker (test infrastructure): this is a lab script, not product code.  Run on a GPU box: python tools/lab/h16_precision.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import videovector_amd as vv                                  # noqa: E402
from videovector_amd.synth import SyntheticVideos, init_weights  # noqa: E402
from oracle import oracle                                     # noqa: E402


def graph_from_H(H, B, C, Nn, margin=2.0, norm=2):
    """scores and loss from the ip2 blob (row ch*B + b), with the oracle's layer functions (fp32, as the reference computes them)"""
    D = H.shape[1]
    E = H.reshape(C + Nn, B, D)
    A = sum(E[j] for j in range(1, C)) * np.float32(1.0 / (C - 1))
    Ahat = oracle.normalize_fwd(A.astype(np.float32))
    PN = np.concatenate([E[0]] + [E[C + k] for k in range(Nn)], axis=0)
    Phat = oracle.normalize_fwd(PN).reshape(1 + Nn, B, D)
    s = np.einsum("bd,qbd->bq", Ahat.astype(np.float64), Phat.astype(np.float64)).astype(np.float32)
    s_true = np.repeat(s[:, :1], Nn, axis=1)
    s_bogus = np.ascontiguousarray(s[:, 1:])
    loss, viol = oracle.max_margin_fwd(s_true, s_bogus, margin, norm)
    return s_true, s_bogus, loss, viol


def round_f16(H):
    m = float(np.abs(H).max())
    e = np.frexp(m)[1]                       # m = f 2^e, f in [0.5, 1)
    sc = np.float32(2.0 ** (12 - e))
    return (H * sc).astype(np.float16).astype(np.float32) / sc, sc


def case(name, ds, idx_full, sl, B_glob, C, Nn, F, D, prec="f16"):
    sh = idx_full[sl]
    Bs = sh.shape[0]
    W, b = init_weights(5, D, F)
    uniq, inv = np.unique(sh.reshape(-1), return_inverse=True)
    table = ds.table(F, uniq)
    il = inv.reshape(sh.shape).astype(np.int32)
    ref = oracle.forward_backward(table, il, W, b, C_=C, Nn=Nn, global_count=B_glob * Nn, want=("H", "s_true", "s_bogus"))
    eng = vv.Engine(0, prec)
    eng.table_synth(ds.seed, ds.n_rows, F)
    eng.params_set(W, b)
    cfg = vv.StepConfig(Bs, C, Nn, global_count=B_glob * Nn)
    eng.forward_backward(cfg, sh)
    got = eng.blobs(cfg)
    H32 = got["ip2"]
    H16, sc = round_f16(H32)
    nref = np.maximum(np.linalg.norm(ref["H"], axis=1), 1e-30)
    out = []
    for tag, H in (("fp32 H (today)", H32), ("f16 H", H16)):
        e_emb = (np.linalg.norm(H - ref["H"], axis=1) / nref).max()
        e_rms = np.sqrt(np.mean((np.linalg.norm(H - ref["H"], axis=1) / nref) ** 2))
        st, sb, loss, viol = graph_from_H(H, Bs, C, Nn)
        e_sc = max(np.abs(sb - ref["s_bogus"]).max(), np.abs(st - ref["s_true"]).max())
        out.append((tag, e_emb, e_rms, e_sc, abs(loss - ref["loss"]) / ref["loss"], viol - ref["violations"]))
    # the host evaluation itself against the engine's own scores (it must be the same function)
    st, sb, loss, viol = graph_from_H(H32, Bs, C, Nn)
    chk = max(np.abs(sb - got["negative_scores"]).max(), np.abs(st[:, 0] - got["target_score"][:, 0]).max())
    print("%s  (%s operands, %d items, %d rows, D %d; scale of the f16 copy 2^%d; host graph vs engine scores %.1e)" %
          (name, prec, Bs, sh.size, D, int(np.log2(sc)), chk))
    for tag, e_emb, e_rms, e_sc, e_loss, dv in out:
        print("    %-15s embeddings max %.3e rms %.3e | scores max abs %.3e | loss rel %.3e | violations %+d" % (tag, e_emb, e_rms, e_sc, e_loss, int(dv)))
    del eng
    return out


def main():
    ds = SyntheticVideos(seed=1701, n_videos=2048)
    # configs[1]: batch 1024, C 5, Nn 50, 4096 -> 512
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=1024, context_size=5, num_negative_samples=50,
                     max_buffer_size=5000, negative_swap_percentage=50)
    smp.next(); idx2 = smp.next(); smp.close()
    case("cfg 2 shard", ds, idx2, slice(100, 164), 1024, 5, 50, 4096, 512)
    case("cfg 2 shard", ds, idx2, slice(100, 164), 1024, 5, 50, 4096, 512, prec="bf16")
    # configs[4]: batch 4096, Nn 200, 4096 -> 1024
    smp = vv.Sampler(ds.video_id, ds.n_shots, ds.row_base, batch_size=4096, context_size=5, num_negative_samples=200,
                     max_buffer_size=5000, negative_swap_percentage=50)
    idx5 = smp.next(); smp.close()
    case("cfg 5 shard", ds, idx5, slice(1000, 1032), 4096, 5, 200, 4096, 1024)
    # configs[0]: 1k frames, 128 -> 32, batch 32, 2 negatives
    ds1 = SyntheticVideos(seed=7, n_videos=40)
    smp = vv.Sampler(ds1.video_id, ds1.n_shots, ds1.row_base, batch_size=32, context_size=5, num_negative_samples=2,
                     max_buffer_size=200, negative_swap_percentage=50)
    idx1 = smp.next(); smp.close()
    case("cfg 1", ds1, idx1, slice(0, 32), 32, 5, 2, 128, 32)


if __name__ == "__main__":
    main()
