"""Independent small-case restatements used to cross-check the C oracle (pure Python / numpy
float64; tests only).

  PySampler   -- the WINDOW-mode sampler written a second time, in pure Python, driven by the REAL
                 glibc rand() (ctypes) instead of the oracle's clone.
                 Follows src/caffe/layers/video_sampled_shots_data_layer.cpp:24-44,240-344,
                 371-393,425-507,768-909 and include/caffe/util/rng.hpp:43-54.
  fused_step  -- the training step in closed form in float64, an independent
                 formulation of what the oracle computes layer by layer.
"""
import ctypes

import numpy as np

_libc = ctypes.CDLL("libc.so.6")


class PySampler:
    def __init__(self, video_id, n_shots, row_base, B, C, Nn, max_buffer, swap, max_same=0,
                 max_tries=100, context_type="WINDOW", initial_cursor=0, output_shot_distance=False,
                 max_shot_distance=5.0, negatives=None):
        _libc.srand(1)            # identical to never having called srand
        self.context_type = context_type
        self.out_dist, self.max_dist = output_shot_distance, max_shot_distance
        if context_type == "PAIRWISE":
            C = 2                  # …data_layer.cpp:200-201
        self.calls = 0
        self.vid, self.ns, self.rb = list(video_id), list(n_shots), list(row_base)
        self.B, self.C, self.Nn, self.mb, self.swap, self.max_same = B, C, Nn, max_buffer, swap, max_same
        self.cursor = initial_cursor % len(self.vid)
        self.buffer_ids = list(range(max_buffer if Nn > 0 else 0))
        self.buf_row, self.buf_key, self.keys = [], [], set()
        CN = C + Nn
        self.slot_row = [[-1] * CN for _ in range(B)]
        self.slot_last = [[-1] * CN for _ in range(B)]
        if Nn > 0 and negatives is not None:
            # …data_layer.cpp:253-286, 325-341: every new shot of the negative dataset's records, in order
            nvid, nns, nrb = negatives
            cur = 0
            for _ in range(max_tries * max_buffer):
                v, cur = cur, (cur + 1) % len(nvid)
                for j in range(nns[v]):
                    key = (nvid[v], j)
                    if key not in self.keys:
                        self.keys.add(key)
                        self.buf_key.append(key)
                        self.buf_row.append(nrb[v] + j)
                if len(self.buf_row) >= max_buffer:
                    break
            assert len(self.buf_row) == max_buffer
        elif Nn > 0:
            for _ in range(max_tries * max_buffer):
                v = self.cursor
                self.cursor = (self.cursor + 1) % len(self.vid)
                j = self.rand() % self.ns[v]
                key = (self.vid[v], j)
                if key not in self.keys:
                    self.keys.add(key)
                    self.buf_key.append(key)
                    self.buf_row.append(self.rb[v] + j)
                if len(self.buf_row) >= max_buffer:
                    break
            assert len(self.buf_row) == max_buffer

    def rand(self):
        self.calls += 1
        return _libc.rand()

    def random_unique(self, a, lo, n):
        left, first = len(a) - lo, lo
        for _ in range(n):
            r = first + self.rand() % left
            a[first], a[r] = a[r], a[first]
            first += 1
            left -= 1

    def next(self):
        B, C, Nn = self.B, self.C, self.Nn
        item, labels = 0, [0] * B
        while item < B:
            v = self.cursor
            n = self.ns[v]
            added, ok = 0, False
            lab = self.vid[v]
            if self.context_type == "PAIRWISE":
                if n >= 2:           # …data_layer.cpp:387, 396-422
                    perm = list(range(n))
                    self.random_unique(perm, 0, 2)
                    for c in (0, 1):
                        self.slot_row[item][c] = self.slot_last[item][c] = self.rb[v] + perm[c]
                    if self.out_dist:
                        d = abs(perm[0] - perm[1])
                        lab = int(self.max_dist) if d >= self.max_dist else d
                    ok = True
            elif self.context_type != "WINDOW" and n >= 2 and n >= C:
                # …data_layer.cpp:510-757: target = the last of the C frames, context = the C-1 before it
                perm = list(range(n))
                if self.context_type == "PAST":
                    self.random_unique(perm, 0, C)
                    perm[:C] = sorted(perm[:C])
                    frames = perm[:C]
                else:
                    msl = (n - C) // (C - 1)
                    if self.context_type == "PAST_CONTINUOUS":
                        sl = self.rand() % (msl + 1)
                        begin = self.rand() % (n - (C - 1) * sl - C + 1)
                    else:
                        sl = msl - 1 if msl >= 1 else 0
                        begin = n - (C - 1) * sl - C
                    frames = [begin + i * (sl + 1) for i in range(C)]
                for i, fr in enumerate(frames):
                    c = 0 if i == C - 1 else i + 1
                    self.slot_row[item][c] = self.slot_last[item][c] = self.rb[v] + fr
                ok = True
                if self.context_type == "PAST":
                    if Nn > 0 and n > C:
                        for i in range(C + 1, n):
                            j = C + self.rand() % (i - C + 1)
                            if i != j:
                                perm[i], perm[j] = perm[j], perm[i]
                        nid = C
                        while nid < n and added < self.max_same:
                            if perm[nid] < perm[1]:
                                self.slot_row[item][C + added] = self.rb[v] + perm[nid]
                                added += 1
                            nid += 1
                elif Nn > 0 and begin > 0:
                    nid = begin - 1
                    while nid >= 0 and added < self.max_same:
                        self.slot_row[item][C + added] = self.rb[v] + nid
                        added += 1
                        nid -= 1
            elif n >= 2 and n >= C:
                perm = list(range(n))
                self.random_unique(perm, 0, C)
                perm[:C] = sorted(perm[:C])
                half, ctx = C // 2, 0
                for i in range(C):
                    r = self.rb[v] + perm[i]
                    c = 0 if i == half else ctx + 1
                    if i != half:
                        ctx += 1
                    self.slot_row[item][c] = r
                    self.slot_last[item][c] = r
                ok = True
                if Nn > 0 and n > C:
                    for i in range(C + 1, n):        # std::random_shuffle on perm[C:]
                        j = C + self.rand() % (i - C + 1)
                        if i != j:
                            perm[i], perm[j] = perm[j], perm[i]
                    nid = C
                    while nid < n and added < self.max_same:
                        if perm[nid] < perm[half - 1] or perm[nid] > perm[half + 1]:
                            self.slot_row[item][C + added] = self.rb[v] + perm[nid]   # F-1 copy
                            added += 1
                        nid += 1
            self.cursor = (self.cursor + 1) % len(self.vid)
            if not ok:
                continue
            if Nn > 0:
                self.random_unique(self.buffer_ids, 0, Nn - added)
                for c in range(C + added, C + Nn):
                    r = self.buf_row[self.buffer_ids[c - C - added]]
                    self.slot_row[item][c] = r
                    self.slot_last[item][c] = r
            labels[item] = lab
            item += 1
            if Nn > 0 and self.swap > 0:
                for j in range(n):
                    key = (self.vid[v], j)
                    if key in self.keys:
                        continue
                    if self.rand() % 100 < self.swap:
                        pos = self.rand() % self.mb
                        self.keys.discard(self.buf_key[pos])
                        self.keys.add(key)
                        self.buf_key[pos] = key
                        self.buf_row[pos] = self.rb[v] + j
        return (np.array(self.slot_row, np.int32), np.array(self.slot_last, np.int32),
                np.array(labels, np.int32))


def fused_step(table, idx, W, b, C, Nn, margin=2.0, norm=2, loss_weight=1.0, coeff=None,
               mask=None, drop=0.0, global_count=0):
    """SURVEY App. A steps 1-11 in float64.  idx [B][C+Nn]; mask in channel-major row order."""
    table, W = table.astype(np.float64), W.astype(np.float64)
    B, CN = idx.shape
    D = W.shape[0]
    coeff = np.full(C - 1, 1.0 / (C - 1)) if coeff is None else np.asarray(coeff, np.float64)
    X = table[idx.T.reshape(-1)]                     # row = ch*B + b
    Y = X @ W.T + (0 if b is None else b.astype(np.float64))
    H = np.maximum(Y, 0)
    scale = 1.0 / (1.0 - drop) if drop > 0 else 1.0
    if drop > 0:
        H = H * mask * scale
    E = H.reshape(CN, B, D)
    A = np.tensordot(coeff, E[1:C], axes=(0, 0))
    nA = np.sqrt((A * A).sum(1, keepdims=True))
    Ah = A / (nA + 1e-10)
    P = np.concatenate([E[0:1], E[C:]], 0)           # [1+Nn][B][D]
    nP = np.sqrt((P * P).sum(2, keepdims=True))
    Ph = P / (nP + 1e-10)
    s = (Ah[None] * Ph).sum(2)                       # [1+Nn][B]
    sp, sn = s[0], s[1:].T                           # [B], [B][Nn]
    d = sp[:, None] - sn
    h = np.maximum(0, margin - d)
    n = B * Nn
    loss = loss_weight * ((h * h).sum() if norm == 2 else np.abs(h).sum()) / n
    viol = float((d < 0).sum())
    ng = global_count if global_count else n
    g = loss_weight * (2 * h if norm == 2 else (h > 0).astype(np.float64)) / ng
    dsp = -g.sum(1)
    c = np.concatenate([dsp[None], g.T], 0)          # [1+Nn][B]
    dAh = (c[:, :, None] * Ph).sum(0)
    dPh = c[:, :, None] * Ah[None]

    def nbwd(x, u):
        ss = (x * x).sum(-1, keepdims=True)
        return (ss * u - x * (x * u).sum(-1, keepdims=True)) / (ss ** 1.5 + 1e-10)

    dP, dA = nbwd(P, dPh), nbwd(A, dAh)
    dE = np.zeros_like(E)
    dE[0], dE[C:] = dP[0], dP[1:]
    for j in range(1, C):
        dE[j] = coeff[j - 1] * dA
    dH = dE.reshape(CN * B, D)
    if drop > 0:
        dH = dH * mask * scale
    dY = dH * (Y > 0)
    return dict(loss=loss, violations=viol, s_true=np.repeat(sp[:, None], Nn, 1), s_bogus=sn,
                Y=Y, H=H, ctx=Ah, posneg=Ph.reshape(-1, D), dY=dY, dW=dY.T @ X, db=dY.sum(0))
