"""The WIRING of the training graph, pinned to the reference's own project file.

tests/golden/mednet_train_graph.json is the TRAIN-phase topology of /root/reference/projects/videovec_embedding/
mednet_embedding_train.prototxt (37 layers: names, types, bottoms, tops, layer parameters; made by tests/golden/make_graph_golden.py).
  1. It is EXECUTED here, layer by layer in file order and backwards in reverse order, by a generic interpreter that knows nothing of
     the videovec graph: every layer type maps to the oracle's function for that layer (each pinned to the reference's unit tests of that
     layer, tests/test_oracle_reference_kats.py), blobs consumed by several layers accumulate their diffs as Net::Init's SPLIT layers do
     (net.cpp:226-329 / split_layer.cpp:36-51).  The result must equal the oracle's hand-assembled step (oracle/vv_oracle.c:
     orc_forward_backward) -- loss, violations, scores, ip2, dW, db -- so the assembled oracle IS the reference's wiring of the pinned layers.
  2. The product's generator (videovector_amd/prototxt.py: what examples/*.prototxt and every facade test are made with) must produce
     exactly this topology for the shipped parameters: the graph the HIP path is matched against is the reference's graph.
"""
import json
import os

import numpy as np
import pytest

from prototxt_parse import parse_prototxt, train_topology

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "mednet_train_graph.json")))


class Graph:
    """Forward / backward of a list of layer records over named blobs (4-d shapes as Caffe's), in-place tops as new versions."""

    def __init__(self, layers, orc, data, W, b, mask, num_output):
        self.L, self.o, self.W, self.b, self.mask, self.D = layers, orc, W, b, mask, num_output
        self.val, self.cur, self.tape = {}, {}, []
        self.data = data

    def _get(self, name):
        return self.val[(name, self.cur[name])]

    def _put(self, name, arr):
        self.cur[name] = self.cur.get(name, -1) + 1
        self.val[(name, self.cur[name])] = np.ascontiguousarray(arr, np.float32)
        return (name, self.cur[name])

    def forward(self):
        o = self.o
        for l in self.L:
            t, p = l["type"], l["param"]
            ins = [(n, self.cur[n]) for n in l["bottom"]]
            x = [self.val[k] for k in ins]
            if t == "VIDEO_SAMPLED_SHOTS_DATA":
                outs = [self.data]
            elif t == "VIDEO_SHOT_WINDOW_TEST_DATA":
                outs = [self.data, np.zeros((self.data.shape[0], 1, 1, 1), np.float32)]      # data, video_ids
            elif t == "RETRIEVAL_STATS":
                outs = [np.zeros(1, np.float32)] * 3                  # (its own known-answer test: tests/test_oracle_reference_kats.py)
            elif t == "SLICE":
                dim, n = p["slice_dim"], len(l["top"])
                assert x[0].shape[dim] % n == 0                    # slice_layer.cpp: no slice_point -> equal pieces
                outs = o.slice_fwd(x[0], dim, [x[0].shape[dim] // n] * n)
            elif t == "CONCAT":
                outs = [o.concat_fwd(x, p["concat_dim"])]
            elif t == "FLATTEN":
                outs = [x[0].reshape(x[0].shape[0], -1, 1, 1)]
            elif t == "INNER_PRODUCT":
                X = x[0].reshape(x[0].shape[0], -1)
                outs = [o.inner_product_fwd(X, self.W, self.b).reshape(X.shape[0], self.D, 1, 1)]
            elif t == "RELU":
                outs = [o.relu_fwd(x[0])]
            elif t == "DROPOUT":
                outs = [o.dropout_fwd(x[0], self.mask.reshape(x[0].shape), p["dropout_ratio"], True)]
            elif t == "ELTWISE":
                outs = [o.eltwise_fwd(p["operation"], x, p["coeff"] or None)]
            elif t == "NORMALIZATION":
                outs = [o.normalize_fwd(x[0].reshape(x[0].shape[0], -1)).reshape(x[0].shape)]
            elif t == "SUM":
                k = int(p.get("sum_num_output", 1))
                outs = [o.sum_fwd(x[0].reshape(x[0].shape[0], -1), k).reshape(x[0].shape[0], k, 1, 1)]
            elif t == "MAX_MARGIN_LOSS":
                st, sb = x[0].reshape(x[0].shape[0], -1), x[1].reshape(x[1].shape[0], -1)
                self.loss, self.viol = o.max_margin_fwd(st, sb, p["margin"], {"L1": 1, "L2": 2}[p["norm"]])
                self.loss *= p["loss_weight"][0]
                outs = [np.float32([self.loss]), np.float32([self.viol])]
            else:
                raise AssertionError("layer type %s is not part of the training path" % t)
            assert len(outs) == len(l["top"]), l["name"]
            self.tape.append((l, ins, [self._put(n, a) for n, a in zip(l["top"], outs)]))

    def backward(self):
        o = self.o
        diff = {}

        def add(key, g):                       # several consumers of one blob: the SPLIT layer's sum (split_layer.cpp:36-51)
            g = np.ascontiguousarray(g, np.float32).reshape(self.val[key].shape)
            diff[key] = g if key not in diff else diff[key] + g
        self.dW = self.db = None
        for l, ins, outs in reversed(self.tape):
            t, p = l["type"], l["param"]
            x = [self.val[k] for k in ins]
            dy = [diff.get(k) for k in outs]
            if t == "MAX_MARGIN_LOSS":
                st, sb = x[0].reshape(x[0].shape[0], -1), x[1].reshape(x[1].shape[0], -1)
                dt, dbg = o.max_margin_bwd(st, sb, p["margin"], {"L1": 1, "L2": 2}[p["norm"]], p["loss_weight"][0])
                add(ins[0], dt); add(ins[1], dbg)
                continue
            if t == "VIDEO_SAMPLED_SHOTS_DATA" or all(d is None for d in dy):
                continue                       # a data layer has no bottoms (layer.hpp: nothing is propagated into it)
            if t == "SLICE":
                zs = [d if d is not None else np.zeros_like(self.val[k]) for d, k in zip(dy, outs)]
                add(ins[0], o.concat_fwd(zs, p["slice_dim"]))
            elif t == "CONCAT":
                dim = p["concat_dim"]
                pieces = o.slice_fwd(dy[0], dim, [a.shape[dim] for a in x])
                for k, g in zip(ins, pieces): add(k, g)
            elif t == "FLATTEN":
                add(ins[0], dy[0])
            elif t == "INNER_PRODUCT":
                X = x[0].reshape(x[0].shape[0], -1)
                self.dW, self.db, dX = o.inner_product_bwd(X, self.W, dy[0].reshape(X.shape[0], -1))
                add(ins[0], dX)
            elif t == "RELU":
                add(ins[0], o.relu_bwd(x[0], dy[0]))
            elif t == "DROPOUT":
                add(ins[0], o.dropout_bwd(dy[0], self.mask.reshape(dy[0].shape), p["dropout_ratio"], True))
            elif t == "ELTWISE":
                for j, k in enumerate(ins): add(k, o.eltwise_bwd(p["operation"], x, dy[0], j, p["coeff"] or None))
            elif t == "NORMALIZATION":
                add(ins[0], o.normalize_bwd(x[0].reshape(x[0].shape[0], -1), dy[0].reshape(x[0].shape[0], -1)))
            elif t == "SUM":
                add(ins[0], o.sum_bwd(dy[0].reshape(dy[0].shape[0], -1), int(np.prod(x[0].shape[1:]))))
            else:
                raise AssertionError(t)


def rel(a, r):
    return float(np.linalg.norm(a.astype(np.float64) - r) / max(np.linalg.norm(r), 1e-30))


@pytest.mark.parametrize("ratio_override", [None, 0.0])
def test_reference_topology_executed_with_pinned_layers_equals_the_assembled_oracle(oracle, ratio_override):
    layers = json.loads(json.dumps(GOLD["layers"]))
    data_p = layers[0]["param"]
    C, Nn = data_p["context_size"], data_p["num_negative_samples"]
    assert (C, Nn) == (5, 10) and len(layers) == 37
    B, F, D = 12, 40, 24                                          # the wiring is what is pinned; sizes are the test's
    rng = np.random.default_rng(7)
    table = np.abs(rng.standard_normal((300, F))).astype(np.float32)
    idx = rng.integers(0, 300, size=(B, C + Nn)).astype(np.int32)
    W = (0.1 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.05 * rng.standard_normal(D)).astype(np.float32)
    drop = [l for l in layers if l["type"] == "DROPOUT"][0]
    if ratio_override is not None:
        layers = [l for l in layers if l["type"] != "DROPOUT"] if ratio_override == 0.0 else layers
    ratio = 0.0 if ratio_override == 0.0 else drop["param"]["dropout_ratio"]
    assert drop["param"]["dropout_ratio"] == pytest.approx(0.9)
    mask = (rng.random(((C + Nn) * B, D)) >= ratio).astype(np.uint8)
    data = table[idx].reshape(B, C + Nn, 1, F)                     # the data layer's top (…data_layer.cpp:439-452): item-major, one channel per slot
    g = Graph(layers, oracle, data, W, b, mask, D)
    g.forward()
    g.backward()
    loss_l = [l for l in layers if l["type"] == "MAX_MARGIN_LOSS"][0]["param"]
    ref = oracle.forward_backward(table, idx, W, b, C_=C, Nn=Nn, margin=loss_l["margin"], norm={"L1": 1, "L2": 2}[loss_l["norm"]],
                                  loss_weight=loss_l["loss_weight"][0], ctx_coeff=[l for l in layers if l["name"] == "context_average"][0]["param"]["coeff"],
                                  dropout_ratio=ratio, dropout_mask=mask if ratio > 0 else None,
                                  want=("H", "s_true", "s_bogus", "dW", "db"))
    assert abs(g.loss - ref["loss"]) <= 1e-6 * abs(ref["loss"]) and g.viol == ref["violations"]
    assert np.allclose(g._get("target_score").reshape(B, Nn), ref["s_true"], rtol=0, atol=1e-6)
    assert np.allclose(g._get("negative_scores").reshape(B, Nn), ref["s_bogus"], rtol=0, atol=1e-6)
    assert np.allclose(g._get("ip2").reshape(-1, D), ref["H"], rtol=0, atol=1e-6)
    assert rel(g.dW, ref["dW"]) <= 2e-6 and rel(g.db, ref["db"]) <= 2e-6
    assert np.abs(ref["dW"]).max() > 0


def test_generated_prototxt_is_the_reference_graph():
    """videovector_amd/prototxt.py with the shipped parameters: layer for layer the reference's TRAIN topology (the data layer's `source`
    aside) -- names, types, bottoms, tops, slice / concat dims, eltwise operations and coefficients, fc7's size and multipliers, the SUM
    layer's replication, dropout ratio, margin, norm and loss weights."""
    from videovector_amd.prototxt import train_net
    d = GOLD["layers"][0]["param"]
    ip = [l for l in GOLD["layers"] if l["type"] == "INNER_PRODUCT"][0]["param"]
    drop = [l for l in GOLD["layers"] if l["type"] == "DROPOUT"][0]["param"]
    txt = train_net("synthetic://videos=50", d["batch_size"], d["context_size"], d["num_negative_samples"], ip["num_output"],
                    max_same=d["max_same_video_negs"], dropout=drop["dropout_ratio"], max_buffer=d["max_buffer_size"])
    mine = train_topology(parse_prototxt(txt))
    assert len(mine["layers"]) == len(GOLD["layers"]) == 37
    # the same graph up to the NAMES of intermediate blobs and layers (labels: the generator writes pos_neg / neg_prod_k where the shipped file
    # says pos_neg_norm / negative_emb_k_prod): layer for layer the same type and parameters, and ONE consistent renaming of the blobs
    ren, back = {}, {}
    for a, r in zip(mine["layers"], GOLD["layers"]):
        assert a["type"] == r["type"] and a["param"] == r["param"], (a["name"], r["name"], a["param"], r["param"])
        assert len(a["bottom"]) == len(r["bottom"]) and len(a["top"]) == len(r["top"]), r["name"]
        for x, y in zip(a["bottom"], r["bottom"]):
            assert ren.get(x) == y, (r["name"], x, y)             # a bottom is a blob an earlier layer produced: already mapped, to the same blob
        for x, y in zip(a["top"], r["top"]):
            if x in ren and x in a["bottom"]:                      # in-place layer (DROPOUT on ip2)
                assert ren[x] == y
                continue
            assert x not in ren and y not in back, (r["name"], x, y)
            ren[x] = y; back[y] = x
    # what a user of the net addresses by name keeps the reference's name: the parameters' layer, the loss tops, the data top, ip2
    for name in ("data", "ip1_nonorm", "ip2", "context_feature", "target_score", "negative_scores", "loss_output", "train_violations"):
        assert ren[name] == name
    assert [l["name"] for l in mine["layers"] if l["type"] == "INNER_PRODUCT"] == ["fc7"]


def test_reference_test_phase_topology_is_the_extraction_path(oracle):
    """The TEST phase of the same project file (10 layers): four context frames sliced, concatenated, flattened, sliced again and averaged
    (ELTWISE SUM, 0.25 each), fc7, ReLU, NORMALIZATION, retrieval statistics.  Executed with the pinned layer functions it must be the
    oracle's embedding of the averaged frames -- what vv_embed_mean / extract_features are tested against -- and the product's generator
    must produce the same ten layers."""
    layers = GOLD["test_layers"]
    assert [l["type"] for l in layers] == ["VIDEO_SHOT_WINDOW_TEST_DATA", "SLICE", "CONCAT", "FLATTEN", "SLICE", "ELTWISE", "INNER_PRODUCT", "RELU",
                                           "NORMALIZATION", "RETRIEVAL_STATS"]
    N, K, F, D = 9, 4, 40, 24
    rng = np.random.default_rng(11)
    table = np.abs(rng.standard_normal((100, F))).astype(np.float32)
    rows = rng.integers(0, 100, size=(N, K))
    W = (0.1 * rng.standard_normal((D, F))).astype(np.float32)
    b = (0.05 * rng.standard_normal(D)).astype(np.float32)
    g = Graph(layers, oracle, table[rows].reshape(N, K, 1, F), W, b, None, D)
    g.forward()
    mean = (table[rows].astype(np.float32) * np.float32(0.25)).sum(axis=1).astype(np.float32)
    ref = oracle.embed(mean, np.arange(N, dtype=np.int32), W, b, relu=True, l2norm=True)
    assert np.allclose(g._get("ip2_norm").reshape(N, D), ref, rtol=0, atol=2e-6)
    from videovector_amd.prototxt import train_net
    d = GOLD["layers"][0]["param"]
    txt = train_net("synthetic://videos=50", d["batch_size"], d["context_size"], d["num_negative_samples"], 4096,
                    max_same=d["max_same_video_negs"], dropout=0.9, max_buffer=d["max_buffer_size"],
                    test_source="synthetic://videos=50;windows=20", test_batch=673, test_frames=4, id_to_class_file="/tmp/none")
    mine = train_topology(parse_prototxt(txt), "TEST")["layers"]
    assert [(l["type"], len(l["bottom"]), len(l["top"])) for l in mine] == [(l["type"], len(l["bottom"]), len(l["top"])) for l in layers]
    for a, r in zip(mine, layers):
        pa = {k: v for k, v in a["param"].items() if k != "batch_size"}
        pr = {k: v for k, v in r["param"].items() if k != "batch_size"}
        assert pa == pr, (a["name"], pa, pr)
