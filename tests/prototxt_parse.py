"""A small reader of protobuf TEXT format (nested `name { ... }` / `name: value`, `#` comments) and the extraction of a net's
TRAIN-phase topology from it -- test infrastructure shared by tests/golden/make_graph_golden.py (the reference's project file) and
tests/test_oracle_graph_topology.py (the product generator's output).  Repeated fields become lists."""
import re

_TOK = re.compile(r'\s*(?:(#[^\n]*)|("(?:[^"\\]|\\.)*")|([{}:])|([^\s{}:#"]+))')


def _tokens(text):
    pos = 0
    while pos < len(text):
        m = _TOK.match(text, pos)
        if not m:
            if text[pos:].strip() == "":
                return
            raise ValueError("cannot tokenise at %d: %r" % (pos, text[pos:pos + 30]))
        pos = m.end()
        if m.group(1):
            continue
        yield m.group(2) or m.group(3) or m.group(4)


def _value(tok):
    if tok.startswith('"'):
        return tok[1:-1]
    try:
        return int(tok)
    except ValueError:
        try:
            return float(tok)
        except ValueError:
            return tok          # an enum name / true / false


def _block(it):
    out = {}
    for tok in it:
        if tok == "}":
            return out
        name = tok
        nxt = next(it)
        if nxt == ":":
            nxt = next(it)
            val = _block(it) if nxt == "{" else _value(nxt)
        elif nxt == "{":
            val = _block(it)
        else:
            raise ValueError("expected ':' or '{' after %s" % name)
        out.setdefault(name, []).append(val)
    return out


def parse_prototxt(text):
    return _block(iter(list(_tokens(text)) + ["}"]))


def _one(d, k, default=None):
    return d[k][0] if k in d else default


def in_phase(layer, phase):
    """net.cpp:226-329 FilterNet with a NetState of `phase` only: include rules select, a layer without rules is in every phase"""
    inc = layer.get("include", [])
    if not inc:
        return not any(_one(r, "phase") == phase for r in layer.get("exclude", []))
    return any(_one(r, "phase", phase) == phase for r in inc)


def train_topology(net, phase="TRAIN"):
    layers = []
    for l in net["layers"]:
        if not in_phase(l, phase):
            continue
        e = {"name": _one(l, "name"), "type": _one(l, "type"), "bottom": l.get("bottom", []), "top": l.get("top", [])}
        p = {}
        # (caffe.proto defaults where the block or the field is absent: SliceParameter.slice_dim 1, ConcatParameter.concat_dim 1, SumParameter.num_output 1)
        if e["type"] == "SLICE": p["slice_dim"] = _one(l["slice_param"][0], "slice_dim", 1) if "slice_param" in l else 1
        if e["type"] == "CONCAT": p["concat_dim"] = _one(l["concat_param"][0], "concat_dim", 1) if "concat_param" in l else 1
        if "eltwise_param" in l:
            p["operation"] = _one(l["eltwise_param"][0], "operation", "SUM")
            p["coeff"] = [float(x) for x in l["eltwise_param"][0].get("coeff", [])]
        if "inner_product_param" in l: p["num_output"] = _one(l["inner_product_param"][0], "num_output")
        if e["type"] == "SUM": p["sum_num_output"] = _one(l["sum_param"][0], "num_output", 1) if "sum_param" in l else 1
        if "dropout_param" in l: p["dropout_ratio"] = float(_one(l["dropout_param"][0], "dropout_ratio", 0.5))
        if "max_margin_loss_param" in l:
            p["norm"] = _one(l["max_margin_loss_param"][0], "norm", "L2"); p["margin"] = float(_one(l["max_margin_loss_param"][0], "margin", 1.0))
        if "loss_weight" in l: p["loss_weight"] = [float(x) for x in l["loss_weight"]]
        if "blobs_lr" in l: p["blobs_lr"] = [float(x) for x in l["blobs_lr"]]
        if "weight_decay" in l: p["weight_decay"] = [float(x) for x in l["weight_decay"]]
        if "video_shot_window_test_data_param" in l:
            p["batch_size"] = _one(l["video_shot_window_test_data_param"][0], "batch_size")
        if "video_sampled_shots_data_param" in l:
            d = l["video_sampled_shots_data_param"][0]
            for k in ("batch_size", "num_negative_samples", "max_buffer_size", "negative_swap_percentage", "max_same_video_negs", "context_type", "context_size"):
                if k in d: p[k] = _one(d, k)
        e["param"] = p
        layers.append(e)
    return {"name": _one(net, "name"), "layers": layers}
