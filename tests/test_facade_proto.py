"""CPU tests of the C++ facade's protobuf stand-in (caffe_facade/src/proto_lite.cpp): its binary wire
format is checked in both directions against the REAL protobuf runtime (google.protobuf, with the
reference's field numbers re-declared here through descriptor_pb2), its text parser against the
reference's own project files when /root/reference is present, and Net::FilterNet's phase rules."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "caffe_facade", "build", "proto_tool")
REF = "/root/reference/projects/videovec_embedding"


@pytest.fixture(scope="module")
def tool():
    subprocess.run(["make", "-C", os.path.join(ROOT, "videovector_amd", "csrc"), "-s", "-j4"], check=True)
    subprocess.run(["make", "-C", os.path.join(ROOT, "caffe_facade"), "-s", "-j4"], check=True)
    return TOOL


@pytest.fixture(scope="module")
def pb():
    """Message classes built from the reference's field numbers (src/caffe/proto/caffe.proto:5-15,
    51-66, 176-180, 215-389 subset) with the real protobuf runtime."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="caffe_subset.proto", package="caffe", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for (num, fname, ftype, label, tname, packed) in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = ".caffe." + tname
            if packed:
                f.options.packed = True
        return m
    O, R = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("BlobProto", [(1, "num", F.TYPE_INT32, O, "", 0), (2, "channels", F.TYPE_INT32, O, "", 0),
                      (3, "height", F.TYPE_INT32, O, "", 0), (4, "width", F.TYPE_INT32, O, "", 0),
                      (5, "data", F.TYPE_FLOAT, R, "", 1), (6, "diff", F.TYPE_FLOAT, R, "", 1)])
    msg("SolverState", [(1, "iter", F.TYPE_INT32, O, "", 0), (2, "learned_net", F.TYPE_STRING, O, "", 0),
                        (3, "history", F.TYPE_MESSAGE, R, "BlobProto", 0)])
    msg("InnerProductParameter", [(1, "num_output", F.TYPE_UINT32, O, "", 0), (2, "bias_term", F.TYPE_BOOL, O, "", 0),
                                  (5, "regularization", F.TYPE_DOUBLE, O, "", 0)])
    e = fd.enum_type.add(name="LayerType")
    for n, v in (("NONE", 0), ("INNER_PRODUCT", 14), ("RELU", 18), ("MAX_MARGIN_LOSS", 43),
                 ("VIDEO_SAMPLED_SHOTS_DATA", 49)):
        e.value.add(name=n, number=v)
    msg("LayerParameter", [(2, "bottom", F.TYPE_STRING, R, "", 0), (3, "top", F.TYPE_STRING, R, "", 0),
                           (4, "name", F.TYPE_STRING, O, "", 0), (5, "type", F.TYPE_ENUM, O, "LayerType", 0),
                           (6, "blobs", F.TYPE_MESSAGE, R, "BlobProto", 0), (7, "blobs_lr", F.TYPE_FLOAT, R, "", 0),
                           (8, "weight_decay", F.TYPE_FLOAT, R, "", 0), (35, "loss_weight", F.TYPE_FLOAT, R, "", 0),
                           (17, "inner_product_param", F.TYPE_MESSAGE, O, "InnerProductParameter", 0)])
    msg("NetParameter", [(1, "name", F.TYPE_STRING, O, "", 0), (2, "layers", F.TYPE_MESSAGE, R, "LayerParameter", 0)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = lambda n: message_factory.GetMessageClass(pool.FindMessageTypeByName("caffe." + n))
    return {n: get(n) for n in ("BlobProto", "SolverState", "LayerParameter", "NetParameter")}


def test_wire_real_protobuf_to_proto_lite(tool, pb, tmp_path):
    st = pb["SolverState"](iter=2000, learned_net="x/y_iter_2000.caffemodel")
    rng = np.random.default_rng(0)
    w = rng.standard_normal(12).astype(np.float32)
    h = st.history.add(num=1, channels=1, height=3, width=4)
    h.data.extend(w.tolist())
    st.history.add(num=1, channels=1, height=1, width=3).data.extend([0.5, -1.25, 3e-9])
    (tmp_path / "s.bin").write_bytes(st.SerializeToString())
    subprocess.run([tool, "bin2text", "SolverState", str(tmp_path / "s.bin"), str(tmp_path / "s.txt")], check=True)
    subprocess.run([tool, "text2bin", "SolverState", str(tmp_path / "s.txt"), str(tmp_path / "s2.bin")], check=True)
    back = pb["SolverState"]()
    back.ParseFromString((tmp_path / "s2.bin").read_bytes())
    assert back == st                                  # text round trip preserves every float bit
    assert (tmp_path / "s2.bin").read_bytes() == st.SerializeToString()   # and the bytes (packed data)


def test_wire_proto_lite_to_real_protobuf(tool, pb, tmp_path):
    (tmp_path / "n.txt").write_text('''
name: "n"   # a comment
layers { name: "fc7" type: INNER_PRODUCT bottom: "a" top: "b" blobs_lr: 1 blobs_lr: 2
         weight_decay: 1 weight_decay: 0
         inner_product_param { num_output: 4096 regularization: 0.25 }
         blobs { num: 1 channels: 1 height: 2 width: 2 data: 1 data: -2.5 data: 1e-3 data: 4 } }
layers { name: "max_margin_loss" type: MAX_MARGIN_LOSS bottom: 'b' top: "l" loss_weight: 1.0 loss_weight: 0 }
''')
    subprocess.run([tool, "text2bin", "NetParameter", str(tmp_path / "n.txt"), str(tmp_path / "n.bin")], check=True)
    net = pb["NetParameter"]()
    net.ParseFromString((tmp_path / "n.bin").read_bytes())
    assert net.name == "n" and len(net.layers) == 2
    fc = net.layers[0]
    assert fc.name == "fc7" and fc.type == 14 and list(fc.blobs_lr) == [1, 2] and list(fc.weight_decay) == [1, 0]
    assert fc.inner_product_param.num_output == 4096 and fc.inner_product_param.regularization == 0.25
    assert np.allclose(list(fc.blobs[0].data), [1, -2.5, 1e-3, 4]) and fc.blobs[0].height == 2
    assert net.layers[1].type == 43 and list(net.layers[1].loss_weight) == [1.0, 0.0]
    assert list(net.layers[1].bottom) == ["b"]


def test_unknown_wire_fields_are_preserved(tool, pb, tmp_path):
    # a field this build does not model (LayerParameter.convolution_param = 10) must survive a round trip
    raw = pb["LayerParameter"](name="c").SerializeToString() + bytes([0x52, 0x02, 0x08, 0x07])
    net = bytes([0x12, len(raw)]) + raw
    (tmp_path / "u.bin").write_bytes(net)
    subprocess.run([tool, "bin2text", "NetParameter", str(tmp_path / "u.bin"), str(tmp_path / "u.txt")], check=True)
    # binary -> binary through the in-memory form
    subprocess.run([tool, "text2bin", "NetParameter", str(tmp_path / "u.txt"), str(tmp_path / "u2.bin")], check=True)
    assert b"c" in (tmp_path / "u2.bin").read_bytes()


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")
def test_reference_project_files_parse_and_filter(tool, tmp_path):
    # the shipped net: 37 layers in TRAIN phase, 10 in TEST
    for phase, n in (("TRAIN", 37), ("TEST", 10)):
        out = tmp_path / ("%s.prototxt" % phase)
        subprocess.run([tool, "filter", os.path.join(REF, "mednet_embedding_train.prototxt"), phase, str(out)],
                       check=True, capture_output=True)
        txt = out.read_text()
        assert txt.count("\nlayers {") + txt.startswith("layers {") == n
    txt = (tmp_path / "TRAIN.prototxt").read_text()
    for frag in ("batch_size: 128", "num_negative_samples: 10", "max_same_video_negs: 6", "context_type: WINDOW",
                 "dropout_ratio: 0.9", "margin: 2", "norm: L2", "coeff: 0.25", "num_output: 4096"):
        assert frag in txt, frag
    subprocess.run([tool, "text2bin", "SolverParameter", os.path.join(REF, "mednet_embedding_train_solver.prototxt"),
                    str(tmp_path / "s.bin")], check=True)
    subprocess.run([tool, "bin2text", "SolverParameter", str(tmp_path / "s.bin"), str(tmp_path / "s.txt")], check=True)
    s = (tmp_path / "s.txt").read_text()
    for frag in ("base_lr: 0.001", 'lr_policy: "inv"', "gamma: 0.001", "power: 0.75", "momentum: 0.9",
                 "weight_decay: 0.0005", "snapshot: 2000", "solver_mode: GPU", "max_iter: 200000"):
        assert frag in s, frag


def test_generated_prototxt_has_the_shipped_structure(tool, tmp_path):
    from videovector_amd.prototxt import train_net
    # same graph as the shipped file when generated with its parameters: 37 TRAIN layers
    p = tmp_path / "g.prototxt"
    p.write_text(train_net("synthetic://videos=50", 128, 5, 10, 4096, max_same=6, dropout=0.9))
    subprocess.run([tool, "filter", str(p), "TRAIN", str(tmp_path / "f.prototxt")], check=True, capture_output=True)
    txt = (tmp_path / "f.prototxt").read_text()
    assert txt.count("\nlayers {") + txt.startswith("layers {") == 37
    subprocess.run([tool, "filter", str(p), "TEST", str(tmp_path / "t.prototxt")], check=True, capture_output=True)
    t = (tmp_path / "t.prototxt").read_text()
    assert "fc7" in t and "VIDEO_SAMPLED_SHOTS_DATA" not in t
