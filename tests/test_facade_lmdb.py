"""LMDB `source:` databases: the facade's read-only LMDB walker and its VideoShots /
TestVideoShotWindows record decoders.  Records are serialised by the real google.protobuf runtime (field
numbers of src/caffe/proto/video_shot_sentences.proto and caffe.Datum re-declared through descriptor_pb2);
the database file comes from tests/lmdb_writer.py -- see its docstring for what that does and does not prove."""
import os
import subprocess

import numpy as np
import pytest

from lmdb_writer import write_lmdb

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "caffe_facade", "build", "proto_tool")


@pytest.fixture(scope="module")
def tool():
    if not os.path.exists(TOOL):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "caffe_facade")])
    return TOOL


@pytest.fixture(scope="module")
def pb():
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name="vss.proto", package="vss", syntax="proto2")

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for num, fname, typ, label, tname in fields:
            f = m.field.add(name=fname, number=num, type=typ, label=label)
            if tname:
                f.type_name = tname
    O, R = F.LABEL_OPTIONAL, F.LABEL_REPEATED
    msg("Datum", [(1, "channels", F.TYPE_INT32, O, None), (2, "height", F.TYPE_INT32, O, None), (3, "width", F.TYPE_INT32, O, None),
                  (4, "data", F.TYPE_BYTES, O, None), (5, "label", F.TYPE_INT32, O, None), (6, "float_data", F.TYPE_FLOAT, R, None)])
    msg("VideoShots", [(1, "video_id", F.TYPE_INT32, O, None), (2, "shot_ids", F.TYPE_INT32, R, None),
                       (3, "shot_words", F.TYPE_MESSAGE, R, ".vss.Datum"), (4, "video_name", F.TYPE_STRING, O, None)])
    msg("TestVideoShotWindows", [(1, "video_id", F.TYPE_INT32, O, None), (2, "positive_shot_id", F.TYPE_INT32, R, None),
                                 (3, "video_name", F.TYPE_STRING, O, None), (4, "positive_shot_words", F.TYPE_MESSAGE, R, ".vss.Datum"),
                                 (5, "context_shot_words", F.TYPE_MESSAGE, R, ".vss.Datum"),
                                 (6, "negative_shot_words", F.TYPE_MESSAGE, R, ".vss.Datum"), (7, "negative_shot_id", F.TYPE_INT32, R, None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    get = getattr(message_factory, "GetMessageClass", None)
    cls = (lambda n: get(pool.FindMessageTypeByName(n))) if get else \
        (lambda n: message_factory.MessageFactory(pool).GetPrototype(pool.FindMessageTypeByName(n)))
    return {n: cls("vss." + n) for n in ("Datum", "VideoShots", "TestVideoShotWindows")}


def fnv1a64(b):
    h = 1469598103934665603
    for c in b:
        h = ((h ^ c) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_tree_walk_small_big_and_multilevel(tool, tmp_path):
    rng = np.random.default_rng(5)
    items = []
    for i in range(6000):                                   # > 100 leaves -> two branch levels
        n = int(rng.integers(0, 300)) if i % 7 else int(rng.integers(5000, 40000))      # inline and overflow values
        items.append((b"%08d_key" % (i * 3), rng.integers(0, 256, n, dtype=np.uint8).tobytes()))
    items.append((b"", b"empty key"))
    stats = write_lmdb(str(tmp_path / "db"), items)
    assert stats["depth"] >= 3 and stats["overflow"] > 0
    out = tmp_path / "dump.txt"
    subprocess.check_call([tool, "lmdbdump", str(tmp_path / "db"), str(out)])
    lines = out.read_text().splitlines()
    assert lines[0] == "entries %d" % len(items)
    want = ["%s %d %016x" % (k.decode(), len(v), fnv1a64(v)) for k, v in sorted(items)]
    assert lines[1:] == want


def test_empty_database_and_bad_files(tool, tmp_path):
    write_lmdb(str(tmp_path / "empty"), [])
    out = tmp_path / "o.txt"
    subprocess.check_call([tool, "lmdbdump", str(tmp_path / "empty"), str(out)])
    assert out.read_text() == "entries 0\n"
    os.makedirs(tmp_path / "junk")
    (tmp_path / "junk" / "data.mdb").write_bytes(b"\0" * 10000)
    r = subprocess.run([tool, "lmdbdump", str(tmp_path / "junk"), str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "magic" in r.stderr
    r = subprocess.run([tool, "lmdbdump", str(tmp_path / "nothing"), str(out)], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


def make_shots_db(pb, path, n_videos=23, F=96, seed=3, packed_every=0):
    rng = np.random.default_rng(seed)
    items, vids = [], []
    for v in range(n_videos):
        n = int(rng.integers(3, 40))
        m = pb["VideoShots"]()
        m.video_id = 1000 + 7 * v
        m.video_name = "HVC%06d" % v
        feats = (rng.integers(0, 32, (n, F)) / 8).astype(np.float32)
        ids = (np.arange(n) * 2 + 5).tolist()
        for j in range(n):
            m.shot_ids.append(ids[j])
            m.shot_words.add(channels=F, height=1, width=1).float_data.extend(feats[j].tolist())
        items.append((b"%08d_%s" % (v, m.video_name.encode()), m.SerializeToString()))
        vids.append((m.video_id, ids, feats))
    write_lmdb(path, items)
    return vids


def test_videoshots_records_decode(tool, pb, tmp_path):
    vids = make_shots_db(pb, str(tmp_path / "train_db"))
    out = tmp_path / "ds.txt"
    subprocess.check_call([tool, "dbload", str(tmp_path / "train_db"), "shots", str(out)], stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    rows = sum(len(i) for _, i, _ in vids)
    total = float(sum(f.astype(np.float64).sum() for _, _, f in vids))
    head = lines[0].split()
    assert head[:6] == ["rows", str(rows), "F", "96", "videos", str(len(vids))]
    assert abs(float(head[-1]) - total) < 1e-3
    base = 0
    for (vid, ids, _), line in zip(vids, lines[1:]):
        assert line == "video %d n %d base %d ids %s" % (vid, len(ids), base, " ".join(map(str, ids)))
        base += len(ids)


def test_test_window_records_decode(tool, pb, tmp_path):
    rng = np.random.default_rng(11)
    items, want, total = [], [], 0.0
    for w in range(57):
        m = pb["TestVideoShotWindows"]()
        m.video_id = int(rng.integers(0, 20))
        for _ in range(4):
            f = (rng.integers(0, 32, 64) / 8).astype(np.float32)
            total += float(f.sum())
            m.context_shot_words.add().float_data.extend(f.tolist())
        items.append((b"%08d" % w, m.SerializeToString()))
        want.append("window %d row0 %d" % (m.video_id, 4 * w))
    write_lmdb(str(tmp_path / "test_db"), items)
    out = tmp_path / "ds.txt"
    subprocess.check_call([tool, "dbload", str(tmp_path / "test_db"), "windows", str(out)], stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    head = lines[0].split()
    assert head[:14] == ["rows", "228", "F", "64", "videos", "0", "windows", "57", "k", "4", "pos", "0", "neg", "0"]
    assert abs(float(head[-1]) - total) < 1e-3
    assert lines[1:] == want


def make_windows_db(pb, path, n_windows=23, k=4, npos=2, nneg=3, F=32, seed=5):
    """TestVideoShotWindows records WITH positive and negative shot words, serialised in field order (positives,
    context, negatives).  Returns [(video_id, ctx [k,F], pos [npos,F], neg [nneg,F])]."""
    rng = np.random.default_rng(seed)
    items, out = [], []
    for w in range(n_windows):
        m = pb["TestVideoShotWindows"]()
        m.video_id = int(rng.integers(0, 9))
        groups = []
        for field, n in ((m.context_shot_words, k), (m.positive_shot_words, npos), (m.negative_shot_words, nneg)):
            f = (rng.integers(0, 32, (n, F)) / 8).astype(np.float32)
            for j in range(n):
                field.add().float_data.extend(f[j].tolist())
            groups.append(f)
        m.positive_shot_id.extend(range(100, 100 + npos))
        m.negative_shot_id.extend(range(200, 200 + nneg))
        items.append((b"%08d" % w, m.SerializeToString()))
        out.append((m.video_id,) + tuple(groups))
    write_lmdb(path, items)
    return out


def test_test_windows_with_positives_and_negatives_decode_in_channel_order(tool, pb, tmp_path):
    # video_shot_window_test_data_layer.cpp:207-232: context words, then positives, then negatives -- the wire order
    # of the record is positives (field 4), context (5), negatives (6)
    wins = make_windows_db(pb, str(tmp_path / "test_db"))
    out = tmp_path / "ds.txt"
    subprocess.check_call([tool, "dbload", str(tmp_path / "test_db"), "windows", str(out)], stderr=subprocess.DEVNULL)
    lines = out.read_text().splitlines()
    head = lines[0].split()
    assert head[:14] == ["rows", str(23 * 9), "F", "32", "videos", "0", "windows", "23", "k", "4", "pos", "2", "neg", "3"]
    for w, (vid, ctx, pos, neg) in enumerate(wins):
        first = " ".join("%g" % x for x in np.concatenate([ctx[:, 0], pos[:, 0], neg[:, 0]]))
        assert lines[1 + w] == "window %d row0 %d first %s" % (vid, 9 * w, first)


def test_large_records_spanning_hundreds_of_overflow_pages(tool, pb, tmp_path):
    # fc7-sized records: 4096 unpacked floats per shot (5 bytes each on the wire), ~60 shots -> >1 MB per value
    import time
    vids = make_shots_db(pb, str(tmp_path / "big_db"), n_videos=6, F=4096, seed=21)
    out = tmp_path / "ds.txt"
    t0 = time.time()
    subprocess.check_call([tool, "dbload", str(tmp_path / "big_db"), "shots", str(out)], stderr=subprocess.DEVNULL)
    dt = time.time() - t0
    head = out.read_text().splitlines()[0].split()
    rows = sum(len(i) for _, i, _ in vids)
    total = float(sum(f.astype(np.float64).sum() for _, _, f in vids))
    assert head[:4] == ["rows", str(rows), "F", "4096"] and abs(float(head[-1]) - total) < 1e-2
    assert os.path.getsize(tmp_path / "big_db" / "data.mdb") > 2 * 1024 * 1024 and dt < 20
