"""Driver of tools/asan_host.sh: valid / truncated / corrupted LMDB files and VideoShots records through the
sanitizer build of proto_tool; fails when the sanitizers report anything."""
import os
import random
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from lmdb_writer import write_lmdb  # noqa: E402

tool = sys.argv[1]
env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
d = tempfile.mkdtemp()
reports = 0


def run(args):
    global reports
    r = subprocess.run([tool] + args, capture_output=True, text=True, env=env)
    if "AddressSanitizer" in r.stderr or "runtime error" in r.stderr:
        reports += 1
        print("SANITIZER REPORT for", args, r.stderr[-600:])
    return r


rng = np.random.default_rng(5)
items = [(b"%08d_key" % (i * 3), rng.integers(0, 256, int(rng.integers(0, 300)) if i % 7 else int(rng.integers(5000, 40000)),
                                             dtype=np.uint8).tobytes()) for i in range(3000)]
write_lmdb(d + "/db", items)
assert run(["lmdbdump", d + "/db", d + "/o.txt"]).returncode == 0
raw = open(d + "/db/data.mdb", "rb").read()
random.seed(1)
for t in range(12):
    b = bytearray(raw)
    if t == 0:
        b = b[:len(b) // 2]
    else:
        for _ in range(300):
            b[random.randrange(8192, len(b))] = random.randrange(256)
    os.makedirs(d + "/c%d" % t)
    open(d + "/c%d/data.mdb" % t, "wb").write(bytes(b))
    run(["lmdbdump", d + "/c%d" % t, d + "/o.txt"])


def varint(v):
    out = bytearray()
    while True:
        out.append((v & 0x7F) | (0x80 if v > 0x7F else 0))
        v >>= 7
        if not v:
            return bytes(out)


def video_shots(vid, n, F):          # hand-encoded VideoShots record (video_shot_sentences.proto:14-19)
    rec = b"\x08" + varint(vid)
    for j in range(n):
        rec += b"\x10" + varint(j)
    for j in range(n):
        datum = b"".join(b"\x35" + np.float32(x).tobytes() for x in (rng.integers(0, 32, F) / 8))
        rec += b"\x1a" + varint(len(datum)) + datum
    return rec


recs = [(b"%04d" % v, video_shots(v, int(rng.integers(3, 12)), 64)) for v in range(8)]
random.seed(3)
for t in range(40):
    its = []
    for k, val in recs:
        b = bytearray(val)
        if t > 0:
            for _ in range(3):
                b[random.randrange(len(b))] = random.randrange(256)
            if t % 5 == 0:
                b = b[:random.randrange(1, len(b))]
        its.append((k, bytes(b)))
    write_lmdb(d + "/s%d" % t, its)
    r = run(["dbload", d + "/s%d" % t, "shots", d + "/o.txt"])
    assert t > 0 or r.returncode == 0, r.stderr[-400:]
ref = "/root/reference/projects/videovec_embedding/mednet_embedding_train.prototxt"
if os.path.exists(ref):
    assert run(["filter", ref, "TRAIN", d + "/f.prototxt"]).returncode == 0
print("sanitizer reports:", reports)
sys.exit(1 if reports else 0)
