"""Debug session: dump data / x / ip2 of the positives-negatives test net (tests/test_gpu_lmdb.py) and compare each
with numpy, to find which layer diverges."""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from tests.test_facade_lmdb import make_windows_db
from tests import test_facade_lmdb as tfl
from tests import test_facade_proto as tfp
from tests.test_gpu_facade import write_caffemodel, CAFFE
from videovector_amd.synth import init_weights
from oracle import oracle
oracle.build()

def _raw(fx):
    return getattr(fx, "__wrapped__", None) or fx.__pytest_wrapped__.obj
lmdb_pb = _raw(tfl.pb)()
pbm = _raw(tfp.pb)()
B, k, npos, nneg, F, D = 5, 4, 2, 3, 32, 16
for inc_pos, inc_neg in ((True, False), (False, False), (True, True)):
    tmp = tempfile.mkdtemp()
    wins = make_windows_db(lmdb_pb, tmp + "/test_db", n_windows=11, k=k, npos=npos, nneg=nneg, F=F)
    ch = k + (npos if inc_pos else 0) + (nneg if inc_neg else 0)
    tops = ["w%d" % c for c in range(ch)]
    net = ['name: "windows_with_labels"',
           'layers {\n  name: "win"\n  type: VIDEO_SHOT_WINDOW_TEST_DATA\n  top: "data"\n  top: "label"\n'
           '  video_shot_window_test_data_param {\n    source: "%s"\n    backend: LMDB\n    batch_size: %d\n'
           '    include_positives: %s\n    include_negatives: %s\n  }\n}'
           % (tmp + "/test_db", B, str(inc_pos).lower(), str(inc_neg).lower()),
           'layers {\n  name: "sl"\n  type: SLICE\n  bottom: "data"\n%s\n}' % "\n".join('  top: "%s"' % t for t in tops),
           'layers {\n  name: "cat"\n  type: CONCAT\n%s\n  top: "rows"\n  concat_param { concat_dim: 0 }\n}'
           % "\n".join('  bottom: "%s"' % t for t in tops),
           'layers {\n  name: "flat"\n  type: FLATTEN\n  bottom: "rows"\n  top: "x"\n}',
           'layers {\n  name: "fc7"\n  type: INNER_PRODUCT\n  bottom: "x"\n  top: "ip1_nonorm"\n  inner_product_param {\n'
           '    num_output: %d\n    weight_filler { type: "gaussian" std: 0.02 }\n    bias_filler { type: "constant" }\n  }\n}' % D,
           'layers {\n  name: "fc7_relu"\n  type: RELU\n  bottom: "ip1_nonorm"\n  top: "ip2"\n}']
    open(tmp + "/net.prototxt", "w").write("\n".join(net) + "\n")
    W0, b0 = init_weights(4, D, F, std=0.05)
    write_caffemodel(pbm, tmp + "/w.caffemodel", W0, b0)
    for rep in range(3):
        r = subprocess.run([os.path.join(os.path.dirname(CAFFE), "extract_features"), tmp + "/w.caffemodel", "none",
                            tmp + "/net.prototxt", "data,x,ip1_nonorm,ip2", "%s/fd%d,%s/fx%d,%s/fy%d,%s/fh%d" % ((tmp, rep) * 4), "1", "GPU", "0"],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        def rd(n):
            lines = open("%s/%s%d/text_output.txt" % (tmp, n, rep)).read().strip().split("\n")
            return np.array([[float(x) for x in l.rstrip(",").split(",")] for l in lines[1:]], np.float32)
        data, x, y, h = rd("fd"), rd("fx"), rd("fy"), rd("fh")
        items = []
        for i in range(B):
            _, ctx, pos, neg = wins[i]
            items.append(np.concatenate([ctx] + ([pos] if inc_pos else []) + ([neg] if inc_neg else [])))
        it = np.stack(items)
        xe = it.transpose(1, 0, 2).reshape(ch * B, F)
        ye = xe @ W0.T + b0
        print(inc_pos, inc_neg, rep, "data", np.abs(data - it.reshape(B, -1)).max(), "x", np.abs(x - xe).max(),
              "y", np.abs(y - ye).max(), "h", np.abs(h - np.maximum(ye, 0)).max(), flush=True)
