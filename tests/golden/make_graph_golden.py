"""Writes tests/golden/mednet_train_graph.json: the TOPOLOGY of the reference's own training net -- which layers the TRAIN phase
holds, of which type, wired through which blobs, with which layer parameters -- read out of the reference's project file
/root/reference/projects/videovec_embedding/mednet_embedding_train.prototxt (a model definition, i.e. data; the file itself is not
copied).  The fixture is what pins the WIRING of the assembled graph to the reference: tests/test_oracle_graph_topology.py executes
it layer by layer with the oracle's (separately pinned) layer functions and compares with the oracle's assembled step, and checks that
the product's generator (videovector_amd/prototxt.py) produces exactly this graph.  Run in the build container (the reference tree is
not on the GPU box):  python tests/golden/make_graph_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from prototxt_parse import parse_prototxt, train_topology   # noqa: E402  (tests/prototxt_parse.py)

REF = "/root/reference/projects/videovec_embedding/mednet_embedding_train.prototxt"

if __name__ == "__main__":
    net = parse_prototxt(open(REF).read())
    topo = train_topology(net)
    topo["_source"] = "projects/videovec_embedding/mednet_embedding_train.prototxt (TRAIN phase), via tests/golden/make_graph_golden.py"
    topo["test_layers"] = train_topology(net, "TEST")["layers"]       # the TEST phase of the same file (the extraction / retrieval branch)
    json.dump(topo, open(os.path.join(HERE, "mednet_train_graph.json"), "w"), indent=1, sort_keys=True)
    print("%d TRAIN layers, %d TEST layers" % (len(topo["layers"]), len(topo["test_layers"])))
