#!/bin/bash
# Host-side code (proto codec, LMDB reader, record decoders, graph filter) under AddressSanitizer + UBSan, fed valid,
# truncated and randomly corrupted inputs.  CPU only (GPU sanitizers are not available on the pool).
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/asan
g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -Icaffe_facade/include \
    -o /tmp/asan/proto_tool caffe_facade/tools/proto_tool.cpp caffe_facade/src/proto_lite.cpp caffe_facade/src/lmdb_reader.cpp \
    caffe_facade/src/facade.cpp caffe_facade/src/layers_gpu.cpp caffe_facade/src/net.cpp caffe_facade/src/solver.cpp \
    -Lvideovector_amd/lib -lvideovec -Wl,-rpath,$PWD/videovector_amd/lib
python3 tools/asan_host_fuzz.py /tmp/asan/proto_tool
