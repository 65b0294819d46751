#!/bin/bash
# The host sampler (videovector_amd/csrc/sampler.cc) under ThreadSanitizer and under AddressSanitizer + UBSan: serial, 1-4 thread
# pipelines and the shared-memory ring with two consumers against the same stream.  CPU only.
set -e
cd "$(dirname "$0")/.."
mkdir -p /tmp/san
for mode in thread address,undefined; do
  g++ -O1 -g -std=c++17 -march=x86-64-v3 -pthread -fsanitize=$mode -fno-omit-frame-pointer -Iinclude \
      -o /tmp/san/sampler_$$ tools/san_sampler_main.cc videovector_amd/csrc/sampler.cc -lrt
  echo "== -fsanitize=$mode"
  VV_SAMPLER_PIN=0 /tmp/san/sampler_$$
  rm -f /tmp/san/sampler_$$
done
