// dump_blobs -- runs ONE ForwardBackward of a TRAIN-phase net and writes the named blobs, as
// Net::blob_by_name (net.cpp:846-857) hands them out, to <out_dir>/<blob>.bin:
//   int32 num, channels, height, width, then float32 data.
// Used by tests/test_gpu_facade.py to check that every named blob of the graph can be inspected although the
// graph runs as a fused plan.
//   dump_blobs net.prototxt weights.caffemodel out_dir blob1[,blob2,...] [iterations=1]
#include <sys/stat.h>

#include <fstream>

#include "caffe/net.hpp"

using namespace caffe;

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: dump_blobs net.prototxt weights.caffemodel out_dir blob1[,blob2,...] [iterations]\n"); return 2; }
  Caffe::SetDevice(0);
  Caffe::set_mode(Caffe::GPU);
  Caffe::set_phase(Caffe::TRAIN);
  Net<float> net(argv[1], Caffe::TRAIN);
  net.CopyTrainedLayersFrom(string(argv[2]));
  const int iters = argc > 5 ? atoi(argv[5]) : 1;
  vector<Blob<float>*> bottom;
  for (int i = 0; i < iters; ++i) net.ForwardBackward(bottom);
  mkdir(argv[3], 0775);
  string names = argv[4];
  size_t p = 0;
  while (p <= names.size()) {
    size_t e = names.find(',', p); if (e == string::npos) e = names.size();
    const string n = names.substr(p, e - p);
    p = e + 1;
    if (n.empty()) continue;
    CHECK(net.has_blob(n)) << "Unknown blob " << n;
    const shared_ptr<Blob<float> > b = net.blob_by_name(n);
    std::ofstream f(string(argv[3]) + "/" + n + ".bin", std::ios::binary);
    const int32_t shape[4] = {b->num(), b->channels(), b->height(), b->width()};
    f.write((const char*)shape, sizeof(shape));
    f.write((const char*)b->cpu_data(), sizeof(float) * (size_t)b->count());
  }
  return 0;
}
