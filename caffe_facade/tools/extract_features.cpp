// extract_features -- feature extraction for the videovec path (the reference's tools/extract_features.cpp:32-211).
//
//   extract_features <trained.caffemodel> <base.caffemodel | - | none> <net.prototxt>
//                    <blob[,blob...]> <dir[,dir...]> <num_mini_batches> GPU [device_id] [--num_classes=N]
//
// What is contract with the reference: the positional arguments and their order; the two-stage load by layer name
// (the base model first, then the trained one, so the trained `fc7` wins: :99-103); num_mini_batches forward passes of
// the TEST-phase net; and the text file "<dir>/text_output.txt" -- a "#features" line, then one line per row of the
// blob, every value followed by a comma, values streamed with the default ostream precision (:127-129, 162-164).
// The LevelDB copy of the same rows (:121-126, 166-176) is not written: there is no leveldb in this build.
// The upstream CaffeNet (conv1..fc6) is not part of this build either: the prototxt's data layer must deliver
// pre-extracted fc6 rows (VIDEO_SHOT_WINDOW_TEST_DATA with one context frame per record).
#include <sys/stat.h>

#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>

#include "caffe/net.hpp"

using namespace caffe;

namespace {

struct Args {
  string trained, base, prototxt;
  vector<string> blobs, dirs;
  int batches = 0, device = 0;
};

vector<string> SplitCommas(const string& s) {
  vector<string> parts;
  std::stringstream ss(s);
  for (string item; std::getline(ss, item, ',');) parts.push_back(item);
  if (!s.empty() && s.back() == ',') parts.push_back("");
  return parts;
}

const char kUsage[] =
    "This program takes in a trained network and an input data layer, and then extract features of the input data "
    "produced by the net.\nUsage: extract_features  pretrained_net_param  imagenet_net_param  feature_extraction_proto_file"
    "  extract_feature_blob_name1[,name2,...]  save_feature_dir1[,dir2,...]  num_mini_batches  [CPU/GPU]  [DEVICE_ID=0]";

bool Parse(int argc, char** argv, Args* a) {
  vector<string> pos;
  for (int i = 1; i < argc; ++i) {
    const string s = argv[i];
    if (s.compare(0, 14, "--num_classes=") == 0) FLAGS_num_classes = atoi(s.c_str() + 14);
    else if (s.compare(0, 21, "--max_tries_for_negs=") == 0) FLAGS_max_tries_for_negs = atoi(s.c_str() + 21);
    else pos.push_back(s);
  }
  if (pos.size() < 6) return false;
  a->trained = pos[0]; a->base = pos[1]; a->prototxt = pos[2];
  a->blobs = SplitCommas(pos[3]); a->dirs = SplitCommas(pos[4]);
  a->batches = atoi(pos[5].c_str());
  CHECK(pos.size() > 6 && pos[6] == "GPU") << "Using CPU is not possible: this build is the GPU path only";
  if (pos.size() > 7) a->device = atoi(pos[7].c_str());
  return true;
}

// one output file per requested blob
class FeatureText {
 public:
  FeatureText(const string& dir, const string& blob) : blob_(blob), path_(dir + "/text_output.txt") {
    mkdir(dir.c_str(), 0775);
    out_.open(path_.c_str());
    CHECK(out_.good()) << "Failed to open " << path_;
    LOG(ERROR) << "Opened: " << path_;
    out_ << "#features\n";
  }
  void Append(const Blob<float>& b) {
    const int rows = b.num(), dim = b.count() / b.num();
    for (int r = 0; r < rows; ++r) {
      const float* v = b.cpu_data() + b.offset(r);
      for (int k = 0; k < dim; ++k) out_ << v[k] << ",";
      out_ << "\n";
      if (++rows_ % 1000 == 0) Report();
    }
  }
  void Close() { out_.close(); Report(); }
 private:
  void Report() const { LOG(ERROR) << "Extracted features of " << rows_ << " query images for feature blob " << blob_; }
  string blob_, path_;
  std::ofstream out_;
  int rows_ = 0;
};

}  // namespace

int main(int argc, char** argv) {
  Args a;
  if (!Parse(argc, argv, &a)) { LOG(ERROR) << kUsage; return 1; }
  LOG(ERROR) << "Using GPU";
  LOG(ERROR) << "Using Device_id=" << a.device;
  Caffe::SetDevice(a.device);
  Caffe::set_mode(Caffe::GPU);
  Caffe::set_phase(Caffe::TEST);

  Net<float> net(a.prototxt, Caffe::TEST);
  if (a.base != "-" && a.base != "none") net.CopyTrainedLayersFrom(a.base);
  net.CopyTrainedLayersFrom(a.trained);

  CHECK_EQ(a.blobs.size(), a.dirs.size()) << " the number of blob names and leveldb names must be equal";
  vector<std::unique_ptr<FeatureText> > files;
  for (size_t i = 0; i < a.blobs.size(); ++i) {
    CHECK(net.has_blob(a.blobs[i])) << "Unknown feature blob name " << a.blobs[i] << " in the network " << a.prototxt;
    files.emplace_back(new FeatureText(a.dirs[i], a.blobs[i]));
  }
  LOG(ERROR) << "Extracting Features";
  for (int pass = 0; pass < a.batches; ++pass) {
    net.Forward(vector<Blob<float>*>());
    for (size_t i = 0; i < files.size(); ++i) files[i]->Append(*net.blob_by_name(a.blobs[i]));
  }
  for (auto& f : files) f->Close();
  LOG(ERROR) << "Successfully extracted the features!";
  return 0;
}
