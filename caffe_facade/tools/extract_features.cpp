// extract_features -- the reference's feature extraction tool (tools/extract_features.cpp:32-211) for
// the videovec path: TEST-phase net, two caffemodels loaded by layer name (the second overwrites
// `fc7`), num_mini_batches forward passes, every row of the named blobs written as one line of
// "<dir>/text_output.txt" ("#features" header, "%g," per value -- the reference streams floats with the
// default ostream precision).  The LevelDB copy of the same rows is not written (no leveldb here).
//   extract_features pretrained_net_param imagenet_net_param feature_extraction_proto_file
//                    blob_name1[,name2,...] save_dir1[,dir2,...] num_mini_batches [GPU] [DEVICE_ID=0]
// The upstream CaffeNet (conv1..fc6) is not part of this build: the prototxt's data layer must
// deliver pre-extracted fc6 rows (VIDEO_SHOT_WINDOW_TEST_DATA with one context frame per record).
#include <sys/stat.h>

#include <cstring>

#include <fstream>

#include "caffe/net.hpp"

using namespace caffe;

static vector<string> split(const string& s) {
  vector<string> out; size_t p = 0;
  while (p <= s.size()) { size_t e = s.find(',', p); if (e == string::npos) e = s.size(); out.push_back(s.substr(p, e - p)); p = e + 1; }
  return out;
}

int main(int argc, char** argv) {
  const int num_required_args = 7;
  if (argc < num_required_args) {
    LOG(ERROR) << "This program takes in a trained network and an input data layer, and then extract features of the "
                  "input data produced by the net.\nUsage: extract_features  pretrained_net_param  imagenet_net_param"
                  "  feature_extraction_proto_file  extract_feature_blob_name1[,name2,...]  save_feature_dir1[,dir2,...]"
                  "  num_mini_batches  [CPU/GPU]  [DEVICE_ID=0]";
    return 1;
  }
  int arg_pos = num_required_args;
  CHECK(argc > arg_pos && strcmp(argv[arg_pos], "GPU") == 0) << "Using CPU is not possible: this build is the GPU path only";
  const int device_id = argc > arg_pos + 1 ? atoi(argv[arg_pos + 1]) : 0;
  LOG(ERROR) << "Using GPU";
  LOG(ERROR) << "Using Device_id=" << device_id;
  Caffe::SetDevice(device_id);
  Caffe::set_mode(Caffe::GPU);
  Caffe::set_phase(Caffe::TEST);
  arg_pos = 0;
  const string pretrained_binary_proto(argv[++arg_pos]);
  const string imagenet_binary_proto(argv[++arg_pos]);
  const string feature_extraction_proto(argv[++arg_pos]);
  shared_ptr<Net<float> > net(new Net<float>(feature_extraction_proto, Caffe::TEST));
  if (imagenet_binary_proto != "-" && imagenet_binary_proto != "none") net->CopyTrainedLayersFrom(imagenet_binary_proto);
  net->CopyTrainedLayersFrom(pretrained_binary_proto);
  const vector<string> blob_names = split(argv[++arg_pos]);
  const vector<string> dir_names = split(argv[++arg_pos]);
  CHECK_EQ(blob_names.size(), dir_names.size()) << " the number of blob names and leveldb names must be equal";
  for (size_t i = 0; i < blob_names.size(); ++i)
    CHECK(net->has_blob(blob_names[i])) << "Unknown feature blob name " << blob_names[i] << " in the network " << feature_extraction_proto;
  vector<shared_ptr<std::ofstream> > texts;
  for (size_t i = 0; i < dir_names.size(); ++i) {
    mkdir(dir_names[i].c_str(), 0775);
    texts.push_back(shared_ptr<std::ofstream>(new std::ofstream((dir_names[i] + "/text_output.txt").c_str())));
    CHECK(texts.back()->good()) << "Failed to open " << dir_names[i] << "/text_output.txt";
    LOG(ERROR) << "Opened: " << dir_names[i] + "/text_output.txt";
    (*texts.back()) << "#features\n";
  }
  const int num_mini_batches = atoi(argv[++arg_pos]);
  LOG(ERROR) << "Extacting Features";
  vector<Blob<float>*> input_vec;
  vector<int> image_indices(blob_names.size(), 0);
  for (int batch_index = 0; batch_index < num_mini_batches; ++batch_index) {
    net->Forward(input_vec);
    for (size_t i = 0; i < blob_names.size(); ++i) {
      const shared_ptr<Blob<float> > feature_blob = net->blob_by_name(blob_names[i]);
      const int batch_size = feature_blob->num();
      const int dim_features = feature_blob->count() / batch_size;
      for (int n = 0; n < batch_size; ++n) {
        const float* d = feature_blob->cpu_data() + feature_blob->offset(n);
        for (int k = 0; k < dim_features; ++k) (*texts[i]) << d[k] << ",";
        (*texts[i]) << "\n";
        if (++image_indices[i] % 1000 == 0)
          LOG(ERROR) << "Extracted features of " << image_indices[i] << " query images for feature blob " << blob_names[i];
      }
    }
  }
  for (size_t i = 0; i < blob_names.size(); ++i) {
    texts[i]->close();
    LOG(ERROR) << "Extracted features of " << image_indices[i] << " query images for feature blob " << blob_names[i];
  }
  LOG(ERROR) << "Successfully extracted the features!";
  return 0;
}
