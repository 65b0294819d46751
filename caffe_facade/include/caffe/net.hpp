// net.hpp -- Net with the reference's interface (include/caffe/net.hpp, src/caffe/net.cpp).
// Init performs the reference's phase filtering (net.cpp:226-329), blob wiring (333-402) and
// per-layer SetUp (shape inference), then a graph matcher recognises the videovec TRAIN graph
// (projects/videovec_embedding/mednet_embedding_train.prototxt:1-671) and installs the fused HIP
// plan; ForwardBackward / Update run that plan through the C ABI (include/videovec.h).  Any other
// graph is fatal: this build is a drop-in for that path only.
#pragma once
#include <map>
#include <set>

#include "caffe/layer.hpp"

namespace caffe {

typedef pl::Message NetParameter;
typedef pl::Message NetState;

template <typename Dtype>
class Net {
 public:
  explicit Net(const NetParameter& param) { Init(param); }
  explicit Net(const string& param_file, Caffe::Phase phase = Caffe::TRAIN);
  virtual ~Net();
  void Init(const NetParameter& param);

  // net.hpp:78-83: forward + backward of one prefetched batch; returns the weighted loss
  Dtype ForwardBackward(const vector<Blob<Dtype>*>& bottom);
  const vector<Blob<Dtype>*>& Forward(const vector<Blob<Dtype>*>& bottom, Dtype* loss = NULL);
  const vector<Blob<Dtype>*>& ForwardPrefilled(Dtype* loss = NULL);
  void Backward() {}                              // gradients are produced by ForwardBackward
  // layer-by-layer execution (net.cpp:501-514, 567-578); used when the graph is not the fused pattern
  Dtype ForwardFromTo(int start, int end);
  void BackwardFromTo(int start, int end);
  bool sequential() const { return sequential_; }
  // net.cpp:803-839: applies the update prepared by the solver (fused decay + momentum + step)
  void Update();
  void SetUpdateHyperParams(float rate, float momentum, float weight_decay, const string& reg, int solver_type = 0,
                            float delta = 1e-8f);

  // net.cpp:638-667: a TEST net takes the weights of the train net (device-to-device through the host)
  void ShareTrainedLayersWith(Net* other);
  void CopyTrainedLayersFrom(const NetParameter& param);        // net.cpp:691-764
  void CopyTrainedLayersFrom(const string trained_filename);
  void ToProto(NetParameter* param, bool write_diff = false);   // net.cpp:773-801

  inline const string& name() const { return name_; }
  inline const vector<string>& layer_names() const { return layer_names_; }
  inline const vector<string>& blob_names() const { return blob_names_; }
  inline const vector<shared_ptr<Blob<Dtype> > >& blobs() const { return blobs_; }
  inline const vector<shared_ptr<Layer<Dtype> > >& layers() const { return layers_; }
  // parameter blobs (host mirrors, refreshed from the device on access)
  vector<shared_ptr<Blob<Dtype> > >& params();
  inline vector<float>& params_lr() { return params_lr_; }
  inline vector<float>& params_weight_decay() { return params_weight_decay_; }
  inline const vector<Blob<Dtype>*>& output_blobs() const { return net_output_blobs_; }
  inline const vector<int>& output_blob_indices() const { return net_output_blob_indices_; }
  inline const vector<Dtype>& blob_loss_weights() const { return blob_loss_weights_; }
  bool has_blob(const string& blob_name);
  // net.cpp:846-857.  Blobs the fused plan can materialise are filled from the device:
  // ip2, target_score, negative_scores, loss_output, train_violations (names from the prototxt).
  const shared_ptr<Blob<Dtype> > blob_by_name(const string& blob_name);
  bool has_layer(const string& layer_name);
  const shared_ptr<Layer<Dtype> > layer_by_name(const string& layer_name);
  void set_debug_info(const bool value) { debug_info_ = value; }
  // Solver::Step: the next ForwardBackward is followed by Update with the hyper-parameters already set (vv_update_hint, include/videovec.h:
  // the library may apply the update where the gradient is produced).  Not with debug_info (its lines read the diffs) nor on the
  // layer-by-layer executor.
  void HintUpdate();
  // The reference's Solver::Solve uses ForwardBackward's return value only on display iterations
  // (solver.cpp:194-196).  Reading the loss back costs a device synchronisation, so the solver says when it
  // wants it; otherwise ForwardBackward returns the last value read and the GPU keeps running ahead.
  void set_loss_needed(const bool value) { loss_needed_ = value; }
  // SGD history of the parameter blobs (owned by the device context; host views on demand)
  void GetHistory(vector<shared_ptr<Blob<Dtype> > >* history);
  void SetHistory(const vector<shared_ptr<Blob<Dtype> > >& history);

  static void FilterNet(const NetParameter& param, NetParameter* param_filtered);   // net.cpp:226-268
  static bool StateMeetsRule(const NetState& state, const pl::Message& rule, const string& layer_name);

  // description of the recognised graph (for logs / tests)
  struct FusedPlan {
    int B = 0, C = 0, Nn = 0, F = 0, D = 0;
    float margin = 1.f; int norm = VV_NORM_L1; float loss_weight = 1.f;
    vector<float> ctx_coeff; float dropout_ratio = 0.f; float ip_regularization = 0.f; bool weighted_loss = false;
    int data_layer = -1, ip_layer = -1, loss_layer = -1;
    string ip2_blob, target_score_blob, negative_scores_blob, loss_blob, violations_blob;
    // TEST / extraction graph: window-mean -> fc7 -> ReLU [-> NORMALIZATION] [-> RETRIEVAL_STATS]
    bool test = false;
    int stats_layer = -1; bool relu = false;
    string ip1_blob, norm_blob, label_blob;
    string stat_blobs[3];
  };
  const FusedPlan& plan() const { return plan_; }

 protected:
  void MatchVideovecTrainGraph();
  void MatchVideovecTestGraph();
  void SetUpSequential(bool is_test);
  Dtype SequentialStep();
  bool sequential_ = false;
  vector<bool> layer_need_backward_;
  vector<vector<bool> > bottom_need_backward_;
  Dtype ForwardTest();
  vv_ctx* ctx_ = nullptr;            // train net: the process context; a TEST net owns a second one
  bool own_ctx_ = false;
  void PushParamsToDevice();
  void PullParamsFromDevice();
  vector<shared_ptr<Layer<Dtype> > > layers_;
  vector<string> layer_names_;
  std::map<string, int> layer_names_index_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;
  vector<string> blob_names_;
  std::map<string, int> blob_names_index_;
  vector<vector<Blob<Dtype>*> > bottom_vecs_, top_vecs_;
  vector<vector<int> > bottom_id_vecs_, top_id_vecs_;
  vector<Dtype> blob_loss_weights_;
  vector<int> net_output_blob_indices_;
  vector<Blob<Dtype>*> net_output_blobs_;
  string name_;
  vector<shared_ptr<Blob<Dtype> > > params_;
  vector<float> params_lr_, params_weight_decay_;
  bool params_stale_ = false;        // device copy is newer than the host blobs
  bool debug_info_ = false;
  FusedPlan plan_;
  struct BlobSym { int kind = 0, a = 0, reps = 1; };
  vector<BlobSym> blob_sym_;     // what every named blob of the TRAIN graph is in terms of the fused plan (blob_by_name)
  void MaterializeTrainBlob(int blob_id);
  vv_step_cfg cfg_;
  vector<int32_t> idx_, last_src_, label_;
  vector<float> item_weight_;
  bool loss_needed_ = true;
  float last_loss_ = 0.f;
  uint64_t iter_ = 0;
};

}  // namespace caffe
