// layer.hpp -- the reference's operator interface (include/caffe/layer.hpp:25-404) and the layer
// classes of the videovec graph.  Same class names, constructor, SetUp / Reshape / Forward /
// Backward signatures (2014-era `vector<Blob*>* top` API), blob-count checks and factory
// (src/caffe/layer_factory.cpp:177-309).
//
// Execution model of this build: layers carry configuration, parameter blobs, shape inference (Reshape) and their
// Forward_gpu / Backward_gpu (layers_gpu.cpp: one small HIP operator each, through vv_op_* of the C ABI).  When
// Net::Init recognises the videovec TRAIN / TEST graph, the whole graph runs as ONE fused HIP plan instead (net.hpp)
// and the per-layer functions are not called; any other arrangement of these layers runs layer by layer
// (Net::ForwardFromTo / BackwardFromTo), as does a single Layer::Forward / Backward.  There is no CPU execution path:
// Forward_cpu / Backward_cpu are fatal (the reference's CPU path is restated only as the test oracle).
#pragma once
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>

#include "caffe/blob.hpp"

namespace caffe {

typedef pl::Message LayerParameter;   // message "LayerParameter" (caffe.proto:215-389)

template <typename Dtype>
class Layer {
 public:
  explicit Layer(const LayerParameter& param);       // copies param.blobs (layer.hpp:33-43)
  virtual ~Layer() {}
  void SetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {   // layer.hpp:59-64
    CheckBlobCounts(bottom, *top);
    LayerSetUp(bottom, top);
    Reshape(bottom, top);
    SetLossWeights(top);
  }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {}
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) = 0;
  Dtype Forward(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  void Backward(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down,
                vector<Blob<Dtype>*>* bottom);
  // Net::BackwardFromTo: bottoms whose diff already holds another consumer's contribution (fan-out without SPLIT
  // layers, see net.hpp) receive this layer's contribution ADDED instead of written
  void set_accumulate_bottom(const vector<bool>& acc) { accumulate_bottom_ = acc; }
  vector<shared_ptr<Blob<Dtype> > >& blobs() { return blobs_; }
  const LayerParameter& layer_param() const { return layer_param_; }
  virtual void ToProto(LayerParameter* param, bool write_diff = false);
  inline Dtype loss(const int top_index) const { return (int)loss_.size() > top_index ? loss_[top_index] : Dtype(0); }
  inline void set_loss(const int top_index, const Dtype value) {
    if ((int)loss_.size() <= top_index) loss_.resize(top_index + 1, Dtype(0));
    loss_[top_index] = value;
  }
  virtual string type() const = 0;                   // LayerType enum name (LayerParameter_LayerType)
  virtual const string type_name() const { return type(); }
  virtual int ExactNumBottomBlobs() const { return -1; }
  virtual int MinBottomBlobs() const { return -1; }
  virtual int MaxBottomBlobs() const { return -1; }
  virtual int ExactNumTopBlobs() const { return -1; }
  virtual int MinTopBlobs() const { return -1; }
  virtual int MaxTopBlobs() const { return -1; }
  virtual bool EqualNumBottomTopBlobs() const { return false; }
  virtual bool AutoTopBlobs() const { return false; }
  virtual bool AllowForceBackward(const int bottom_index) const { return true; }
  inline bool param_propagate_down(const int param_id) { return (int)param_propagate_down_.size() > param_id ? param_propagate_down_[param_id] : false; }
  inline void set_param_propagate_down(const int param_id, const bool value) {
    if ((int)param_propagate_down_.size() <= param_id) param_propagate_down_.resize(param_id + 1, true);
    param_propagate_down_[param_id] = value;
  }
 protected:
  LayerParameter layer_param_;
  vector<shared_ptr<Blob<Dtype> > > blobs_;
  vector<bool> param_propagate_down_;
  vector<Dtype> loss_;
  vector<bool> accumulate_bottom_;
  // layer.hpp:308-337 of the reference.  GPU = HIP here.
  virtual void Forward_cpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) = 0;
  virtual void Backward_cpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom);
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) = 0;
  virtual void CheckBlobCounts(const vector<Blob<Dtype>*>& bottom, const vector<Blob<Dtype>*>& top);   // layer.hpp:346-380
  inline void SetLossWeights(vector<Blob<Dtype>*>* top) {                                              // layer.hpp:387-401
    const int n = layer_param_.size("loss_weight");
    if (n) {
      CHECK_EQ((int)top->size(), n) << "loss_weight must be unspecified or specified once per top blob.";
      for (int i = 0; i < n; ++i) set_loss(i, (Dtype)layer_param_.get_num("loss_weight", i));
    }
  }
};

// ---- dataset behind `source:` of the data layers (stands for the VideoShots LMDB; data_source.cpp)
struct VideoDataset {
  vector<int32_t> video_id, n_shots, shot_ids;
  vector<int64_t> row_base;
  int64_t n_rows = 0;
  int F = 0;
  bool synthetic = false; uint64_t seed = 0;
  vector<float> features;          // [n_rows][F] when not synthetic
  // TEST-phase records (proto TestVideoShotWindows, video_shot_sentences.proto:22-30): each window is
  // win_k context frames of one video, stored as table rows
  int win_k = 0;
  vector<int32_t> win_rows, win_video_id;
  bool SameTable(const VideoDataset& o) const { return synthetic && o.synthetic && seed == o.seed && n_rows == o.n_rows && F == o.F; }
  // per window win_rows holds win_k context rows, then win_pos positive rows, then win_neg negative rows
  int win_pos = 0, win_neg = 0;
  vector<int32_t> win_pos_ids;     // positive_shot_id count of each record
  enum Kind { kShots, kTestWindows };
  static shared_ptr<VideoDataset> Open(const string& source, Kind kind = kShots, const string& backend = "LMDB");
  static shared_ptr<VideoDataset> OpenLmdbVideoShots(const string& source);
  static shared_ptr<VideoDataset> OpenLmdbTestWindows(const string& source);
  void UploadTable(vv_ctx* ctx) const;   // vv_table_synth / vv_table_set
  // VideoSampledShotsDataParameter.negative_dataset: the records that fill the initial negative buffer; their rows
  // are appended to `features`
  vector<int32_t> neg_video_id, neg_n_shots, neg_shot_ids;
  vector<int64_t> neg_row_base;
  void AppendNegatives(const VideoDataset& neg);
};

#define VV_LAYER_GPU_DECL                                                                                              \
 protected:                                                                                                            \
  virtual void Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);                             \
  virtual void Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom);
#define VV_LAYER_BOILER(Name, TypeStr)                                            \
  VV_LAYER_GPU_DECL                                                               \
 public:                                                                           \
  explicit Name(const LayerParameter& param) : Layer<Dtype>(param) {}            \
  virtual string type() const { return TypeStr; }

// VIDEO_SAMPLED_SHOTS_DATA (include/caffe/data_layers.hpp:222-286).  Produces table-row indices
// (vv_sampler_next) instead of feature copies; top[0] keeps the reference shape (B, C+Nn, F, 1).
template <typename Dtype>
class VideoSampledShotsDataLayer : public Layer<Dtype> {
  VV_LAYER_BOILER(VideoSampledShotsDataLayer, "VIDEO_SAMPLED_SHOTS_DATA")
  virtual ~VideoSampledShotsDataLayer();
  virtual int ExactNumBottomBlobs() const { return 0; }
  virtual int MinTopBlobs() const { return 1; }
  virtual int MaxTopBlobs() const { return 2; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {}
  // next prefetch batch: idx / last_src [B][C+Nn], label [B]
  void NextBatch(vector<int32_t>* idx, vector<int32_t>* last_src, vector<int32_t>* label);
  // the batch the last Forward_gpu delivered (layer-by-layer execution)
  const vector<int32_t>& last_idx() const { return fw_idx_; }
  const vector<int32_t>& last_label() const { return fw_label_; }
  int batch_size() const { return batch_size_; }
  int context_size() const { return context_size_; }
  int num_negative_samples() const { return num_negative_samples_; }
  int feature_size() const { return feature_size_; }
  const shared_ptr<VideoDataset>& dataset() const { return dataset_; }
 private:
  shared_ptr<VideoDataset> dataset_;
  // prefetch: the sampler's background threads keep kPrefetchDepth index batches ahead of the solver
  // (BasePrefetchingDataLayer, base_data_layer.cpp:52-95; InternalThread, internal_thread.cpp:14-37)
  static constexpr int kPrefetchDepth = 128;   // covers the stretches in which the sampler falls behind the GPU step
  void CreatePrefetchThread();     // starts the prefetch threads (first call)
  void JoinPrefetchThread();       // the wait happens inside vv_sampler_next
  bool prefetching_ = false;
  int rand_skip_ = 0;
  vector<int32_t> fw_idx_, fw_last_, fw_label_;
  bool per_rank_ = false;                  // data-parallel: every rank runs its own sampler (VV_SAMPLER_MODE, facade.cpp)
  vv_sampler* sampler_ = nullptr;          // data-parallel with ONE logical sampler (VV_SAMPLER_MODE=node): rank 0 only
  vv_batch_ring* ring_ = nullptr;          // the prefetch ring (rank 0: the sampler's own; other ranks: attached by name)
  int batch_size_ = 0, context_size_ = 0, num_negative_samples_ = 0, feature_size_ = 0;
};

template <typename Dtype>
class SliceLayer : public Layer<Dtype> {       // slice_layer.cpp:13-58
  VV_LAYER_BOILER(SliceLayer, "SLICE")
  virtual int ExactNumBottomBlobs() const { return 1; }
  virtual int MinTopBlobs() const { return 2; }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
};
template <typename Dtype>
class ConcatLayer : public Layer<Dtype> {      // concat_layer.cpp:10-53
  VV_LAYER_BOILER(ConcatLayer, "CONCAT")
  virtual int MinBottomBlobs() const { return 2; }
  virtual int ExactNumTopBlobs() const { return 1; }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
};
template <typename Dtype>
class FlattenLayer : public Layer<Dtype> {     // flatten_layer.cpp:9-16
  VV_LAYER_BOILER(FlattenLayer, "FLATTEN")
  virtual int ExactNumBottomBlobs() const { return 1; }
  virtual int ExactNumTopBlobs() const { return 1; }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
    (*top)[0]->Reshape(bottom[0]->num(), bottom[0]->count() / bottom[0]->num(), 1, 1);
  }
};
template <typename Dtype>
class SplitLayer : public Layer<Dtype> {       // split_layer.cpp:10-25
  VV_LAYER_BOILER(SplitLayer, "SPLIT")
  virtual int ExactNumBottomBlobs() const { return 1; }
  virtual int MinTopBlobs() const { return 1; }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
    for (size_t i = 0; i < top->size(); ++i) (*top)[i]->ReshapeLike(*bottom[0]);
  }
};
template <typename Dtype>
class InnerProductLayer : public Layer<Dtype> {   // inner_product_layer.cpp:12-58
  VV_LAYER_BOILER(InnerProductLayer, "INNER_PRODUCT")
  virtual int ExactNumBottomBlobs() const { return 1; }
  virtual int ExactNumTopBlobs() const { return 1; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
    (*top)[0]->Reshape(bottom[0]->num(), N_, 1, 1);
  }
  int num_output() const { return N_; }
 private:
  int N_ = 0, K_ = 0;
  bool bias_term_ = true;
};
template <typename Dtype>
class NeuronShapeLayer : public Layer<Dtype> {    // neuron_layer.cpp: top shaped like bottom (still abstract: no arithmetic)
 public:
  explicit NeuronShapeLayer(const LayerParameter& param) : Layer<Dtype>(param) {}
  virtual int ExactNumBottomBlobs() const { return 1; }
  virtual int ExactNumTopBlobs() const { return 1; }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) { (*top)[0]->ReshapeLike(*bottom[0]); }
};
template <typename Dtype>
class ReLULayer : public NeuronShapeLayer<Dtype> {            // relu_layer.cpp / .cu
  VV_LAYER_GPU_DECL
 public:
  explicit ReLULayer(const LayerParameter& param) : NeuronShapeLayer<Dtype>(param) {}
  virtual string type() const { return "RELU"; }
};
template <typename Dtype>
class DropoutLayer : public NeuronShapeLayer<Dtype> {         // dropout_layer.cpp / .cu
  VV_LAYER_GPU_DECL
 public:
  explicit DropoutLayer(const LayerParameter& param) : NeuronShapeLayer<Dtype>(param) {}
  virtual ~DropoutLayer();
  virtual string type() const { return "DROPOUT"; }
 private:
  void* mask_ = nullptr; int mask_count_ = 0;      // uint8 device mask of the last TRAIN forward (rand_vec_ in the reference)
  uint64_t calls_ = 0;
};
template <typename Dtype>
class NormalizationLayer : public NeuronShapeLayer<Dtype> {   // normalization_layer.cpp / .cu
  VV_LAYER_GPU_DECL
 public:
  explicit NormalizationLayer(const LayerParameter& param) : NeuronShapeLayer<Dtype>(param) {}
  virtual string type() const { return "NORMALIZATION"; }
  // the reference reports MVN here (include/caffe/common_layers.hpp:392-394, quirk Q5)
};
template <typename Dtype>
class EltwiseLayer : public Layer<Dtype> {        // eltwise_layer.cpp:12-50
  VV_LAYER_BOILER(EltwiseLayer, "ELTWISE")
  virtual int MinBottomBlobs() const { return 2; }
  virtual int ExactNumTopBlobs() const { return 1; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  const vector<Dtype>& coeffs() const { return coeffs_; }
  string op() const { return this->layer_param_.get_msg("eltwise_param").get_enum("operation"); }
 private:
  vector<Dtype> coeffs_;
};
template <typename Dtype>
class SumLayer : public Layer<Dtype> {            // sum_layer.cpp:10-29
  VV_LAYER_BOILER(SumLayer, "SUM")
  virtual int ExactNumBottomBlobs() const { return 1; }
  virtual int ExactNumTopBlobs() const { return 1; }
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
    (*top)[0]->Reshape(bottom[0]->num(), (int)this->layer_param_.get_msg("sum_param").get_num("num_output"), 1, 1);
  }
};
template <typename Dtype>
class MaxMarginLossLayer : public Layer<Dtype> {  // max_margin_loss_layer.cpp:14-51, loss_layer.cpp:13-28
  VV_LAYER_BOILER(MaxMarginLossLayer, "MAX_MARGIN_LOSS")
  virtual int MinBottomBlobs() const { return 2; }
  virtual int MaxBottomBlobs() const { return 3; }
  virtual int MinTopBlobs() const { return 1; }
  virtual int MaxTopBlobs() const { return 2; }
  virtual bool AutoTopBlobs() const { return true; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  // third bottom -> weight of one loss term: the value itself (use_direct_weight) or the id_to_weight_file entry,
  // 0 for ids the file does not list (std::map::operator[], max_margin_loss_layer.cpp:90-96)
  float WeightOf(float third_bottom_value) const;
  virtual ~MaxMarginLossLayer();
 private:
  std::map<int, float> video_id_to_weight_;
  bool use_direct_weight_ = false;
  void* weight_dev_ = nullptr; int weight_count_ = 0;    // per-term weights of the last forward (third bottom)
};

// VIDEO_SHOT_WINDOW_TEST_DATA (video_shot_window_test_data_layer.cpp:37-265): one record per item,
// channels = context frames (then positives, negatives -- not present in the windows built here);
// second top = video id.  Produces table rows of the context frames.
template <typename Dtype>
class VideoShotWindowTestDataLayer : public Layer<Dtype> {
  VV_LAYER_BOILER(VideoShotWindowTestDataLayer, "VIDEO_SHOT_WINDOW_TEST_DATA")
  virtual int ExactNumBottomBlobs() const { return 0; }
  virtual int MinTopBlobs() const { return 1; }
  virtual int MaxTopBlobs() const { return 2; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {}
  void NextBatch(vector<int32_t>* rows, vector<int32_t>* video_ids);    // rows [B][k]
  int batch_size() const { return batch_size_; }
  // channels of the top: context words, then the included positives, then the included negatives
  int channels() const { return dataset_->win_k + positive_size_ + negative_size_; }
  int context_size() const { return channels(); }
  const shared_ptr<VideoDataset>& dataset() const { return dataset_; }
 private:
  shared_ptr<VideoDataset> dataset_;
  int batch_size_ = 0, positive_size_ = 0, negative_size_ = 0;
  size_t cursor_ = 0;
};
// RETRIEVAL_STATS (retrieval_stats_layer.cpp:19-90): tops = mean AP, hit@1, hit@5
template <typename Dtype>
class RetrievalStatsLayer : public Layer<Dtype> {
  VV_LAYER_BOILER(RetrievalStatsLayer, "RETRIEVAL_STATS")
  virtual int ExactNumBottomBlobs() const { return 2; }
  virtual int ExactNumTopBlobs() const { return 3; }
  virtual void LayerSetUp(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  virtual void Reshape(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top);
  const vector<int32_t>& map_ids() const { return map_ids_; }
  const vector<int32_t>& map_cls() const { return map_cls_; }
  bool exclude_same_video_shots() const { return this->layer_param_.get_msg("retrieval_stats_param").get_bool("exclude_same_video_shots"); }
 private:
  vector<int32_t> map_ids_, map_cls_;
};

// layer_factory.cpp:177-309: the 13 hot-path types (+SPLIT); LOG(FATAL) on anything else
template <typename Dtype>
Layer<Dtype>* GetLayer(const LayerParameter& param);

}  // namespace caffe
