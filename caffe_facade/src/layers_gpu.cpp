// layers_gpu.cpp -- Forward_gpu / Backward_gpu of the hot-path layer classes (the reference's *.cu files of the same
// layers), each a call or two into the per-layer operators of the C ABI (vv_op_*, include/videovec.h), plus the
// non-virtual Layer::Forward / Layer::Backward wrappers (include/caffe/layer.hpp:409-458).
//
// These run when a graph is executed layer by layer (Net::ForwardFromTo / BackwardFromTo).  The recognised videovec
// graphs run as one fused plan instead and never come through here.
#include "caffe/layer.hpp"

namespace caffe {

namespace {
inline vv_ctx* X() { return Caffe::ctx(); }
template <typename Dtype> int Inner(const Blob<Dtype>* b) { return b->count() / (b->num() ? b->num() : 1); }
}  // namespace

// ---------------------------------------------------------------------------------------------- Layer
template <typename Dtype>
void Layer<Dtype>::Forward_cpu(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>*) {
  LOG(FATAL) << "Layer " << layer_param_.get_str("name") << " (" << type() << "): this build has no CPU execution path "
             << "(the reference's CPU path exists only as the test oracle)";
}
template <typename Dtype>
void Layer<Dtype>::Backward_cpu(const vector<Blob<Dtype>*>&, const vector<bool>&, vector<Blob<Dtype>*>*) {
  LOG(FATAL) << "Layer " << layer_param_.get_str("name") << " (" << type() << "): this build has no CPU execution path";
}

// layer.hpp:409-428: run the layer, then the layer's loss = sum over its loss tops of loss_weight * top value (the
// reference computes dot(top.data, top.diff) with the diff pre-filled by SetLossWeights; the loss tops of this path are scalars)
template <typename Dtype>
Dtype Layer<Dtype>::Forward(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  CHECK(Caffe::mode() == Caffe::GPU) << "Unknown caffe mode.";
  Forward_gpu(bottom, top);
  Dtype loss = 0;
  for (size_t t = 0; t < top->size(); ++t) {
    const Dtype w = this->loss((int)t);
    if (w == 0) continue;
    const Blob<Dtype>* b = (*top)[t];
    const Dtype* d = b->cpu_data();
    for (int i = 0; i < b->count(); ++i) loss += w * d[i];
  }
  return loss;
}

// layer.hpp:431-445, plus the accumulation of fan-in diffs that the reference delegates to inserted SPLIT layers
template <typename Dtype>
void Layer<Dtype>::Backward(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  CHECK(Caffe::mode() == Caffe::GPU) << "Unknown caffe mode.";
  vector<shared_ptr<Blob<Dtype> > > scratch(bottom->size());
  vector<Blob<Dtype>*> targets = *bottom;
  for (size_t i = 0; i < bottom->size(); ++i) {
    if (i >= accumulate_bottom_.size() || !accumulate_bottom_[i] || !propagate_down[i]) continue;
    scratch[i].reset(new Blob<Dtype>());
    scratch[i]->ReshapeLike(*(*bottom)[i]);
    scratch[i]->ShareData(*(*bottom)[i]);          // same data, a diff of its own
    targets[i] = scratch[i].get();
  }
  Backward_gpu(top, propagate_down, &targets);
  for (size_t i = 0; i < bottom->size(); ++i)
    if (scratch[i]) VV_CHECK(vv_op_axpby(X(), (*bottom)[i]->count(), 1.f, scratch[i]->gpu_diff(), 1.f, (*bottom)[i]->mutable_gpu_diff()));
}

// ---------------------------------------------------------------------------------------------- data layers
// BasePrefetchingDataLayer::Forward_gpu (base_data_layer.cu:7-21): the prefetched batch becomes the top blob.  Here the
// batch is row indices; the top blob (B, C+Nn, F, 1) is gathered out of the HBM-resident table.
template <typename Dtype>
void VideoSampledShotsDataLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>* top) {
  NextBatch(&fw_idx_, &fw_last_, &fw_label_);
  Blob<Dtype>* data = (*top)[0];
  VV_CHECK(vv_op_gather_rows(X(), fw_idx_.data(), (int64_t)fw_idx_.size(), data->mutable_gpu_data()));
  // quirk Q1 (…data_layer.cpp:492): a same-video negative keeps the LAST feature of whatever the slot held before
  const int F = feature_size_;
  for (size_t i = 0; i < fw_idx_.size(); ++i) {
    if (fw_last_[i] == fw_idx_[i]) continue;
    float v = 0.f;
    if (fw_last_[i] >= 0) { vector<float> row((size_t)F); int32_t r = fw_last_[i]; VV_CHECK(vv_table_get(X(), &r, 1, row.data())); v = row[F - 1]; }
    VV_CHECK(vv_dev_upload(X(), data->mutable_gpu_data() + i * (size_t)F + F - 1, &v, sizeof(float)));
  }
  if (top->size() > 1) {
    Dtype* l = (*top)[1]->mutable_cpu_data();
    for (int i = 0; i < batch_size_; ++i) l[i] = (Dtype)fw_label_[i];                     // …data_layer.cpp:879
  }
}
template <typename Dtype>
void VideoSampledShotsDataLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>&, vector<Blob<Dtype>*>*) {}

template <typename Dtype>
void VideoShotWindowTestDataLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>&, vector<Blob<Dtype>*>* top) {
  vector<int32_t> rows, vids;
  NextBatch(&rows, &vids);
  VV_CHECK(vv_op_gather_rows(X(), rows.data(), (int64_t)rows.size(), (*top)[0]->mutable_gpu_data()));
  if (top->size() > 1) {
    Dtype* l = (*top)[1]->mutable_cpu_data();
    for (int i = 0; i < batch_size_; ++i) l[i] = (Dtype)vids[i];                            // …test_data_layer.cpp:262
  }
}
template <typename Dtype>
void VideoShotWindowTestDataLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>&, vector<Blob<Dtype>*>*) {}

// ---------------------------------------------------------------------------------------------- SLICE / CONCAT
// slice_layer.cu:10-64.  dim 0: consecutive blocks of items; dim 1: per item a block of channels.
template <typename Dtype>
void SliceLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int dim = (int)this->layer_param_.get_msg("slice_param").get_int("slice_dim");
  const Blob<Dtype>* b = bottom[0];
  const int64_t inner = b->height() * b->width();
  int64_t off = 0;
  for (size_t i = 0; i < top->size(); ++i) {
    Blob<Dtype>* t = (*top)[i];
    if (dim == 0) { VV_CHECK(vv_op_copy2d(X(), b->gpu_data() + off, t->count(), t->mutable_gpu_data(), t->count(), 1, t->count(), 0)); off += t->count(); }
    else {
      const int64_t cols = t->channels() * inner;
      VV_CHECK(vv_op_copy2d(X(), b->gpu_data() + off, b->channels() * inner, t->mutable_gpu_data(), cols, b->num(), cols, 0));
      off += cols;
    }
  }
}
template <typename Dtype>
void SliceLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0]) return;
  const int dim = (int)this->layer_param_.get_msg("slice_param").get_int("slice_dim");
  Blob<Dtype>* b = (*bottom)[0];
  const int64_t inner = b->height() * b->width();
  int64_t off = 0;
  for (size_t i = 0; i < top.size(); ++i) {
    const Blob<Dtype>* t = top[i];
    if (dim == 0) { VV_CHECK(vv_op_copy2d(X(), t->gpu_diff(), t->count(), b->mutable_gpu_diff() + off, t->count(), 1, t->count(), 0)); off += t->count(); }
    else {
      const int64_t cols = t->channels() * inner;
      VV_CHECK(vv_op_copy2d(X(), t->gpu_diff(), cols, b->mutable_gpu_diff() + off, b->channels() * inner, b->num(), cols, 0));
      off += cols;
    }
  }
}
// concat_layer.cu:10-75
template <typename Dtype>
void ConcatLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int dim = (int)this->layer_param_.get_msg("concat_param").get_int("concat_dim");
  Blob<Dtype>* t = (*top)[0];
  const int64_t inner = t->height() * t->width();
  int64_t off = 0;
  for (size_t i = 0; i < bottom.size(); ++i) {
    const Blob<Dtype>* b = bottom[i];
    if (dim == 0) { VV_CHECK(vv_op_copy2d(X(), b->gpu_data(), b->count(), t->mutable_gpu_data() + off, b->count(), 1, b->count(), 0)); off += b->count(); }
    else {
      const int64_t cols = b->channels() * inner;
      VV_CHECK(vv_op_copy2d(X(), b->gpu_data(), cols, t->mutable_gpu_data() + off, t->channels() * inner, t->num(), cols, 0));
      off += cols;
    }
  }
}
template <typename Dtype>
void ConcatLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  const int dim = (int)this->layer_param_.get_msg("concat_param").get_int("concat_dim");
  const Blob<Dtype>* t = top[0];
  const int64_t inner = t->height() * t->width();
  int64_t off = 0;
  for (size_t i = 0; i < bottom->size(); ++i) {
    Blob<Dtype>* b = (*bottom)[i];
    const int64_t cols = dim == 0 ? b->count() : b->channels() * inner;
    if (propagate_down[i]) {
      if (dim == 0) VV_CHECK(vv_op_copy2d(X(), t->gpu_diff() + off, b->count(), b->mutable_gpu_diff(), b->count(), 1, b->count(), 0));
      else VV_CHECK(vv_op_copy2d(X(), t->gpu_diff() + off, t->channels() * inner, b->mutable_gpu_diff(), cols, t->num(), cols, 0));
    }
    off += cols;
  }
}

// ---------------------------------------------------------------------------------------------- FLATTEN / SPLIT
// flatten_layer.cpp:18-30: top and bottom share memory
template <typename Dtype>
void FlattenLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) { (*top)[0]->ShareData(*bottom[0]); }
template <typename Dtype>
void FlattenLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (propagate_down[0]) VV_CHECK(vv_op_copy2d(X(), top[0]->gpu_diff(), top[0]->count(), (*bottom)[0]->mutable_gpu_diff(), top[0]->count(), 1, top[0]->count(), 0));
}
// split_layer.cu:10-33: tops share the bottom's data; the bottom's diff is the sum of the tops' diffs
template <typename Dtype>
void SplitLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  for (size_t i = 0; i < top->size(); ++i) (*top)[i]->ShareData(*bottom[0]);
}
template <typename Dtype>
void SplitLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0]) return;
  const int64_t n = (*bottom)[0]->count();
  for (size_t i = 0; i < top.size(); ++i)
    VV_CHECK(vv_op_copy2d(X(), top[i]->gpu_diff(), n, (*bottom)[0]->mutable_gpu_diff(), n, 1, n, i > 0));
}

// ---------------------------------------------------------------------------------------------- INNER_PRODUCT
// inner_product_layer.cu:12-59 on the parameters the process context holds (the layer's blobs_ are their host mirror)
template <typename Dtype>
void InnerProductLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  VV_CHECK(vv_op_inner_product(X(), bottom[0]->gpu_data(), bottom[0]->num(), (*top)[0]->mutable_gpu_data()));
}
template <typename Dtype>
void InnerProductLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>*) {
  CHECK(!propagate_down[0]) << "INNER_PRODUCT: the gradient w.r.t. the bottom blob is not built (the fc layer of the videovec "
                               "path reads the data layer, which needs none)";
  VV_CHECK(vv_op_inner_product_bwd(X(), top[0]->gpu_diff(), top[0]->num(),
                                   (float)this->layer_param_.get_msg("inner_product_param").get_num("regularization")));
}

// ---------------------------------------------------------------------------------------------- neuron layers
template <typename Dtype>
void ReLULayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const float slope = (float)this->layer_param_.get_msg("relu_param").get_num("negative_slope");
  VV_CHECK(vv_op_relu(X(), bottom[0]->count(), bottom[0]->gpu_data(), (*top)[0]->mutable_gpu_data(), slope));
}
template <typename Dtype>
void ReLULayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0]) return;
  const float slope = (float)this->layer_param_.get_msg("relu_param").get_num("negative_slope");
  VV_CHECK(vv_op_relu_bwd(X(), top[0]->count(), (*bottom)[0]->gpu_data(), top[0]->gpu_diff(), (*bottom)[0]->mutable_gpu_diff(), slope));
}
template <typename Dtype>
DropoutLayer<Dtype>::~DropoutLayer() { if (mask_ && Caffe::has_ctx()) vv_dev_free(Caffe::ctx(), mask_); }
template <typename Dtype>
void DropoutLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int n = bottom[0]->count();
  const float ratio = (float)this->layer_param_.get_msg("dropout_param").get_num("dropout_ratio");
  if (Caffe::phase() != Caffe::TRAIN) {                                                    // dropout_layer.cu:37-39
    if ((*top)[0] != bottom[0]) VV_CHECK(vv_op_copy2d(X(), bottom[0]->gpu_data(), n, (*top)[0]->mutable_gpu_data(), n, 1, n, 0));
    return;
  }
  if (mask_count_ < n) { if (mask_) vv_dev_free(X(), mask_); VV_CHECK(vv_dev_alloc(X(), (size_t)n, &mask_)); mask_count_ = n; }
  const uint64_t seed = (uint64_t)Caffe::random_seed() * 0x9E3779B97F4A7C15ull + calls_++;
  VV_CHECK(vv_op_dropout(X(), n, bottom[0]->gpu_data(), (*top)[0]->mutable_gpu_data(), (uint8_t*)mask_, ratio, seed, 1));
}
template <typename Dtype>
void DropoutLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0]) return;
  const int n = top[0]->count();
  if (Caffe::phase() != Caffe::TRAIN) {      // dropout_layer.cu:68-70: outside TRAIN the diff passes through unchanged
    VV_CHECK(vv_op_axpby(X(), n, 1.f, top[0]->gpu_diff(), 0.f, (*bottom)[0]->mutable_gpu_diff()));
    return;
  }
  const float ratio = (float)this->layer_param_.get_msg("dropout_param").get_num("dropout_ratio");
  VV_CHECK(vv_op_dropout(X(), n, top[0]->gpu_diff(), (*bottom)[0]->mutable_gpu_diff(), (uint8_t*)mask_, ratio, 0, 0));
}
template <typename Dtype>
void NormalizationLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  VV_CHECK(vv_op_normalize(X(), bottom[0]->num(), Inner(bottom[0]), bottom[0]->gpu_data(), (*top)[0]->mutable_gpu_data()));
}
template <typename Dtype>
void NormalizationLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0]) return;
  VV_CHECK(vv_op_normalize_bwd(X(), top[0]->num(), Inner(top[0]), (*bottom)[0]->gpu_data(), top[0]->gpu_diff(), (*bottom)[0]->mutable_gpu_diff()));
}

// ---------------------------------------------------------------------------------------------- ELTWISE / SUM
// eltwise_layer.cu:34-54
template <typename Dtype>
void EltwiseLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int64_t n = (*top)[0]->count();
  Dtype* y = (*top)[0]->mutable_gpu_data();
  if (op() == "SUM") {
    for (size_t i = 0; i < bottom.size(); ++i) VV_CHECK(vv_op_axpby(X(), n, coeffs_[i], bottom[i]->gpu_data(), i ? 1.f : 0.f, y));
  } else if (op() == "PROD") {
    VV_CHECK(vv_op_mul(X(), n, bottom[0]->gpu_data(), bottom[1]->gpu_data(), y, 0));
    for (size_t i = 2; i < bottom.size(); ++i) VV_CHECK(vv_op_mul(X(), n, y, bottom[i]->gpu_data(), y, 0));
  } else LOG(FATAL) << "Unknown elementwise operation.";                                   // eltwise_layer.cpp:70
}
// eltwise_layer.cu:83-119 (stable_prod_grad: the product of the OTHER bottoms times the top diff)
template <typename Dtype>
void EltwiseLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  const int64_t n = top[0]->count();
  for (size_t i = 0; i < bottom->size(); ++i) {
    if (!propagate_down[i]) continue;
    Dtype* dx = (*bottom)[i]->mutable_gpu_diff();
    if (op() == "SUM") VV_CHECK(vv_op_axpby(X(), n, coeffs_[i], top[0]->gpu_diff(), 0.f, dx));
    else {
      bool first = true;
      for (size_t j = 0; j < bottom->size(); ++j) {
        if (j == i) continue;
        if (first) VV_CHECK(vv_op_mul(X(), n, (*bottom)[j]->gpu_data(), top[0]->gpu_diff(), dx, 0));
        else VV_CHECK(vv_op_mul(X(), n, dx, (*bottom)[j]->gpu_data(), dx, 0));
        first = false;
      }
    }
  }
}
// sum_layer.cu:10-55
template <typename Dtype>
void SumLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  VV_CHECK(vv_op_rowsum(X(), bottom[0]->num(), Inner(bottom[0]), bottom[0]->gpu_data(), Inner((*top)[0]), (*top)[0]->mutable_gpu_data()));
}
template <typename Dtype>
void SumLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0]) return;
  VV_CHECK(vv_op_rowsum_bwd(X(), top[0]->num(), Inner((*bottom)[0]), Inner(top[0]), top[0]->gpu_diff(), (*bottom)[0]->mutable_gpu_diff()));
}

// ---------------------------------------------------------------------------------------------- MAX_MARGIN_LOSS
template <typename Dtype>
MaxMarginLossLayer<Dtype>::~MaxMarginLossLayer() { if (weight_dev_ && Caffe::has_ctx()) vv_dev_free(Caffe::ctx(), weight_dev_); }
// max_margin_loss_layer.cpp:53-127 (Forward_cpu; the reference has no GPU version and synchronises here too)
template <typename Dtype>
void MaxMarginLossLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int count = bottom[0]->count();
  CHECK_EQ(bottom[1]->count(), count) << "the two score blobs must hold the same number of terms (quirk Q3)";
  const float* w = nullptr;
  if (bottom.size() > 2) {                                                                 // :82-97 the third bottom -> term weights
    CHECK_EQ(bottom[2]->count(), count);
    vector<float> hw((size_t)count);
    const Dtype* ids = bottom[2]->cpu_data();
    for (int i = 0; i < count; ++i) hw[i] = WeightOf((float)ids[i]);
    if (weight_count_ < count) { if (weight_dev_) vv_dev_free(X(), weight_dev_); VV_CHECK(vv_dev_alloc(X(), (size_t)count * 4, &weight_dev_)); weight_count_ = count; }
    VV_CHECK(vv_dev_upload(X(), weight_dev_, hw.data(), (size_t)count * 4));
    w = (const float*)weight_dev_;
  }
  const pl::Message& mp = this->layer_param_.get_msg("max_margin_loss_param");
  float loss = 0, viol = 0;
  VV_CHECK(vv_op_max_margin(X(), count, bottom[0]->gpu_data(), bottom[1]->gpu_data(), w, (float)mp.get_num("margin"),
                            mp.get_enum("norm") == "L2" ? VV_NORM_L2 : VV_NORM_L1, &loss, &viol));
  (*top)[0]->mutable_cpu_data()[0] = loss;
  if (top->size() > 1) (*top)[1]->mutable_cpu_data()[0] = viol;                            // :123-126
}
// max_margin_loss_layer.cpp:129-214: both bottoms' diffs are always written (quirk Q4)
template <typename Dtype>
void MaxMarginLossLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>& top, const vector<bool>& propagate_down, vector<Blob<Dtype>*>* bottom) {
  if (!propagate_down[0] && !propagate_down[1]) return;
  const int count = (*bottom)[0]->count();
  const pl::Message& mp = this->layer_param_.get_msg("max_margin_loss_param");
  VV_CHECK(vv_op_max_margin_bwd(X(), count, (*bottom)[0]->gpu_data(), (*bottom)[1]->gpu_data(),
                                bottom->size() > 2 ? (const float*)weight_dev_ : nullptr, (float)mp.get_num("margin"),
                                mp.get_enum("norm") == "L2" ? VV_NORM_L2 : VV_NORM_L1, (float)this->loss(0),
                                (*bottom)[0]->mutable_gpu_diff(), (*bottom)[1]->mutable_gpu_diff()));
}

// ---------------------------------------------------------------------------------------------- RETRIEVAL_STATS
// retrieval_stats_layer.cpp:143-355 (Forward_cpu in the reference): the Gram matrix on the GPU, ranking on the host
template <typename Dtype>
void RetrievalStatsLayer<Dtype>::Forward_gpu(const vector<Blob<Dtype>*>& bottom, vector<Blob<Dtype>*>* top) {
  const int n = bottom[0]->num();
  vector<int32_t> vids((size_t)n);
  for (int i = 0; i < n; ++i) vids[i] = (int32_t)bottom[1]->cpu_data()[i];
  float m = 0, h1 = 0, h5 = 0;
  VV_CHECK(vv_retrieval_stats(X(), bottom[0]->cpu_data(), n, Inner(bottom[0]), vids.data(), map_ids_.data(), map_cls_.data(),
                              (int)map_ids_.size(), exclude_same_video_shots() ? 1 : 0, &m, &h1, &h5));
  const float v[3] = {m, h1, h5};
  for (int t = 0; t < 3; ++t) (*top)[t]->mutable_cpu_data()[0] = v[t];
}
template <typename Dtype>
void RetrievalStatsLayer<Dtype>::Backward_gpu(const vector<Blob<Dtype>*>&, const vector<bool>&, vector<Blob<Dtype>*>*) {}

template class Layer<float>;
template class VideoSampledShotsDataLayer<float>;
template class VideoShotWindowTestDataLayer<float>;
template class SliceLayer<float>;
template class ConcatLayer<float>;
template class FlattenLayer<float>;
template class SplitLayer<float>;
template class InnerProductLayer<float>;
template class ReLULayer<float>;
template class DropoutLayer<float>;
template class NormalizationLayer<float>;
template class EltwiseLayer<float>;
template class SumLayer<float>;
template class MaxMarginLossLayer<float>;
template class RetrievalStatsLayer<float>;

}  // namespace caffe
