// blob.hpp -- Blob and SyncedMemory with the reference's interface (include/caffe/blob.hpp:18-141,
// include/caffe/syncedmem.hpp:40-68).  A blob's data and diff live in a SyncedMemory each: a host copy and a device
// copy (HIP memory of the process context, through vv_dev_*), allocated on first touch and synchronised on access by
// the same head state machine (UNINITIALIZED / HEAD_AT_CPU / HEAD_AT_GPU / SYNCED).  The fused training plan keeps
// its activations inside the context and only the host side of a blob is touched; the layer-by-layer executor
// (Layer::Forward_gpu / Backward_gpu) works on the device side.  Only float is instantiated (the reference trains
// Solver<float>, tools/caffe.cpp:107).
#pragma once
#include "caffe/common.hpp"
#include "caffe/proto_lite.hpp"

namespace caffe {

class SyncedMemory {            // syncedmem.hpp:40-68
 public:
  explicit SyncedMemory(size_t size) : size_(size) {}
  ~SyncedMemory();
  enum SyncedHead { UNINITIALIZED, HEAD_AT_CPU, HEAD_AT_GPU, SYNCED };
  const void* cpu_data();
  void* mutable_cpu_data();
  const void* gpu_data();
  void* mutable_gpu_data();
  SyncedHead head() const { return head_; }
  size_t size() const { return size_; }
 private:
  void to_cpu();
  void to_gpu();
  vv_ctx* ctx();                   // the context the device copy lives under (the current one when it is allocated)
  vv_ctx* owner_ = nullptr;
  std::vector<char> host_;
  void* dev_ = nullptr;
  size_t size_;
  SyncedHead head_ = UNINITIALIZED;
};

template <typename Dtype>
class Blob {
 public:
  Blob() : num_(0), channels_(0), height_(0), width_(0), count_(0) {}
  Blob(const int num, const int channels, const int height, const int width) { Reshape(num, channels, height, width); }
  void Reshape(const int num, const int channels, const int height, const int width);
  void ReshapeLike(const Blob& other) { Reshape(other.num(), other.channels(), other.height(), other.width()); }
  inline int num() const { return num_; }
  inline int channels() const { return channels_; }
  inline int height() const { return height_; }
  inline int width() const { return width_; }
  inline int count() const { return count_; }
  inline int offset(const int n, const int c = 0, const int h = 0, const int w = 0) const {
    return ((n * channels_ + c) * height_ + h) * width_ + w;
  }
  inline Dtype data_at(const int n, const int c, const int h, const int w) const { return cpu_data()[offset(n, c, h, w)]; }
  inline Dtype diff_at(const int n, const int c, const int h, const int w) const { return cpu_diff()[offset(n, c, h, w)]; }
  const Dtype* cpu_data() const { return data_ ? (const Dtype*)data_->cpu_data() : nullptr; }
  const Dtype* cpu_diff() const { return diff_ ? (const Dtype*)diff_->cpu_data() : nullptr; }
  Dtype* mutable_cpu_data() { return data_ ? (Dtype*)data_->mutable_cpu_data() : nullptr; }
  Dtype* mutable_cpu_diff() { return diff_ ? (Dtype*)diff_->mutable_cpu_data() : nullptr; }
  const Dtype* gpu_data() const { return data_ ? (const Dtype*)data_->gpu_data() : nullptr; }
  const Dtype* gpu_diff() const { return diff_ ? (const Dtype*)diff_->gpu_data() : nullptr; }
  Dtype* mutable_gpu_data() { return data_ ? (Dtype*)data_->mutable_gpu_data() : nullptr; }
  Dtype* mutable_gpu_diff() { return diff_ ? (Dtype*)diff_->mutable_gpu_data() : nullptr; }
  void Update();                                   // data -= diff   (blob.cpp:112-136)
  Dtype asum_data() const;
  Dtype asum_diff() const;
  void CopyFrom(const Blob<Dtype>& source, bool copy_diff = false, bool reshape = false);
  void FromProto(const pl::Message& proto);        // blob.cpp:242-268
  void ToProto(pl::Message* proto, bool write_diff = false) const;   // blob.cpp:270-322
  void ShareData(const Blob& other) { CHECK_EQ(count_, other.count()); data_ = other.data_; }
  void ShareDiff(const Blob& other) { CHECK_EQ(count_, other.count()); diff_ = other.diff_; }
 private:
  shared_ptr<SyncedMemory> data_, diff_;
  int num_, channels_, height_, width_, count_;
  int capacity_ = 0;
};

}  // namespace caffe
