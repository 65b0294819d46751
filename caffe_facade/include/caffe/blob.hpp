// blob.hpp -- host-side Blob with the reference's interface (include/caffe/blob.hpp:18-141).
// Device memory lives inside the vv context; a Blob here is the host view that snapshot / inspection
// code reads (cpu_data / cpu_diff), exactly what Net::ToProto, CopyTrainedLayersFrom and the solver
// history need.  Only float is instantiated (the reference trains Solver<float>, tools/caffe.cpp:107).
#pragma once
#include "caffe/common.hpp"
#include "caffe/proto_lite.hpp"

namespace caffe {

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
  const Dtype* cpu_data() const { return data_ ? data_->data() : nullptr; }
  const Dtype* cpu_diff() const { return diff_ ? diff_->data() : nullptr; }
  Dtype* mutable_cpu_data() { return data_ ? data_->data() : nullptr; }
  Dtype* mutable_cpu_diff() { return diff_ ? diff_->data() : nullptr; }
  void Update();                                   // data -= diff   (blob.cpp:112-136)
  Dtype asum_data() const;
  Dtype asum_diff() const;
  void CopyFrom(const Blob<Dtype>& source, bool copy_diff = false, bool reshape = false);
  void FromProto(const pl::Message& proto);        // blob.cpp:242-268
  void ToProto(pl::Message* proto, bool write_diff = false) const;   // blob.cpp:270-322
  void ShareData(const Blob& other) { CHECK_EQ(count_, other.count()); data_ = other.data_; }
  void ShareDiff(const Blob& other) { CHECK_EQ(count_, other.count()); diff_ = other.diff_; }
 private:
  shared_ptr<vector<Dtype> > data_, diff_;
  int num_, channels_, height_, width_, count_;
};

}  // namespace caffe
