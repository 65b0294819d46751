// common.hpp -- process-wide runtime state and glog-style logging for the videovec facade.
// Mirrors the reference's include/caffe/common.hpp:70-147 (class Caffe: mode / phase / device / seed;
// a process-global singleton, so one process drives one GPU).  Caffe::GPU means HIP on gfx950.
#pragma once
#include <cstdlib>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "../../../include/videovec.h"

namespace caffe {

using std::string;
using std::vector;
using std::shared_ptr;

// ---- logging (format of glog: "I0102 15:04:05.123456 tid file:line] msg"; FATAL aborts like
// the reference's CHECK / LOG(FATAL) convention, e.g. src/caffe/net.cpp:353)
class LogMessage {
 public:
  LogMessage(const char* file, int line, char sev);
  ~LogMessage();
  std::ostream& stream() { return ss_; }
 private:
  std::ostringstream ss_;
  char sev_;
};
struct LogVoidify { void operator&(std::ostream&) {} };
void SetLogFile(const std::string& path);    // GLOG_log_dir equivalent: also tee to a file

#define LOG(sev) ::caffe::LogMessage(__FILE__, __LINE__, #sev[0]).stream()
#define CHECK(c) (c) ? (void)0 : ::caffe::LogVoidify() & ::caffe::LogMessage(__FILE__, __LINE__, 'F').stream() << "Check failed: " #c " "
#define CHECK_OP(a, b, op) CHECK((a) op (b)) << "(" << (a) << " vs. " << (b) << ") "
#define CHECK_EQ(a, b) CHECK_OP(a, b, ==)
#define CHECK_NE(a, b) CHECK_OP(a, b, !=)
#define CHECK_LE(a, b) CHECK_OP(a, b, <=)
#define CHECK_LT(a, b) CHECK_OP(a, b, <)
#define CHECK_GE(a, b) CHECK_OP(a, b, >=)
#define CHECK_GT(a, b) CHECK_OP(a, b, >)
#define VV_CHECK(call) do { int rc_ = (call); CHECK(rc_ == 0) << #call << " failed (" << rc_ << "): " << vv_last_error(); } while (0)

// gflags defined in the reference's layer files (video_sampled_shots_data_layer.cpp:20, retrieval_stats_layer.cpp:16);
// `caffe` and `extract_features` accept --max_tries_for_negs=N and --num_classes=N
extern int FLAGS_max_tries_for_negs;
extern int FLAGS_num_classes;

class Caffe {
 public:
  enum Brew { CPU, GPU };
  enum Phase { TRAIN, TEST };
  static Caffe& Get();
  static Brew mode() { return Get().mode_; }
  static Phase phase() { return Get().phase_; }
  // There is no CPU execution path in this build: set_mode(CPU) is fatal (the reference's CPU
  // path is restated only as the test oracle).
  static void set_mode(Brew mode);
  static void set_phase(Phase phase) { Get().phase_ = phase; }
  static void SetDevice(const int device_id);              // common.cpp:127-145
  static void set_random_seed(const unsigned int seed) { Get().seed_ = seed; }
  static unsigned int random_seed() { return Get().seed_; }
  static int device() { return Get().device_; }
  // Data-parallel job shape: one process per GPU (no counterpart in the reference, which is single-device).  Read
  // once from the environment a launcher such as torch.distributed.run sets: WORLD_SIZE, RANK, LOCAL_RANK (VV_WORLD_SIZE
  // / VV_RANK / VV_LOCAL_RANK take precedence); job_id() keeps the shared-memory and id-file names of concurrent jobs apart.
  static int world() { return Get().world_; }
  static int rank() { return Get().rank_; }
  static int local_rank() { return Get().local_rank_; }
  static const std::string& job_id() { return Get().job_id_; }
  // MFMA operand precision of the context ("f16" default, "bf16"); env VV_PREC overrides
  static void set_precision(const std::string& p);
  // The HIP context of this process (created on first use)
  static vv_ctx* ctx();
  static bool has_ctx();
  // what ctx() returns while set (a TEST net's own context during its layer-by-layer pass); returns the previous one
  static vv_ctx* set_current_ctx(vv_ctx* c);
  // A net with a context of its own (TEST / extraction nets) registers it for the context's lifetime: device memory
  // keeps using the context (hence the stream) it was allocated under for as long as that context lives.
  static void register_ctx(vv_ctx* c);
  static void unregister_ctx(vv_ctx* c);
  static bool is_live(vv_ctx* c);
  static void Reset();                                     // destroy the context (tests)
 private:
  Caffe();
  int world_ = 1, rank_ = 0, local_rank_ = 0;
  std::string job_id_;
  Brew mode_ = GPU;
  Phase phase_ = TRAIN;
  int device_ = 0;
  int prec_ = VV_PREC_F16;
  unsigned int seed_ = 1701;
  vv_ctx* ctx_ = nullptr;
  vv_ctx* current_ = nullptr;
  std::vector<vv_ctx*> live_;
};

}  // namespace caffe
