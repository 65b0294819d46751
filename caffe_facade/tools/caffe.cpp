// caffe.cpp -- the reference's command line tool (tools/caffe.cpp:18-30,66-287) for the MI355X videovec path.
//   caffe train --solver=<solver.prototxt> [--weights=<.caffemodel>] [--snapshot=<.solverstate>] [--gpu=N]
//   caffe test  --model=<net.prototxt> --weights=<.caffemodel> [--iterations=50] [--gpu=N]     (TEST-phase net)
//   caffe time  --model=<net.prototxt> [--iterations=50] [--gpu=N]      (TRAIN-phase net; the graph runs as a fused
//               plan, so the per-layer lines of the reference become per-kernel lines)
//   caffe device_query --gpu=N
// Extra flags of this build: --precision=f16|bf16, --log_file=<path> (what GLOG_log_dir gives the
// reference's train script, projects/videovec_embedding/train_mednet_embedding.sh:6).
#include <cstring>
#include <ctime>
#include <map>
#include <sstream>

#include "caffe/solver.hpp"

using namespace caffe;

static std::map<std::string, std::string> g_flags;
static std::string flag(const char* name, const char* def = "") { auto it = g_flags.find(name); return it == g_flags.end() ? def : it->second; }

static int train() {
  CHECK_GT(flag("solver").size(), 0u) << "Need a solver definition to train.";
  CHECK(!flag("snapshot").size() || !flag("weights").size()) << "Give a snapshot to resume training or weights to finetune but not both.";
  SolverParameter solver_param("SolverParameter");
  pl::ReadProtoFromTextFileOrDie(flag("solver"), &solver_param);
  int gpu = atoi(flag("gpu", "-1").c_str());
  if (gpu < 0 && solver_param.get_enum("solver_mode") == "GPU") gpu = (int)solver_param.get_int("device_id");
  CHECK_GE(gpu, 0) << "solver_mode: CPU is not available: this build is the GPU path only";
  LOG(INFO) << "Use GPU with device ID " << gpu;
  Caffe::SetDevice(gpu);
  Caffe::set_mode(Caffe::GPU);
  if (flag("precision").size()) Caffe::set_precision(flag("precision"));
  LOG(INFO) << "Starting Optimization";
  shared_ptr<Solver<float> > solver(GetSolver<float>(solver_param));
  if (flag("snapshot").size()) {
    LOG(INFO) << "Resuming from " << flag("snapshot");
    solver->Solve(flag("snapshot"));
  } else if (flag("weights").size()) {
    LOG(INFO) << "Finetuning from " << flag("weights");
    solver->net()->CopyTrainedLayersFrom(flag("weights"));
    solver->Solve();
  } else {
    solver->Solve();
  }
  LOG(INFO) << "Optimization Done.";
  return 0;
}

// tools/caffe.cpp:68-76
static int device_query() {
  const int gpu = atoi(flag("gpu", "-1").c_str());
  CHECK_GT(gpu, -1) << "Need a device ID to query.";
  LOG(INFO) << "Querying device ID = " << gpu;
  char buf[2048];
  CHECK_EQ(vv_device_query(gpu, buf, sizeof(buf)), 0) << vv_last_error();
  std::istringstream in(buf);
  for (std::string line; std::getline(in, line);) LOG(INFO) << line;
  return 0;
}

// tools/caffe.cpp:127-189
static int test() {
  CHECK_GT(flag("model").size(), 0u) << "Need a model definition to score.";
  CHECK_GT(flag("weights").size(), 0u) << "Need model weights to score.";
  const int gpu = atoi(flag("gpu", "0").c_str());
  CHECK_GE(gpu, 0) << "Use CPU. -- not available: this build is the GPU path only";
  LOG(INFO) << "Use GPU with device ID " << gpu;
  Caffe::SetDevice(gpu);
  Caffe::set_mode(Caffe::GPU);
  if (flag("precision").size()) Caffe::set_precision(flag("precision"));
  Caffe::set_phase(Caffe::TEST);
  Net<float> caffe_net(flag("model"), Caffe::TEST);
  caffe_net.CopyTrainedLayersFrom(flag("weights"));
  const int iterations = atoi(flag("iterations", "50").c_str());
  LOG(INFO) << "Running for " << iterations << " iterations.";
  vector<Blob<float>*> bottom_vec;
  vector<int> test_score_output_id;
  vector<float> test_score;
  float loss = 0;
  for (int i = 0; i < iterations; ++i) {
    float iter_loss;
    const vector<Blob<float>*>& result = caffe_net.Forward(bottom_vec, &iter_loss);
    loss += iter_loss;
    int idx = 0;
    for (size_t j = 0; j < result.size(); ++j) {
      const float* result_vec = result[j]->cpu_data();
      for (int k = 0; k < result[j]->count(); ++k, ++idx) {
        const float score = result_vec[k];
        if (i == 0) { test_score.push_back(score); test_score_output_id.push_back((int)j); }
        else test_score[idx] += score;
        LOG(INFO) << "Batch " << i << ", " << caffe_net.blob_names()[caffe_net.output_blob_indices()[j]] << " = " << score;
      }
    }
  }
  loss /= iterations;
  LOG(INFO) << "Loss: " << loss;
  for (size_t i = 0; i < test_score.size(); ++i) {
    const int blob_index = caffe_net.output_blob_indices()[test_score_output_id[i]];
    const float loss_weight = caffe_net.blob_loss_weights()[blob_index];
    std::ostringstream loss_msg_stream;
    const float mean_score = test_score[i] / iterations;
    if (loss_weight) loss_msg_stream << " (* " << loss_weight << " = " << loss_weight * mean_score << " loss)";
    LOG(INFO) << caffe_net.blob_names()[blob_index] << " = " << mean_score << loss_msg_stream.str();
  }
  return 0;
}

// tools/caffe.cpp:193-265.  The reference times every layer's Forward and Backward separately; here the graph is one
// fused plan, so what can be timed are its kernels (device time from events on their dispatch packets) and the whole
// forward-backward(-update) iteration (host clock around the loop, device drained at both ends).
static int time_net() {
  CHECK_GT(flag("model").size(), 0u) << "Need a model definition to time.";
  const int gpu = atoi(flag("gpu", "0").c_str());
  CHECK_GE(gpu, 0) << "Use CPU. -- not available: this build is the GPU path only";
  LOG(INFO) << "Use GPU with device ID " << gpu;
  Caffe::SetDevice(gpu);
  Caffe::set_mode(Caffe::GPU);
  if (flag("precision").size()) Caffe::set_precision(flag("precision"));
  Caffe::set_phase(Caffe::TRAIN);
  Net<float> caffe_net(flag("model"), Caffe::TRAIN);
  LOG(INFO) << "Performing Forward";
  float initial_loss;
  caffe_net.Forward(vector<Blob<float>*>(), &initial_loss);
  LOG(INFO) << "Initial loss: " << initial_loss;
  LOG(INFO) << "Performing Backward";       // part of the same fused pass
  const int iterations = atoi(flag("iterations", "50").c_str());
  LOG(INFO) << "*** Benchmark begins ***";
  LOG(INFO) << "Testing for " << iterations << " iterations.";
  vv_ctx* ctx = Caffe::ctx();
  vector<Blob<float>*> bottom;
  caffe_net.set_loss_needed(false);
  for (int j = 0; j < 3; ++j) { caffe_net.ForwardBackward(bottom); caffe_net.Update(); }
  CHECK_EQ(vv_synchronize(ctx), 0);
  CHECK_EQ(vv_profile_enable(ctx, iterations >= 10 ? iterations / 10 : 1), 0);
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int j = 0; j < iterations; ++j) { caffe_net.ForwardBackward(bottom); caffe_net.Update(); }
  CHECK_EQ(vv_synchronize(ctx), 0);
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double total_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
  const char* kernels[] = {"dedup", "fwd_gemm", "score_loss", "segsum", "wgrad_gemm", "reduce", "sgd"};
  for (const char* k : kernels) {
    double ms = 0; int64_t n = 0;
    CHECK_EQ(vv_profile_get(ctx, k, &ms, &n), 0);
    if (n) LOG(INFO) << k << "\tkernel: " << ms << " milliseconds (average of " << n << " launches).";
  }
  CHECK_EQ(vv_profile_enable(ctx, 0), 0);
  LOG(INFO) << "Forward-backward-update iteration: " << total_ms / iterations << " milliseconds (includes the host-side sampler).";
  LOG(INFO) << "Total Time: " << total_ms << " milliseconds.";
  LOG(INFO) << "*** Benchmark ends ***";
  return 0;
}

int main(int argc, char** argv) {
  std::string action;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    if (a.compare(0, 1, "-") == 0) {
      a = a.substr(a.find_first_not_of('-'));
      const size_t eq = a.find('=');
      if (eq == std::string::npos) { if (i + 1 < argc && argv[i + 1][0] != '-') g_flags[a] = argv[++i]; else g_flags[a] = "true"; }
      else g_flags[a.substr(0, eq)] = a.substr(eq + 1);
    } else if (action.empty()) action = a;
  }
  if (flag("log_file").size()) SetLogFile(flag("log_file"));
  if (action == "train") return train();
  if (action == "test") return test();
  if (action == "time") return time_net();
  if (action == "device_query") return device_query();
  fprintf(stderr, "caffe: command line brew\nusage: caffe <command> <args>\n\ncommands:\n"
                  "  train           train or finetune a model\n"
                  "  test            score a model\n"
                  "  device_query    show GPU diagnostic information\n"
                  "  time            benchmark model execution time\n");
  return 1;
}
