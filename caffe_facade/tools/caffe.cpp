// caffe.cpp -- the reference's command line tool (tools/caffe.cpp:18-30,66-287) for the MI355X videovec path.
//   caffe train --solver=<solver.prototxt> [--weights=<.caffemodel>] [--snapshot=<.solverstate>] [--gpu=N]
//   caffe test  --model=<net.prototxt> --weights=<.caffemodel> [--iterations=50] [--gpu=N]     (TEST-phase net)
//   caffe time  --model=<net.prototxt> [--iterations=50] [--gpu=N]      (TRAIN-phase net; the graph runs as a fused
//               plan, so the per-layer lines of the reference become per-kernel lines)
//   caffe device_query --gpu=N
// Layer-file gflags of the reference: --max_tries_for_negs=N (video_sampled_shots_data_layer.cpp:20), --num_classes=N
// (retrieval_stats_layer.cpp:16).  Data-parallel: launch one process per GPU with WORLD_SIZE / RANK / LOCAL_RANK set
// (python -m torch.distributed.run --no-python ... caffe train ..., or any launcher); --gpu defaults to LOCAL_RANK then.
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
  if (gpu < 0 && Caffe::world() > 1) gpu = Caffe::local_rank();      // one process per GPU (torch.distributed.run style launch)
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

// device selection shared by test / time: --gpu (default 0); there is no CPU mode in this build
static void UseGpuFlag() {
  const int gpu = atoi(flag("gpu", "0").c_str());
  CHECK_GE(gpu, 0) << "Use CPU. -- not available: this build is the GPU path only";
  LOG(INFO) << "Use GPU with device ID " << gpu;
  Caffe::SetDevice(gpu);
  Caffe::set_mode(Caffe::GPU);
  if (flag("precision").size()) Caffe::set_precision(flag("precision"));
}

// `caffe test` (tools/caffe.cpp:127-189): --iterations forward passes of the TEST-phase net; every scalar of every output
// blob is logged per batch ("Batch i, name = v") and as its mean at the end ("name = mean (* w = w*mean loss)").
static int test() {
  CHECK_GT(flag("model").size(), 0u) << "Need a model definition to score.";
  CHECK_GT(flag("weights").size(), 0u) << "Need model weights to score.";
  UseGpuFlag();
  Caffe::set_phase(Caffe::TEST);
  Net<float> net(flag("model"), Caffe::TEST);
  net.CopyTrainedLayersFrom(flag("weights"));
  const int passes = atoi(flag("iterations", "50").c_str());
  LOG(INFO) << "Running for " << passes << " iterations.";
  struct Scalar { int blob; double sum; };
  vector<Scalar> scalars;
  double loss_sum = 0;
  for (int pass = 0; pass < passes; ++pass) {
    float pass_loss = 0;
    const vector<Blob<float>*>& outs = net.Forward(vector<Blob<float>*>(), &pass_loss);
    loss_sum += pass_loss;
    size_t k = 0;
    for (size_t j = 0; j < outs.size(); ++j)
      for (int e = 0; e < outs[j]->count(); ++e, ++k) {
        if (k == scalars.size()) scalars.push_back(Scalar{net.output_blob_indices()[j], 0.0});
        const float v = outs[j]->cpu_data()[e];
        scalars[k].sum += v;
        LOG(INFO) << "Batch " << pass << ", " << net.blob_names()[scalars[k].blob] << " = " << v;
      }
  }
  LOG(INFO) << "Loss: " << (float)(loss_sum / passes);
  for (const Scalar& sc : scalars) {
    const float mean = (float)(sc.sum / passes), w = net.blob_loss_weights()[sc.blob];
    std::ostringstream tail;
    if (w) tail << " (* " << w << " = " << w * mean << " loss)";
    LOG(INFO) << net.blob_names()[sc.blob] << " = " << mean << tail.str();
  }
  return 0;
}

// tools/caffe.cpp:193-265.  The reference times every layer's Forward and Backward separately; here the graph is one
// fused plan, so what can be timed are its kernels (device time from events on their dispatch packets) and the whole
// forward-backward(-update) iteration (host clock around the loop, device drained at both ends).
static int time_net() {
  CHECK_GT(flag("model").size(), 0u) << "Need a model definition to time.";
  UseGpuFlag();
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
  const char* kernels[] = {"dedup", "fwd_gemm", "score_loss", "segsum", "wgrad_gemm", "reduce", "sgd", "reduce_sgd"};
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
  // every rank of a data-parallel job logs to its own file
  if (flag("log_file").size()) SetLogFile(Caffe::rank() == 0 ? flag("log_file") : flag("log_file") + ".rank" + std::to_string(Caffe::rank()));
  if (flag("max_tries_for_negs").size()) FLAGS_max_tries_for_negs = atoi(flag("max_tries_for_negs").c_str());
  if (flag("num_classes").size()) FLAGS_num_classes = atoi(flag("num_classes").c_str());
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
