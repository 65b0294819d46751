// caffe.cpp -- the `caffe train` command of the reference (tools/caffe.cpp:18-30,80-123,268-287)
// for the MI355X videovec path.
//   caffe train --solver=<solver.prototxt> [--weights=<.caffemodel>] [--snapshot=<.solverstate>] [--gpu=N]
// Extra flags of this build: --precision=f16|bf16, --log_file=<path> (what GLOG_log_dir gives the
// reference's train script, projects/videovec_embedding/train_mednet_embedding.sh:6).
#include <cstring>
#include <map>

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
  fprintf(stderr, "caffe: command line brew\nusage: caffe <command> <args>\n\ncommands:\n"
                  "  train           train or finetune a model\n"
                  "(test, time and device_query are outside the videovec training path built here)\n");
  return 1;
}
