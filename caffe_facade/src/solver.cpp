// solver.cpp -- Solver / SGDSolver (reference: src/caffe/solver.cpp).  The arithmetic of
// ComputeUpdateValue + Net::Update runs as one fused HIP kernel; this file keeps the reference's
// control flow, log lines (parsed by caffe_utils/plot_training_stats.py) and snapshot formats.
#include "caffe/solver.hpp"

#include <cmath>
#include <cstdio>
#include <sstream>
#include <string>

namespace caffe {

template <typename Dtype>
Solver<Dtype>::Solver(const SolverParameter& param) : param_("SolverParameter"), iter_(0) { Init(param); }
template <typename Dtype>
Solver<Dtype>::Solver(const string& param_file) : param_("SolverParameter"), iter_(0) {
  SolverParameter param("SolverParameter");
  pl::ReadProtoFromTextFileOrDie(param_file, &param);
  Init(param);
}

// ---- which nets a SolverParameter names -------------------------------------------------------------------------
// The reference accepts the train net from one of four fields and the test nets from three kinds of source
// (solver.cpp:46-157).  Here the fields are tables; what is contract is which field wins, the CHECK messages and the
// "Creating ..." log lines.
namespace {

// level and stages of `src` laid over `dst` (NetState::MergeFrom for the two fields the path uses)
void OverlayState(const NetState& src, NetState* dst) {
  if (src.has("level")) dst->set_int("level", src.get_int("level"));
  for (int i = 0; i < src.size("stage"); ++i) dst->add_str("stage", src.get_str("stage", i));
}

struct NetField { const char* name; bool is_file; const char* what; };
const NetField kTrainFields[] = {     // later entries override earlier ones (solver.cpp:52-71)
  {"train_net_param", false, "Creating training net specified in train_net_param."},
  {"train_net", true, "Creating training net from train_net file: "},
  {"net_param", false, "Creating training net specified in net_param."},
  {"net", true, "Creating training net from net file: "},
};

NetParameter ReadNet(const SolverParameter& sp, const char* field, bool is_file, int index) {
  NetParameter np("NetParameter");
  if (is_file) pl::ReadProtoFromTextFileOrDie(sp.get_str(field, index), &np);
  else np = sp.get_msg(field, index);
  return np;
}

NetParameter TrainNetOf(const SolverParameter& sp) {
  int given = 0;
  for (const NetField& f : kTrainFields) given += sp.has(f.name) ? 1 : 0;
  const string field_names = "net, net_param, train_net, train_net_param";
  CHECK_GE(given, 1) << "SolverParameter must specify a train net using one of these fields: " << field_names;
  CHECK_LE(given, 1) << "SolverParameter must not contain more than one of these fields specifying a train_net: " << field_names;
  NetParameter np("NetParameter");
  for (const NetField& f : kTrainFields) {
    if (!sp.has(f.name)) continue;
    if (f.is_file) LOG(INFO) << f.what << sp.get_str(f.name); else LOG(INFO) << f.what;
    np = ReadNet(sp, f.name, f.is_file, 0);
  }
  // net state, weakest first: phase TRAIN, the net's own state, the solver's train_state (solver.cpp:73-80)
  NetState state("NetState");
  state.set_enum("phase", "TRAIN");
  if (np.has("state")) OverlayState(np.get_msg("state"), &state);
  if (sp.has("train_state")) OverlayState(sp.get_msg("train_state"), &state);
  *np.mutable_msg("state") = state;
  return np;
}

vector<NetParameter> TestNetsOf(const SolverParameter& sp) {
  const bool generic_inline = sp.has("net_param"), generic_file = sp.has("net");
  CHECK_LE(generic_inline + generic_file, 1) << "Both net_param and net_file may not be specified.";
  const int explicit_nets = sp.size("test_net_param") + sp.size("test_net");
  const int iters_given = sp.size("test_iter");
  if (generic_inline || generic_file) CHECK_GE(iters_given, explicit_nets) << "test_iter must be specified for each test network.";
  else CHECK_EQ(iters_given, explicit_nets) << "test_iter must be specified for each test network.";
  const int instances = iters_given;                       // explicit ones first, the generic net fills the rest
  if (sp.size("test_state")) CHECK_EQ(sp.size("test_state"), instances) << "test_state must be unspecified or specified once per test net.";
  if (instances) CHECK_GT(sp.get_int("test_interval"), 0);
  vector<NetParameter> nets;
  for (int i = 0; i < sp.size("test_net_param"); ++i) nets.push_back(ReadNet(sp, "test_net_param", false, i));
  for (int i = 0; i < sp.size("test_net"); ++i) nets.push_back(ReadNet(sp, "test_net", true, i));
  while ((int)nets.size() < instances) nets.push_back(ReadNet(sp, generic_inline ? "net_param" : "net", !generic_inline, 0));
  for (int i = 0; i < instances; ++i) {
    NetState state("NetState");
    state.set_enum("phase", "TEST");
    if (sp.size("test_state")) OverlayState(sp.get_msg("test_state", i), &state);
    *nets[i].mutable_msg("state") = state;
  }
  return nets;
}

}  // namespace

template <typename Dtype>
void Solver<Dtype>::Init(const SolverParameter& param) {
  LOG(INFO) << "Initializing solver from parameters: \n" << param.PrintText();
  param_ = param;
  if (param_.get_int("random_seed") >= 0) Caffe::set_random_seed((unsigned)param_.get_int("random_seed"));   // solver.cpp:37-39
  InitTrainNet();
  InitTestNets();
  LOG(INFO) << "Solver scaffolding done.";
}

template <typename Dtype>
void Solver<Dtype>::InitTrainNet() { net_.reset(new Net<Dtype>(TrainNetOf(param_))); }

template <typename Dtype>
void Solver<Dtype>::InitTestNets() {
  const vector<NetParameter> specs = TestNetsOf(param_);
  Caffe::set_phase(Caffe::TEST);
  for (size_t i = 0; i < specs.size(); ++i) {
    LOG(INFO) << "Creating test net (#" << i << ")";
    test_nets_.push_back(shared_ptr<Net<Dtype> >(new Net<Dtype>(specs[i])));
  }
  Caffe::set_phase(Caffe::TRAIN);
}

// ---- the training loop -------------------------------------------------------------------------------------------
// Solver::Solve (solver.cpp:159-240).  One iteration = periodic work that looks at iter_ (snapshot, test), then
// ForwardBackward, the display lines, ComputeUpdateValue + Update -- in that order, because the log lines of an
// iteration (loss, outputs, lr) are parsed as a group by caffe_utils/plot_training_stats.py:10-14.
// Data-parallel runs (Caffe::world() > 1): every rank runs this loop; only rank 0 writes snapshots.
namespace {
bool Every(int64_t period, int iter) { return period > 0 && iter % period == 0; }
}

template <typename Dtype>
void Solver<Dtype>::ReportOutputs(const Net<Dtype>& net, const char* prefix, bool with_iter, const vector<Dtype>& values) {
  // one line per scalar of every output blob: "<prefix> #k: name = [iter = I value = ]v[ (* w = w*v loss)]"  (solver.cpp:211-214, 305-315)
  size_t k = 0;
  const vector<Blob<Dtype>*>& outs = net.output_blobs();
  for (size_t j = 0; j < outs.size(); ++j) {
    const int blob = net.output_blob_indices()[j];
    const Dtype w = net.blob_loss_weights()[blob];
    for (int e = 0; e < outs[j]->count(); ++e, ++k) {
      std::ostringstream line;
      line << "    " << prefix << " #" << k << ": " << net.blob_names()[blob] << " = ";
      if (with_iter) line << "iter = " << iter_ << " value = ";
      line << values[k];
      if (w) line << " (* " << w << " = " << w * values[k] << " loss)";
      LOG(INFO) << line.str();
    }
  }
}

template <typename Dtype>
void Solver<Dtype>::Step(bool display) {
  vector<Blob<Dtype>*> no_bottom;
  net_->set_debug_info(display && param_.get_bool("debug_info"));
  net_->set_loss_needed(display);                   // the loss is read back from the device only when it is shown
  PrepareUpdate();                                  // this iteration's rate (a function of iter_) ...
  // ... and: ForwardBackward is followed by Update and nothing reads a diff in between -- unless this solver's snapshots carry the diffs
  // (snapshot_diff, caffe.proto SolverParameter field 16; Net::ToProto, net.cpp:784-800, then reads them back): such a job keeps its
  // gradient, i.e. the update stays its own launch (with the hint the shipped 4096 x 4096 shape applies the rule in the
  // weight-gradient GEMM's epilogue and the gradient never exists outside its registers)
  if (!param_.get_bool("snapshot_diff")) net_->HintUpdate();
  const Dtype loss = net_->ForwardBackward(no_bottom);
  if (display) {
    LOG(INFO) << "Iteration " << iter_ << ", loss = " << loss;                          // solver.cpp:196
    vector<Dtype> values;
    for (Blob<Dtype>* b : net_->output_blobs()) values.insert(values.end(), b->cpu_data(), b->cpu_data() + b->count());
    ReportOutputs(*net_, "Train net output", true, values);
  }
  ComputeUpdateValue();
  net_->Update();
}

template <typename Dtype>
void Solver<Dtype>::Solve(const char* resume_file) {
  Caffe::set_phase(Caffe::TRAIN);
  LOG(INFO) << "Solving " << net_->name();
  PreSolve();
  iter_ = 0;
  if (resume_file) {
    LOG(INFO) << "Restoring previous solver status from " << resume_file;
    Restore(resume_file);
  }
  const int first = iter_, last = (int)param_.get_int("max_iter");
  const int64_t snap = param_.get_int("snapshot"), test = param_.get_int("test_interval"), disp = param_.get_int("display");
  for (; iter_ < last; ++iter_) {
    if (iter_ > first && Every(snap, iter_)) Snapshot();
    if (Every(test, iter_) && (iter_ > 0 || param_.get_bool("test_initialization"))) TestAll();
    Step(Every(disp, iter_));
  }
  // after the last update: final snapshot, one more forward pass for the closing loss line, final test
  net_->set_loss_needed(true);
  if (param_.get_bool("snapshot_after_train")) Snapshot();
  if (Every(disp, iter_)) {
    Dtype loss;
    net_->Forward(vector<Blob<Dtype>*>(), &loss);
    LOG(INFO) << "Iteration " << iter_ << ", loss = " << loss;
  }
  if (Every(test, iter_)) TestAll();
  LOG(INFO) << "Optimization Done.";
}

template <typename Dtype>
void Solver<Dtype>::TestAll() { for (size_t i = 0; i < test_nets_.size(); ++i) Test((int)i); }

// Solver::Test (solver.cpp:251-317): test_iter forward passes of the test net on the current weights; every scalar
// of every output blob is averaged over the passes.
template <typename Dtype>
void Solver<Dtype>::Test(const int test_net_id) {
  LOG(INFO) << "Iteration " << iter_ << ", Testing net (#" << test_net_id << ")";
  Caffe::set_phase(Caffe::TEST);
  Net<Dtype>& net = *test_nets_[test_net_id];
  net.ShareTrainedLayersWith(net_.get());
  const int passes = (int)param_.get_int("test_iter", test_net_id);
  const bool want_loss = param_.get_bool("test_compute_loss");
  vector<Dtype> mean;
  Dtype loss_sum = 0;
  for (int pass = 0; pass < passes; ++pass) {
    Dtype pass_loss = 0;
    const vector<Blob<Dtype>*>& outs = net.Forward(vector<Blob<Dtype>*>(), &pass_loss);
    loss_sum += pass_loss;
    size_t k = 0;
    for (Blob<Dtype>* b : outs)
      for (int e = 0; e < b->count(); ++e, ++k) {
        if (k == mean.size()) mean.push_back(0);
        mean[k] += b->cpu_data()[e];
      }
  }
  for (Dtype& m : mean) m /= passes;
  if (want_loss) LOG(INFO) << "Test loss: " << loss_sum / passes;
  ReportOutputs(net, "Test net output", false, mean);
  Caffe::set_phase(Caffe::TRAIN);
}

// Solver::Snapshot (solver.cpp:320-341): <prefix>_iter_<N>.caffemodel + .solverstate naming the model file.
template <typename Dtype>
void Solver<Dtype>::Snapshot() {
  // data-parallel: parameters are identical on every rank, rank 0 writes.  (Under the sharded update the fp32 master rows live on
  // their owners: reading them back is a collective of the library -- vv_params_get -- so EVERY rank makes the call, then the others leave.)
  if (Caffe::world() > 1) net_->params();
  if (Caffe::rank() != 0) return;
  const string stem = param_.get_str("snapshot_prefix") + "_iter_" + std::to_string(iter_);
  const string model_file = stem + ".caffemodel", state_file = stem + ".solverstate";
  NetParameter learned("NetParameter");
  net_->ToProto(&learned, param_.get_bool("snapshot_diff"));
  LOG(INFO) << "Snapshotting to " << model_file;
  pl::WriteProtoToBinaryFile(learned, model_file);
  SolverState state("SolverState");
  SnapshotSolverState(&state);
  state.set_int("iter", iter_);
  state.set_str("learned_net", model_file);
  LOG(INFO) << "Snapshotting solver state to " << state_file;
  pl::WriteProtoToBinaryFile(state, state_file);
}

// Solver::Restore (solver.cpp:418-429)
template <typename Dtype>
void Solver<Dtype>::Restore(const char* state_file) {
  SolverState state("SolverState");
  pl::ReadProtoFromBinaryFileOrDie(state_file, &state);
  if (state.has("learned_net")) {
    NetParameter learned("NetParameter");
    pl::ReadProtoFromBinaryFileOrDie(state.get_str("learned_net"), &learned);
    net_->CopyTrainedLayersFrom(learned);
  }
  iter_ = (int)state.get_int("iter");
  RestoreSolverState(state);
}

// ------------------------------------------------------------------------------- SGD -----------
template <typename Dtype>
Dtype SGDSolver<Dtype>::GetLearningRate() {
  const SolverParameter& p = this->param_;
  const string lr_policy = p.get_str("lr_policy");
  const Dtype base = (Dtype)p.get_num("base_lr"), gamma = (Dtype)p.get_num("gamma");
  Dtype rate;
  if (lr_policy == "fixed") rate = base;
  else if (lr_policy == "step") rate = base * std::pow(gamma, (Dtype)(this->iter_ / (int)p.get_int("stepsize")));
  else if (lr_policy == "exp") rate = base * std::pow(gamma, (Dtype)this->iter_);
  else if (lr_policy == "inv") rate = base * std::pow(Dtype(1) + gamma * this->iter_, -(Dtype)p.get_num("power"));
  else { LOG(FATAL) << "Unknown learning rate policy: " << lr_policy; rate = 0; }
  return rate;
}
template <typename Dtype>
void SGDSolver<Dtype>::PreSolve() {}     // history lives on the device, zero-initialised with the parameters

template <typename Dtype>
void SGDSolver<Dtype>::ComputeUpdateValue() {
  const Dtype rate = GetLearningRate();
  if (this->param_.get_int("display") && this->iter_ % this->param_.get_int("display") == 0)
    LOG(INFO) << "Iteration " << this->iter_ << ", lr = " << rate;                    // solver.cpp:492-494
  this->net_->SetUpdateHyperParams(rate, (float)this->param_.get_num("momentum"), (float)this->param_.get_num("weight_decay"),
                                   this->param_.get_str("regularization_type"), solver_type(), (float)this->param_.get_num("delta"));
}
template <typename Dtype>
void SGDSolver<Dtype>::PrepareUpdate() {
  this->net_->SetUpdateHyperParams(GetLearningRate(), (float)this->param_.get_num("momentum"), (float)this->param_.get_num("weight_decay"),
                                   this->param_.get_str("regularization_type"), solver_type(), (float)this->param_.get_num("delta"));
}
template <typename Dtype>
const vector<shared_ptr<Blob<Dtype> > >& SGDSolver<Dtype>::history() { this->net_->GetHistory(&history_); return history_; }
template <typename Dtype>
void SGDSolver<Dtype>::SnapshotSolverState(SolverState* state) {
  state->clear("history");
  const vector<shared_ptr<Blob<Dtype> > >& h = history();
  for (size_t i = 0; i < h.size(); ++i) h[i]->ToProto(state->add_msg("history"));
}
template <typename Dtype>
void SGDSolver<Dtype>::RestoreSolverState(const SolverState& state) {
  CHECK_EQ(state.size("history"), 2) << "Incorrect length of history blobs.";
  LOG(INFO) << "SGDSolver: restoring history";
  vector<shared_ptr<Blob<Dtype> > > h(2);
  for (int i = 0; i < 2; ++i) { h[i].reset(new Blob<Dtype>()); h[i]->FromProto(state.get_msg("history", i)); }
  this->net_->SetHistory(h);
}

template <typename Dtype>
Solver<Dtype>* GetSolver(const SolverParameter& param) {
  const string type = param.get_enum("solver_type");
  if (type == "SGD") return new SGDSolver<Dtype>(param);
  if (type == "NESTEROV") return new NesterovSolver<Dtype>(param);
  if (type == "ADAGRAD") return new AdaGradSolver<Dtype>(param);
  LOG(FATAL) << "Unknown SolverType: " << type;                                        // solver.hpp:141
  return nullptr;
}

template class Solver<float>;
template class SGDSolver<float>;
template class UpdateRuleSolver<float, VV_SOLVER_NESTEROV>;
template class UpdateRuleSolver<float, VV_SOLVER_ADAGRAD>;
template Solver<float>* GetSolver(const SolverParameter& param);

}  // namespace caffe
