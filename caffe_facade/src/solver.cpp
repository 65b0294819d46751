// solver.cpp -- Solver / SGDSolver (reference: src/caffe/solver.cpp).  The arithmetic of
// ComputeUpdateValue + Net::Update runs as one fused HIP kernel; this file keeps the reference's
// control flow, log lines (parsed by caffe_utils/plot_training_stats.py) and snapshot formats.
#include "caffe/solver.hpp"

#include <cmath>
#include <cstdio>

namespace caffe {

template <typename Dtype>
Solver<Dtype>::Solver(const SolverParameter& param) : param_("SolverParameter"), iter_(0) { Init(param); }
template <typename Dtype>
Solver<Dtype>::Solver(const string& param_file) : param_("SolverParameter"), iter_(0) {
  SolverParameter param("SolverParameter");
  pl::ReadProtoFromTextFileOrDie(param_file, &param);
  Init(param);
}

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
void Solver<Dtype>::InitTrainNet() {
  const int num_train_nets = param_.has("net") + param_.has("net_param") + param_.has("train_net") + param_.has("train_net_param");
  const string field_names = "net, net_param, train_net, train_net_param";
  CHECK_GE(num_train_nets, 1) << "SolverParameter must specify a train net using one of these fields: " << field_names;
  CHECK_LE(num_train_nets, 1) << "SolverParameter must not contain more than one of these fields specifying a train_net: " << field_names;
  NetParameter net_param("NetParameter");
  if (param_.has("train_net_param")) { LOG(INFO) << "Creating training net specified in train_net_param."; net_param = param_.get_msg("train_net_param"); }
  else if (param_.has("train_net")) { LOG(INFO) << "Creating training net from train_net file: " << param_.get_str("train_net"); pl::ReadProtoFromTextFileOrDie(param_.get_str("train_net"), &net_param); }
  if (param_.has("net_param")) { LOG(INFO) << "Creating training net specified in net_param."; net_param = param_.get_msg("net_param"); }
  if (param_.has("net")) { LOG(INFO) << "Creating training net from net file: " << param_.get_str("net"); pl::ReadProtoFromTextFileOrDie(param_.get_str("net"), &net_param); }
  // precedence of the net state: solver train_state over the net's own (solver.cpp:73-80)
  NetState state("NetState");
  state.set_enum("phase", "TRAIN");
  if (net_param.has("state")) { const NetState& s = net_param.get_msg("state"); if (s.has("level")) state.set_int("level", s.get_int("level")); for (int i = 0; i < s.size("stage"); ++i) state.add_str("stage", s.get_str("stage", i)); state.set_enum("phase", "TRAIN"); }
  if (param_.has("train_state")) { const NetState& s = param_.get_msg("train_state"); if (s.has("level")) state.set_int("level", s.get_int("level")); for (int i = 0; i < s.size("stage"); ++i) state.add_str("stage", s.get_str("stage", i)); }
  *net_param.mutable_msg("state") = state;
  net_.reset(new Net<Dtype>(net_param));
}

template <typename Dtype>
void Solver<Dtype>::InitTestNets() {
  // solver.cpp:84-157.  Sources of test nets, in the reference's order: test_net_param, test_net
  // files, then the generic net_param / net (instantiated test_iter.size() - (others) times).
  const bool has_net_param = param_.has("net_param"), has_net_file = param_.has("net");
  const int num_generic_nets = has_net_param + has_net_file;
  CHECK_LE(num_generic_nets, 1) << "Both net_param and net_file may not be specified.";
  const int num_test_net_params = param_.size("test_net_param"), num_test_net_files = param_.size("test_net");
  const int num_test_nets = num_test_net_params + num_test_net_files;
  if (num_generic_nets) CHECK_GE(param_.size("test_iter"), num_test_nets) << "test_iter must be specified for each test network.";
  else CHECK_EQ(param_.size("test_iter"), num_test_nets) << "test_iter must be specified for each test network.";
  const int num_generic_net_instances = param_.size("test_iter") - num_test_nets;
  const int num_test_net_instances = num_test_nets + num_generic_net_instances;
  if (param_.size("test_state")) CHECK_EQ(param_.size("test_state"), num_test_net_instances) << "test_state must be unspecified or specified once per test net.";
  if (num_test_net_instances) CHECK_GT(param_.get_int("test_interval"), 0);
  vector<NetParameter> net_params;
  for (int i = 0; i < num_test_net_params; ++i) net_params.push_back(param_.get_msg("test_net_param", i));
  for (int i = 0; i < num_test_net_files; ++i) { NetParameter np("NetParameter"); pl::ReadProtoFromTextFileOrDie(param_.get_str("test_net", i), &np); net_params.push_back(np); }
  for (int i = 0; i < num_generic_net_instances; ++i) {
    NetParameter np("NetParameter");
    if (has_net_param) np = param_.get_msg("net_param"); else pl::ReadProtoFromTextFileOrDie(param_.get_str("net"), &np);
    net_params.push_back(np);
  }
  Caffe::set_phase(Caffe::TEST);
  for (int i = 0; i < num_test_net_instances; ++i) {
    NetState state("NetState");
    state.set_enum("phase", "TEST");
    if (param_.size("test_state")) { const NetState& s = param_.get_msg("test_state", i); if (s.has("level")) state.set_int("level", s.get_int("level")); for (int k = 0; k < s.size("stage"); ++k) state.add_str("stage", s.get_str("stage", k)); }
    *net_params[i].mutable_msg("state") = state;
    LOG(INFO) << "Creating test net (#" << i << ")";
    test_nets_.push_back(shared_ptr<Net<Dtype> >(new Net<Dtype>(net_params[i])));
  }
  Caffe::set_phase(Caffe::TRAIN);
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
  const int start_iter = iter_;
  vector<Blob<Dtype>*> bottom_vec;
  for (; iter_ < param_.get_int("max_iter"); ++iter_) {
    if (param_.get_int("snapshot") && iter_ > start_iter && iter_ % param_.get_int("snapshot") == 0) Snapshot();
    if (param_.get_int("test_interval") && iter_ % param_.get_int("test_interval") == 0 &&
        (iter_ > 0 || param_.get_bool("test_initialization"))) TestAll();
    const bool display = param_.get_int("display") && iter_ % param_.get_int("display") == 0;
    net_->set_debug_info(display && param_.get_bool("debug_info"));
    net_->set_loss_needed(display);
    Dtype loss = net_->ForwardBackward(bottom_vec);
    if (display) {
      LOG(INFO) << "Iteration " << iter_ << ", loss = " << loss;                      // solver.cpp:196
      const vector<Blob<Dtype>*>& result = net_->output_blobs();
      int score_index = 0;
      for (size_t j = 0; j < result.size(); ++j) {
        const Dtype* result_vec = result[j]->cpu_data();
        const string& output_name = net_->blob_names()[net_->output_blob_indices()[j]];
        const Dtype loss_weight = net_->blob_loss_weights()[net_->output_blob_indices()[j]];
        for (int k = 0; k < result[j]->count(); ++k) {
          std::ostringstream loss_msg_stream;
          if (loss_weight) loss_msg_stream << " (* " << loss_weight << " = " << loss_weight * result_vec[k] << " loss)";
          LOG(INFO) << "    Train net output #" << score_index++ << ": " << output_name << " = "
                    << "iter = " << iter_ << " value = " << result_vec[k] << loss_msg_stream.str();   // solver.cpp:211-214
        }
      }
    }
    ComputeUpdateValue();
    net_->Update();
  }
  net_->set_loss_needed(true);
  if (param_.get_bool("snapshot_after_train")) Snapshot();
  if (param_.get_int("display") && iter_ % param_.get_int("display") == 0) {
    Dtype loss;
    net_->Forward(bottom_vec, &loss);
    LOG(INFO) << "Iteration " << iter_ << ", loss = " << loss;
  }
  if (param_.get_int("test_interval") && iter_ % param_.get_int("test_interval") == 0) TestAll();
  LOG(INFO) << "Optimization Done.";
}

template <typename Dtype>
void Solver<Dtype>::TestAll() { for (size_t i = 0; i < test_nets_.size(); ++i) Test((int)i); }
template <typename Dtype>
void Solver<Dtype>::Test(const int test_net_id) {                                      // solver.cpp:251-317
  LOG(INFO) << "Iteration " << iter_ << ", Testing net (#" << test_net_id << ")";
  Caffe::set_phase(Caffe::TEST);
  const shared_ptr<Net<Dtype> >& test_net = test_nets_[test_net_id];
  test_net->ShareTrainedLayersWith(net_.get());
  vector<Dtype> test_score;
  vector<int> test_score_output_id;
  vector<Blob<Dtype>*> bottom_vec;
  Dtype loss = 0;
  const int iters = (int)param_.get_int("test_iter", test_net_id);
  for (int i = 0; i < iters; ++i) {
    Dtype iter_loss;
    const vector<Blob<Dtype>*>& result = test_net->Forward(bottom_vec, &iter_loss);
    if (param_.get_bool("test_compute_loss")) loss += iter_loss;
    int idx = 0;
    for (size_t j = 0; j < result.size(); ++j)
      for (int k = 0; k < result[j]->count(); ++k) {
        if (i == 0) { test_score.push_back(result[j]->cpu_data()[k]); test_score_output_id.push_back((int)j); }
        else test_score[idx++] += result[j]->cpu_data()[k];
      }
  }
  if (param_.get_bool("test_compute_loss")) LOG(INFO) << "Test loss: " << loss / iters;
  for (size_t i = 0; i < test_score.size(); ++i) {
    const int output_blob_index = test_net->output_blob_indices()[test_score_output_id[i]];
    const string& output_name = test_net->blob_names()[output_blob_index];
    const Dtype loss_weight = test_net->blob_loss_weights()[output_blob_index];
    std::ostringstream loss_msg_stream;
    const Dtype mean_score = test_score[i] / iters;
    if (loss_weight) loss_msg_stream << " (* " << loss_weight << " = " << loss_weight * mean_score << " loss)";
    LOG(INFO) << "    Test net output #" << i << ": " << output_name << " = " << mean_score << loss_msg_stream.str();
  }
  Caffe::set_phase(Caffe::TRAIN);
}

template <typename Dtype>
void Solver<Dtype>::Snapshot() {
  NetParameter net_param("NetParameter");
  net_->ToProto(&net_param, param_.get_bool("snapshot_diff"));
  char iter_str[32];
  snprintf(iter_str, sizeof(iter_str), "_iter_%d", iter_);
  const string filename = param_.get_str("snapshot_prefix") + iter_str;
  const string model_filename = filename + ".caffemodel";
  LOG(INFO) << "Snapshotting to " << model_filename;
  pl::WriteProtoToBinaryFile(net_param, model_filename);
  SolverState state("SolverState");
  SnapshotSolverState(&state);
  state.set_int("iter", iter_);
  state.set_str("learned_net", model_filename);
  const string snapshot_filename = filename + ".solverstate";
  LOG(INFO) << "Snapshotting solver state to " << snapshot_filename;
  pl::WriteProtoToBinaryFile(state, snapshot_filename);
}

template <typename Dtype>
void Solver<Dtype>::Restore(const char* state_file) {
  SolverState state("SolverState");
  pl::ReadProtoFromBinaryFileOrDie(state_file, &state);
  if (state.has("learned_net")) {
    NetParameter net_param("NetParameter");
    pl::ReadProtoFromBinaryFileOrDie(state.get_str("learned_net"), &net_param);
    net_->CopyTrainedLayersFrom(net_param);
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
