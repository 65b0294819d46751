// solver.hpp -- the solver classes of the facade.  Same public surface as the reference's
// include/caffe/solver.hpp:17-143 (class names, method names, argument meaning, GetSolver), so code written
// against it compiles unchanged; underneath, one fused kernel (k_sgd) applies whichever update rule is selected.
#pragma once
#include "caffe/net.hpp"

namespace caffe {

typedef pl::Message SolverParameter;
typedef pl::Message SolverState;

// Drives training: owns the TRAIN net and the TEST nets, runs the iteration loop, writes / restores snapshots.
template <typename Dtype>
class Solver {
 public:
  explicit Solver(const SolverParameter& solver_param);
  explicit Solver(const string& solver_prototxt);
  virtual ~Solver() {}

  void Init(const SolverParameter& solver_param);            // solver.cpp:32-44
  void InitTrainNet();                                       // solver.cpp:46-82
  void InitTestNets();                                       // solver.cpp:84-157

  // the training loop (solver.cpp:159-240); `solverstate` resumes from a snapshot
  virtual void Solve(const char* solverstate = NULL);
  inline void Solve(const string solverstate) { Solve(solverstate.c_str()); }

  inline shared_ptr<Net<Dtype> > net() { return net_; }
  inline const vector<shared_ptr<Net<Dtype> > >& test_nets() { return test_nets_; }
  int iter() const { return iter_; }

 protected:
  // hooks of the concrete solver
  virtual void PreSolve() {}
  virtual void ComputeUpdateValue() = 0;
  // The iteration's update hyper-parameters handed to the net BEFORE ForwardBackward, without the log line (ComputeUpdateValue prints it
  // where the reference does, behind the loss lines): Solver::Step runs ForwardBackward, ComputeUpdateValue and Update back to back and
  // reads no diff in between, so the net may announce the update to the library (Net::HintUpdate -> vv_update_hint)
  virtual void PrepareUpdate() {}
  virtual void SnapshotSolverState(SolverState* out) = 0;
  virtual void RestoreSolverState(const SolverState& in) = 0;

  void Snapshot();                                           // solver.cpp:320-341
  void Restore(const char* solverstate);                     // solver.cpp:418-429
  void TestAll();
  void Step(bool display);                                   // one iteration of Solve's loop body (solver.cpp:194-220)
  void ReportOutputs(const Net<Dtype>& net, const char* prefix, bool with_iter, const vector<Dtype>& values);
  void Test(const int which_test_net = 0);                   // solver.cpp:251-317

  SolverParameter param_;
  int iter_;
  shared_ptr<Net<Dtype> > net_;
  vector<shared_ptr<Net<Dtype> > > test_nets_;
};

// Learning-rate policies, momentum / decay hyper-parameters and the history blobs (solver.hpp:66-94).  The update
// itself runs on the device; history() hands out host views of the device-resident momentum buffers.
template <typename Dtype>
class SGDSolver : public Solver<Dtype> {
 public:
  explicit SGDSolver(const SolverParameter& solver_param) : Solver<Dtype>(solver_param) {}
  explicit SGDSolver(const string& solver_prototxt) : Solver<Dtype>(solver_prototxt) {}
  const vector<shared_ptr<Blob<Dtype> > >& history();

 protected:
  Dtype GetLearningRate();                                   // solver.cpp:440-460
  virtual int solver_type() const { return VV_SOLVER_SGD; }  // which rule k_sgd applies
  virtual void PreSolve();
  virtual void ComputeUpdateValue();                         // solver.cpp:485-531 (hyper-parameters only)
  virtual void PrepareUpdate();
  virtual void SnapshotSolverState(SolverState* out);        // solver.cpp:578-586
  virtual void RestoreSolverState(const SolverState& in);    // solver.cpp:588-596
  vector<shared_ptr<Blob<Dtype> > > history_;
};

// NesterovSolver (solver.hpp:96-113, solver.cpp:599-655) and AdaGradSolver (solver.hpp:115-126, solver.cpp:714-781)
// differ from SGDSolver only in the rule the fused kernel applies, so both are one class parameterised by the rule.
template <typename Dtype, int RULE>
class UpdateRuleSolver : public SGDSolver<Dtype> {
 public:
  explicit UpdateRuleSolver(const SolverParameter& solver_param) : SGDSolver<Dtype>(solver_param) { CheckHyperParams(); }
  explicit UpdateRuleSolver(const string& solver_prototxt) : SGDSolver<Dtype>(solver_prototxt) { CheckHyperParams(); }

 protected:
  virtual int solver_type() const { return RULE; }
  void CheckHyperParams() {
    if (RULE == VV_SOLVER_ADAGRAD)                           // the reference's constructor_sanity_check, solver.hpp:121-122
      CHECK_EQ(0, this->param_.get_num("momentum")) << "Momentum cannot be used with AdaGrad.";
  }
};
template <typename Dtype> using NesterovSolver = UpdateRuleSolver<Dtype, VV_SOLVER_NESTEROV>;
template <typename Dtype> using AdaGradSolver = UpdateRuleSolver<Dtype, VV_SOLVER_ADAGRAD>;

// SolverParameter.solver_type -> solver object (solver.hpp:128-143)
template <typename Dtype>
Solver<Dtype>* GetSolver(const SolverParameter& solver_param);

}  // namespace caffe
