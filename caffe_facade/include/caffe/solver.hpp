// solver.hpp -- Solver / SGDSolver with the reference's interface (include/caffe/solver.hpp:17-143,
// src/caffe/solver.cpp).  Nesterov / AdaGrad are not built (not used by the project's solver file).
#pragma once
#include "caffe/net.hpp"

namespace caffe {

typedef pl::Message SolverParameter;
typedef pl::Message SolverState;

template <typename Dtype>
class Solver {
 public:
  explicit Solver(const SolverParameter& param);
  explicit Solver(const string& param_file);
  void Init(const SolverParameter& param);          // solver.cpp:32-44
  void InitTrainNet();                              // solver.cpp:46-82
  void InitTestNets();                              // solver.cpp:84-157
  virtual void Solve(const char* resume_file = NULL);   // solver.cpp:159-240
  inline void Solve(const string resume_file) { Solve(resume_file.c_str()); }
  virtual ~Solver() {}
  inline shared_ptr<Net<Dtype> > net() { return net_; }
  inline const vector<shared_ptr<Net<Dtype> > >& test_nets() { return test_nets_; }
  int iter() const { return iter_; }
 protected:
  virtual void PreSolve() {}
  virtual void ComputeUpdateValue() = 0;
  void Snapshot();                                  // solver.cpp:320-341
  void TestAll();
  void Test(const int test_net_id = 0);
  virtual void SnapshotSolverState(SolverState* state) = 0;
  void Restore(const char* resume_file);            // solver.cpp:418-429
  virtual void RestoreSolverState(const SolverState& state) = 0;
  SolverParameter param_;
  int iter_;
  shared_ptr<Net<Dtype> > net_;
  vector<shared_ptr<Net<Dtype> > > test_nets_;
};

template <typename Dtype>
class SGDSolver : public Solver<Dtype> {
 public:
  explicit SGDSolver(const SolverParameter& param) : Solver<Dtype>(param) {}
  explicit SGDSolver(const string& param_file) : Solver<Dtype>(param_file) {}
  // momentum history of every parameter blob (host views of the device buffers)
  const vector<shared_ptr<Blob<Dtype> > >& history();
 protected:
  virtual void PreSolve();
  Dtype GetLearningRate();                          // solver.cpp:440-460
  virtual void ComputeUpdateValue();                // solver.cpp:485-531
  virtual int solver_type() const { return VV_SOLVER_SGD; }
  virtual void SnapshotSolverState(SolverState* state);     // solver.cpp:578-586
  virtual void RestoreSolverState(const SolverState& state);  // solver.cpp:588-596
  vector<shared_ptr<Blob<Dtype> > > history_;
};

// solver.hpp:96-113: same hyper-parameters, the update rule differs (solver.cpp:599-655) -- in the fused k_sgd kernel
template <typename Dtype>
class NesterovSolver : public SGDSolver<Dtype> {
 public:
  explicit NesterovSolver(const SolverParameter& param) : SGDSolver<Dtype>(param) {}
  explicit NesterovSolver(const string& param_file) : SGDSolver<Dtype>(param_file) {}
 protected:
  virtual int solver_type() const { return VV_SOLVER_NESTEROV; }
};

// solver.hpp:115-126, solver.cpp:714-781
template <typename Dtype>
class AdaGradSolver : public SGDSolver<Dtype> {
 public:
  explicit AdaGradSolver(const SolverParameter& param) : SGDSolver<Dtype>(param) { constructor_sanity_check(); }
  explicit AdaGradSolver(const string& param_file) : SGDSolver<Dtype>(param_file) { constructor_sanity_check(); }
 protected:
  virtual int solver_type() const { return VV_SOLVER_ADAGRAD; }
  void constructor_sanity_check() {
    CHECK_EQ(0, this->param_.get_num("momentum")) << "Momentum cannot be used with AdaGrad.";       // solver.hpp:121-122
  }
};

template <typename Dtype>
Solver<Dtype>* GetSolver(const SolverParameter& param);   // solver.hpp:128-143

}  // namespace caffe
