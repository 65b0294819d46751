/*
 * videovec.h -- C ABI of the MI355X-native videovec_embedding training path.
 *
 * The reference (eevignesh/videovector, a Caffe fork) has no FFI: its boundary for this path is the
 * C++ Layer / Solver class hierarchy.  This ABI is what a drop-in replacement of that path binds
 * to; every entry point names the reference interface it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the reference-side stubs (a Layer subclass, ctypes).
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every function returns VV_OK (0)
 * or a VV_ERR_* code; vv_last_error() gives the thread's last message.  Nothing throws across the
 * boundary.  "host" pointers are ordinary CPU memory; the context owns all device memory.
 * Errors the reference reports with CHECK/LOG(FATAL) (abort) are reported here as VV_ERR_ARG.
 */
#ifndef VIDEOVEC_H_
#define VIDEOVEC_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vv_ctx vv_ctx;

enum { VV_OK = 0, VV_ERR_ARG = 1, VV_ERR_HIP = 2, VV_ERR_STATE = 3, VV_ERR_NOGPU = 4 };

/* MFMA operand type of the two GEMMs (accumulation is always fp32). */
enum { VV_PREC_F16 = 0, VV_PREC_BF16 = 1 };

/* Loss norm: MaxMarginLossParameter.Norm (src/caffe/proto/caffe.proto:858-863). */
enum { VV_NORM_L1 = 1, VV_NORM_L2 = 2 };
/* SolverParameter.regularization_type (caffe.proto:130-132). */
enum { VV_REG_L1 = 1, VV_REG_L2 = 2 };
/* SolverParameter.solver_type (caffe.proto SolverType). */
enum { VV_SOLVER_SGD = 0, VV_SOLVER_NESTEROV = 1, VV_SOLVER_ADAGRAD = 2 };

/* One context per process / GPU.  Replaces Caffe::SetDevice + Caffe::set_mode(GPU)
 * (src/caffe/common.cpp:127-145, tools/caffe.cpp:92-104). */
int vv_create(int device, int prec, vv_ctx** out);
/* Caffe::DeviceQuery (src/caffe/common.cpp:147-180): a text description of the device (name, architecture, compute
 * units, clocks, memory, LDS / registers per block) into buf (NUL-terminated, truncated to n). */
int vv_device_query(int device, char* buf, size_t n);
int vv_destroy(vv_ctx* ctx);
const char* vv_last_error(void);
const char* vv_version(void);
/* Run all kernels of this context on an existing hipStream_t (NULL = the context's own stream).
 * The reference has only the default CUDA stream. */
int vv_set_stream(vv_ctx* ctx, void* hip_stream);
int vv_synchronize(vv_ctx* ctx);
/* Row de-duplication (default on; also VV_DEDUP=0/1 in the environment).  The reference sampler draws
 * every item's negatives from one shared buffer (video_sampled_shots_data_layer.cpp:836-875), so the
 * (C+Nn)*B rows of a batch repeat table rows.  With de-duplication the fc projection
 * (InnerProductLayer::Forward, inner_product_layer.cu:12-27) runs once per distinct row and the gradient rows of
 * a row's instances are summed before the weight-gradient product (inner_product_layer.cu:36-42): the same
 * sums, reassociated.  Steps with dropout always take the dense path.  vv_dedup_stats reports the rows of
 * the last forward/backward pass and how many were distinct (rows == unique_rows on the dense path). */
int vv_set_dedup(vv_ctx* ctx, int on);
/* Per-context execution switches by name; none of them changes a result beyond rounding, none is process-global (two contexts
 * of one process keep their own).  Each also has an environment variable that sets its INITIAL value when the context is
 * created.  The reference's counterpart is Caffe::set_mode / the layer's *_param fields (include/caffe/common.hpp:108-135):
 * per-object configuration, not globals.
 *   "dedup" (VV_DEDUP, 1)            row de-duplication, as vv_set_dedup
 *   "seg_bwd" (VV_SEG_BWD, 1)        segment-wise backward of de-duplicated batches (0: per-instance gradient rows + their sums)
 *   "drop_dedup" (VV_DROP_DEDUP, 1)  dropout on the de-duplicated path where the kernels carry per-instance masks (D = 512); 0: dense
 *   "h16" (VV_H16, 1)                ip2 stored as f16 between the forward GEMM and the segment-wise score / backward kernels (de-duplicated batches
 *                                    of D = 512 / 1024): half the bytes of the GEMM's epilogue and of every later read of a row; adds one f16 rounding
 *                                    (2^-12 relative per element; values past 65504 saturate) to the embeddings the loss is computed from --
 *                                    measured against the fp32 oracle on whole batches: rows 2.7e-4 -> 3.5e-4, scores 3.8e-5 -> 5.3e-5, loss
 *                                    unchanged to 1e-6 (inside the 1e-3 the path is held to); 0: fp32 rows (the rounds 1-5 form)
 *   "slab16" (VV_SLAB16, 1)          the weight gradient's split-K partial products as f16 x one power of two per (split, 256 x 256 tile) instead of
 *                                    fp32 (half of the 134 MB they cost per step at the benchmark's shape; a second rounding, 2^-12 relative per
 *                                    partial product, in the gradient path -- every split still accumulates in fp32 and the sum over the splits is
 *                                    taken in fp32: dW against the oracle on the same operands 3.5e-4 -> 4.6e-4 on whole batches); 0: fp32 slabs
 *   "v16" (VV_V16, 1)                with h16, D = 1024 (the one-sweep score kernel: hundreds of negatives per item): the two per-item vectors the
 *                                    segment-wise backward gathers once per instance stored as f16 (one more 2^-12 in the gradient)
 *   "score_pf" (VV_SCORE_PF, 1)      the register-resident score kernel's first-round workgroups request the second round's rows into the XCD's L2
 *                                    while they compute (bit-identical results)
 *   "fuse_update" (VV_FUSE_UPDATE, 1) reduction of the split-K partials and the solver update in one launch (0: two launches)
 *   "fwd_lead" (VV_FWD_LEAD, 1)      the forward GEMM's sibling lead
 *   "fwd_merge" (VV_FWD_MERGE, 0)    the forward GEMM with two phases per barrier pair (bit-identical results; measured not faster)
 *   "wgrad_tr" (VV_WGRAD_TR, 1)      transposed LDS reads in the weight-gradient GEMM (0: the first-round kernel)
 *   "wgrad_lean" (VV_WGRAD_LEAN, 1)  the weight-gradient GEMM's lean instantiations (tables below 4 GiB: gathered rows addressed as base + 32-bit
 *                                    offset, fewer instructions in the loop's LOAD segments; bit-identical results); 0: 64-bit addresses
 *   "score_stream" (VV_SCORE_STREAM, 0)  1: the one-sweep score kernel for every shape
 *   "comm_gate" (VV_COMM_GATE, 1)    the overlapped update gates the next forward GEMM chunk by chunk (0: the stream joins)
 *   "comm_inline" (VV_COMM_INLINE, 1)  the SHARDED update's three steps (reduce-scatter, the rule on this rank's rows, all-gather) are queued
 *                                    on the compute stream itself (0: on the communication stream, the next forward GEMM gated on one flag)
 *   "comm_first_inline" (VV_COMM_FIRST_INLINE, 0)  1: the OVERLAPPED update's first F-chunk (exchange, rule, publish) is queued on the compute stream,
 *                                    the others on the communication stream (measured slower on one rank: off by default)
 *   "wgrad_update" (VV_WGRAD_UPDATE, 1)  a step announced by vv_update_hint whose weight-gradient GEMM has one split of K applies the solver's
 *                                    rule in that GEMM's epilogue (bit-identical parameters; 0: the update as its own launch)
 *   "comm_chunks" (VV_COMM_CHUNKS, 3)  F-chunks of the overlapped update, 1 .. 4
 *   "comm_test_delay_us" (VV_COMM_TEST_DELAY_US, 0)  TEST HOOK: holds the communication stream this long in front of every chunk
 * Ablated / experimental kernels (timing studies whose results may be wrong) are NOT reachable through this library: they and their
 * switches exist only in the lab build (make -C videovector_amd/csrc lab).  Unknown name: VV_ERR_ARG. */
int vv_set_option(vv_ctx* ctx, const char* name, double value);
int vv_get_option(vv_ctx* ctx, const char* name, double* value);      /* ... and their current values */
int vv_dedup_stats(vv_ctx* ctx, int64_t* rows, int64_t* unique_rows);
/* f16 operands: the 16-bit gradient operand of the weight-gradient product (InnerProductLayer::Backward,
 * inner_product_layer.cpp:80-97) carries one power-of-two scale per step.  A gradient value outside f16's range is never
 * applied clipped: the kernels that round gradients record their maxima, and a step whose values did not fit produces
 * them again at a smaller scale before the product runs (exact: powers of two).  `repeats` = steps so far that needed
 * that (as reported by the device, four steps late); `scale` = the scale the next step starts with. */
int vv_grad_scale_stats(vv_ctx* ctx, int64_t* repeats, float* scale);

/* ---- feature table: stands for the rows the data layer copies out of the VideoShots DB
 * (VideoSampledShotsDataLayer::AddSamplesToTop, src/caffe/layers/video_sampled_shots_data_layer.cpp:
 * 439-452, and BasePrefetchingDataLayer::Forward_gpu's H2D copy, base_data_layer.cu:7-21).  The
 * table stays resident in HBM; a batch is then just row indices.  rows: host fp32 [n_rows][F]. */
int vv_table_set(vv_ctx* ctx, const float* rows, int64_t n_rows, int32_t F);
/* Fill the table with the synthetic features of videovector_amd/synth.py (integer hashing only,
 * bit-identical to the host generator). */
int vv_table_synth(vv_ctx* ctx, uint64_t seed, int64_t n_rows, int32_t F);
/* Read rows back as fp32 (host [n][F]); rows == NULL means 0..n-1. */
int vv_table_get(vv_ctx* ctx, const int32_t* rows, int64_t n, float* out);

/* ---- fc7 parameters and SGD history.  InnerProductLayer blobs_[0] (1,1,D,F) row-major D x F and
 * blobs_[1] (1,1,1,D) (src/caffe/layers/inner_product_layer.cpp:29,36); SGDSolver::history_
 * (include/caffe/solver.hpp:79,91).  NULL history = zeros.  All host fp32. */
int vv_params_set(vv_ctx* ctx, int32_t D, const float* W, const float* b, const float* hW,
                  const float* hb);
int vv_params_get(vv_ctx* ctx, float* W, float* b, float* hW, float* hb);

/* ---- one training iteration */
typedef struct {
  /* shapes: VideoSampledShotsDataParameter batch_size / context_size / num_negative_samples
   * (caffe.proto:562-620); F and D come from the table and the parameters. */
  int32_t B, C, Nn;
  /* MAX_MARGIN_LOSS (src/caffe/layers/max_margin_loss_layer.cpp:39-40, caffe.proto:858-868) and
   * the loss weight of its first top (include/caffe/layer.hpp:416-422). */
  float margin;
  int32_t norm;
  float loss_weight;
  /* ELTWISE SUM coefficients of context_average (eltwise_layer.cpp:22-32); host [C-1], NULL =
   * 1/(C-1) each (the shipped prototxt's 0.25 x 4). */
  const float* ctx_coeff;
  /* DROPOUT on ip2 (dropout_layer.cpp:13-22): ratio 0 = layer absent.  mask: host uint8
   * [(C+Nn)*B][D] in the reference's row order (row = ch*B + b), 1 = keep; NULL = a counter-based
   * mask from dropout_seed and the iteration counter. */
  float dropout_ratio;
  const uint8_t* dropout_mask;
  uint64_t dropout_seed;
  /* Loss normaliser for data-parallel shards: the GLOBAL B*Nn; 0 = this batch's B*Nn
   * (max_margin_loss_layer.cpp:67 uses the local count; there is no multi-GPU in the reference). */
  int64_t global_count;
  /* SGDSolver::ComputeUpdateValue (src/caffe/solver.cpp:485-531): rate from the lr policy,
   * momentum, weight_decay, per-blob blobs_lr / weight_decay multipliers {W, b}, regulariser. */
  float lr, momentum, weight_decay;
  float lr_mult[2], decay_mult[2];
  int32_t reg;
  /* SolverParameter.solver_type (caffe.proto: SGD = 0, NESTEROV = 1, ADAGRAD = 2; GetSolver, solver.hpp:128-143):
   * NesterovSolver / AdaGradSolver::ComputeUpdateValue (solver.cpp:599-655, 714-781).  delta = AdaGrad's
   * stability constant (SolverParameter.delta, default 1e-8); AdaGrad requires momentum == 0 (solver.hpp:121). */
  int32_t solver_type;
  float delta;
  /* InnerProductParameter.regularization (inner_product_layer.cpp:80-90): the weight gradient is scaled by
   * 1 + regularization / 2 when regularization > 0.  0 = the shipped files' default. */
  float ip_regularization;
  /* MAX_MARGIN_LOSS's optional third bottom (max_margin_loss_layer.cpp:60-62, 82-97, 152-161, 173-186): a
   * non-negative weight per loss term, here one per batch item (the reference replicates a per-item blob over the
   * Nn terms with a SUM layer): L2 loss term w*max(0,m-d)^2, L1 term w*max(0,m-d).  Either the weights themselves
   * (use_direct_weight) or the values looked up from id_to_weight_file by the caller.  Host [B]; NULL = unweighted. */
  const float* item_weight;
} vv_step_cfg;

/* Defaults of the shipped project files (mednet_embedding_train.prototxt:195-198,655-671,
 * mednet_embedding_train_solver.prototxt:12-14): margin 2, L2, loss_weight 1, no dropout,
 * momentum .9, weight_decay 5e-4, lr_mult {1,2}, decay_mult {1,0}, L2 regulariser. */
void vv_step_cfg_default(vv_step_cfg* cfg);

/* Net::ForwardBackward (include/caffe/net.hpp:78-83) over the videovec TRAIN graph
 * (projects/videovec_embedding/mednet_embedding_train.prototxt:1-671):
 * gather -> fc7 -> ReLU(-> dropout) -> context mean -> L2 normalise -> dot-product scores ->
 * max-margin loss, and its backward down to the fc7 parameter gradients.
 * idx: int32 [B][C+Nn] table rows exactly as the data layer lays out its top blob
 * (video_sampled_shots_data_layer.cpp:214-220: channel 0 target, 1..C-1 context, C.. negatives),
 * -1 = all-zero row.  idx_on_device: 0 = host array; 1 = device pointer whose contents were produced on the context's
 * stream (vv_set_stream): every kernel of the step, the index grouping on the library's second stream included, is
 * ordered after the work already queued there; 2 = device pointer whose contents are complete when the call is made
 * (no ordering is added: static, pre-synchronised batches). */
int vv_forward_backward(vv_ctx* ctx, const vv_step_cfg* cfg, const int32_t* idx, int idx_on_device);
/* The same with the reference's quirk Q1 honoured: a same-video negative (max_same_video_negs > 0)
 * is copied WITHOUT its last feature (video_sampled_shots_data_layer.cpp:492), so that element of
 * the slot keeps what the previous batch left there.  last_src: int32 [B][C+Nn] = the table row
 * whose LAST feature each slot holds (-1 = zero), as vv_sampler_next reports it; both host arrays. */
int vv_forward_backward_q1(vv_ctx* ctx, const vv_step_cfg* cfg, const int32_t* idx,
                           const int32_t* last_src);
/* SGDSolver::ComputeUpdateValue + Net::Update (solver.cpp:485-531, net.cpp:803-839,
 * blob.cpp:112-136) on the gradients of the last vv_forward_backward.
 * (Without a communicator the last stage of the backward pass -- the sum of the weight gradient's split-K partials -- is
 * deferred: called right after vv_forward_backward, vv_apply_update reduces and updates in one launch; vv_loss_get,
 * vv_grads_get, vv_grads_device and vv_grads_bind run the reduction first if it is still due, and once vv_grads_device has
 * handed the buffer out every vv_forward_backward ends with it.  Results are bit for bit
 * those of the eager order; nothing observable through this interface changes.) */
int vv_apply_update(vv_ctx* ctx, const vv_step_cfg* cfg);
/* Both of the above: one iteration of Solver::Solve's loop body (solver.cpp:194,219-220). */
int vv_step(vv_ctx* ctx, const vv_step_cfg* cfg, const int32_t* idx, int idx_on_device);
/* Solver::Step as ONE unit (solver.cpp:177-221: ForwardBackward, ComputeUpdateValue, Update back to back): announces that the NEXT
 * vv_forward_backward* call will be followed by vv_apply_update with exactly these solver parameters (lr, momentum, weight_decay, lr_mult,
 * decay_mult, reg, solver_type, delta) and that nothing reads the gradient in between.  The library may then apply the update where the
 * gradient is produced (option "wgrad_update", on by default): when the weight-gradient GEMM runs with one split of K (large D x F, small batches -- the shipped
 * mednet_embedding_train.prototxt: 4096 x 4096, batch 128) its epilogue applies the solver's rule to the tile it holds, and the 4 D F bytes
 * of dW are neither written nor read back; vv_apply_update then only finishes the step (bias, loss, scale bookkeeping).  Parameters, history
 * and losses are bit for bit those of the un-hinted calls.  Between such a vv_forward_backward* and its vv_apply_update only vv_loss_get
 * is allowed (vv_grads_*, vv_params_get, another forward pass: VV_ERR_STATE -- the gradient of such a step is not kept, the parameters are
 * half-way).  Where the fusion does not apply (several splits of K, a communicator, a bound gradient buffer) the hint changes nothing.
 * vv_step announces itself.  One hint covers one step. */
int vv_update_hint(vv_ctx* ctx, const vv_step_cfg* cfg);

/* Loss (already multiplied by loss_weight) and train_violations of the last forward; synchronises.
 * (the two tops of MAX_MARGIN_LOSS, max_margin_loss_layer.cpp:111-126.) */
int vv_loss_get(vv_ctx* ctx, float* loss, float* violations);

/* ---- gradients.  The flat fp32 device buffer [dW (D*F) | db (D)] that a data-parallel host
 * all-reduces (RCCL) between vv_forward_backward and vv_apply_update; there is no counterpart in
 * the reference (single GPU). */
int vv_grads_device(vv_ctx* ctx, void** dev_ptr, int64_t* n_floats);
/* Make the context write its gradients into caller-owned device memory of D*F + D floats (e.g. a
 * tensor of the collective library); NULL returns to the context's own buffer.  Takes effect for the
 * kernels launched afterwards and does not synchronise, so a host can alternate two buffers: one being
 * all-reduced while the next iteration's gradients are written to the other. */
int vv_grads_bind(vv_ctx* ctx, void* dev_ptr);
/* ---- data parallel: one process per GPU, the flat gradient buffer summed over the ranks between the backward pass
 * and the update.  No counterpart in the reference (single device: Caffe::SetDevice, common.cpp:127-145).
 *   vv_comm_init      after vv_params_set.  transport VV_COMM_RCCL: RCCL (loaded at this call; the copy a host framework
 *                     already loaded is reused); rank 0 writes the communicator id to the file id_path, the others wait
 *                     for it.  VV_COMM_SHM: a host-staged all-reduce through POSIX shared memory named after id_path, for
 *                     tests of the multi-rank path on a box with one device.  VV_COMM_PEER: the one-shot DIRECT exchange of one
 *                     node (<= 16 ranks): every rank's gradient and parameter buffers are mapped into every other rank (hipIpc, the
 *                     handles passed through a shared-memory object named after id_path), a reduce-scatter is one kernel that reads
 *                     this rank's shard of all N buffers over xGMI and adds them in rank order, an all-gather one kernel that pulls
 *                     the foreign shards; the ranks meet at flag words in host-coherent memory, the host is not in the loop.
 *                     Bit for bit the sums of VV_COMM_SHM; also works between processes that share one device.  Pair it with cfg.global_count = the
 *                     global B * Nn, and give every rank its items of the same global batch (vv_batch_ring_next).
 *   vv_allreduce_grads  sums [dW | db] over the ranks on the context's communication stream and makes the compute stream
 *                     wait for it (the host does not block with RCCL).  vv_apply_update calls it when the caller has not.
 *   vv_comm_overlap   on: exact synchronous SGD with the exchange hidden behind the NEXT step's forward pass.  The gradient
 *                     buffer is laid out chunk-major (a few column blocks of dW, each one contiguous message; vv_grads_get
 *                     still returns the blob's row-major D x F); vv_apply_update queues, per chunk, all-reduce -> SGD on the
 *                     chunk's columns -> publish on the communication stream and returns; the next vv_forward_backward starts
 *                     its forward GEMM at once, and the kernel waits for each chunk of W where its K loop reaches it
 *                     (InnerProductLayer::Forward reads every column of W: inner_product_layer.cpp:60-69).  Same sums,
 *                     same update as without overlap; every other entry point that touches the parameters first orders
 *                     itself behind the update in flight.  vv_allreduce_grads is a no-op in this mode. */
enum { VV_COMM_RCCL = 0, VV_COMM_SHM = 1, VV_COMM_PEER = 2 };
int vv_comm_init(vv_ctx* ctx, int32_t world, int32_t rank, const char* id_path, int32_t transport);
int vv_comm_overlap(vv_ctx* ctx, int on);
/* The three exchange schedules by number: 0 = sync, 1 = overlap (as vv_comm_overlap), 2 = SHARDED -- reduce-scatter of the gradients
 * (fp32), the solver's rule (SGDSolver::ComputeUpdateValue, solver.cpp:485-531, + Net::Update, net.cpp:803-839) on this rank's
 * D / world rows of W, the history and the bias, all-gather of the 16-bit copy of W + the bias that the next forward pass reads: 3/4 of
 * the all-reduce's bytes on the wire and 1/world of the update's memory traffic; the same parameters as the other schedules, bit for
 * bit in what the forward pass reads.  The fp32 master W and the history are then complete on their owner only: vv_params_get (and
 * vv_comm_destroy, and leaving the schedule) gather them and are COLLECTIVE calls -- every rank makes them, as every rank of
 * `caffe train` does at a snapshot.  Needs D divisible by world (else the step falls back to sync). */
int vv_comm_schedule(vv_ctx* ctx, int schedule);
int vv_allreduce_grads(vv_ctx* ctx);
int vv_comm_destroy(vv_ctx* ctx);
/* InnerProductLayer blobs_[0]/[1] cpu_diff() after Backward (inner_product_layer.cpp:76-98). */
int vv_grads_get(vv_ctx* ctx, float* dW, float* db);

/* ---- inspection of named blobs of the last forward (Net::blob_by_name, net.cpp:846-857);
 * every output optional (NULL).  Host fp32, in the reference's layouts:
 *   ip2          [(C+Nn)*B][D]  row = ch*B + b   (after ReLU / dropout)
 *   target_score [B][Nn], negative_scores [B][Nn]
 *   ip2_diff     [(C+Nn)*B][D]  the diff of ip1_nonorm */
int vv_blobs_get(vv_ctx* ctx, float* ip2, float* target_score, float* negative_scores,
                 float* ip1_diff);

/* ---- inference: fc7 (+ReLU) (+L2 normalise) of arbitrary table rows -- the extract_features path
 * (tools/extract_features.cpp:99-198 over videovec_extraction.prototxt:179-205) and the TEST
 * branch's embedding (mednet_embedding_train.prototxt:344-352).  rows host int32 [n] (NULL = 0..n-1),
 * out host fp32 [n][D]. */
int vv_embed(vv_ctx* ctx, const int32_t* rows, int64_t n, int relu, int l2norm, float* out);

/* The TEST branch's input: each sample is the coefficient-weighted sum of k table rows
 * (SLICE/CONCAT/SLICE + ELTWISE SUM `average_for_test`, mednet_embedding_train.prototxt:75-177), then
 * fc7 -> ReLU -> optional NORMALIZATION (`test_norm`, :344-352).  rows: host int32 [n][k];
 * coeff: host [k] or NULL for 1/k; out: host fp32 [n][D]. */
int vv_embed_mean(vv_ctx* ctx, const int32_t* rows, int64_t n, int32_t k, const float* coeff, int relu,
                  int l2norm, float* out);

/* RetrievalStatsLayer::Forward_cpu, per-shot retrieval (src/caffe/layers/retrieval_stats_layer.cpp:
 * 104-141, 143-355): distance = -2 X X^T (computed on the GPU), self excluded, ascending sort,
 * mean AP / hit@1 / hit@5 over samples whose class is >= 0.  feat: host fp32 [n][dim]; video_ids [n];
 * the id_to_class_file as two parallel arrays (ids absent from it read as class 0, as the
 * reference's map operator[] does).  Equal distances are ordered by ascending index (std::sort
 * leaves them unspecified; this order reproduces test_retrieval_stats_layer.cpp:82-84).
 * video_level_retrieval and stats_output_file are not built. */
int vv_retrieval_stats(vv_ctx* ctx, const float* feat, int32_t n, int32_t dim, const int32_t* video_ids,
                       const int32_t* map_ids, const int32_t* map_cls, int32_t n_map,
                       int exclude_same_video_shots, float* mean_ap, float* hit_at_1, float* hit_at_5);

/* ---- per-layer operators.  The reference's operator interface is Layer<Dtype>::{Forward_gpu, Backward_gpu}
 * (include/caffe/layer.hpp:308-337); its sequential executor (Net::ForwardFromTo / BackwardFromTo, net.cpp:501-578) calls
 * them layer by layer.  Training here runs the fused plan above; these entry points are what the C++ facade's layer classes
 * call when a graph is executed layer by layer (a graph the fused-plan matcher does not recognise, gradient checks, a
 * single Layer::Forward).  All buffers are fp32 DEVICE memory of this context (vv_dev_alloc; SyncedMemory's GPU side,
 * src/caffe/syncedmem.cpp:55-109), row-major; everything is queued on the context's stream.
 *   vv_op_copy2d      rows x cols block copy between strided buffers, optionally accumulating: SLICE / CONCAT forward and
 *                     backward (slice_layer.cu:10-64, concat_layer.cu:10-75), SPLIT's diff sum (split_layer.cu:18-33)
 *   vv_op_axpby       y = a x + b y: ELTWISE SUM with coefficients (eltwise_layer.cu:40-45, 99-105), diff accumulation
 *   vv_op_mul         y (+)= a .* b: ELTWISE PROD forward and its stable-product backward (eltwise_layer.cu:36-38, 83-97)
 *   vv_op_relu(_bwd)  relu_layer.cu:10-59, with negative_slope
 *   vv_op_dropout     dropout_layer.cu:14-73: make_mask = 1 draws the mask (counter-based hash of seed and element index)
 *                     and applies it; make_mask = 0 applies an existing mask -- the TRAIN backward, y = dy, with the same mask
 *   vv_op_rowsum(_bwd)     SUM (sum_layer.cu:10-55): row sums replicated num_output times; backward broadcasts
 *   vv_op_normalize(_bwd)  NORMALIZATION (normalization_layer.cu:10-97), eps placement of the reference (quirk Q6)
 *   vv_op_max_margin(_bwd) MAX_MARGIN_LOSS (max_margin_loss_layer.cpp:53-214; CPU-only in the reference): weight NULL or
 *                     one non-negative weight per term; forward synchronises and returns loss / violations to the host
 *   vv_op_gather_rows the data layer's batch copy (base_data_layer.cu:7-21): table rows idx (host int32 [n], -1 = zeros) as fp32
 *   vv_op_inner_product(_bwd)  INNER_PRODUCT with the context's parameters (inner_product_layer.cu:12-59): Y = X W^T + b;
 *                     backward writes dW (scaled by 1 + regularization / 2 when > 0) and db into the flat gradient buffer,
 *                     from which vv_apply_update updates.  No gradient w.r.t. X (the layer sits on the data layer). */
int vv_dev_alloc(vv_ctx* ctx, size_t bytes, void** out);     /* zero-filled */
int vv_dev_free(vv_ctx* ctx, void* p);
int vv_dev_upload(vv_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int vv_dev_download(vv_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int vv_dev_memset(vv_ctx* ctx, void* dst_dev, int value, size_t bytes);
int vv_op_copy2d(vv_ctx* ctx, const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows, int64_t cols,
                 int accumulate);
int vv_op_axpby(vv_ctx* ctx, int64_t n, float a, const float* x, float b, float* y);
int vv_op_mul(vv_ctx* ctx, int64_t n, const float* a, const float* b, float* y, int accumulate);
int vv_op_relu(vv_ctx* ctx, int64_t n, const float* x, float* y, float negative_slope);
int vv_op_relu_bwd(vv_ctx* ctx, int64_t n, const float* x, const float* dy, float* dx, float negative_slope);
int vv_op_dropout(vv_ctx* ctx, int64_t n, const float* x, float* y, uint8_t* mask, float ratio, uint64_t seed, int make_mask);
int vv_op_rowsum(vv_ctx* ctx, int64_t rows, int32_t cols, const float* x, int32_t num_output, float* y);
int vv_op_rowsum_bwd(vv_ctx* ctx, int64_t rows, int32_t cols, int32_t num_output, const float* dy, float* dx);
int vv_op_normalize(vv_ctx* ctx, int64_t rows, int32_t cols, const float* x, float* y);
int vv_op_normalize_bwd(vv_ctx* ctx, int64_t rows, int32_t cols, const float* x, const float* dy, float* dx);
int vv_op_max_margin(vv_ctx* ctx, int32_t count, const float* s_true, const float* s_bogus, const float* weight, float margin,
                     int32_t norm, float* loss, float* violations);
int vv_op_max_margin_bwd(vv_ctx* ctx, int32_t count, const float* s_true, const float* s_bogus, const float* weight, float margin,
                         int32_t norm, float loss_weight, float* d_true, float* d_bogus);
int vv_op_gather_rows(vv_ctx* ctx, const int32_t* idx, int64_t n, float* out);
int vv_op_inner_product(vv_ctx* ctx, const float* X, int64_t R, float* Y);
int vv_op_inner_product_bwd(vv_ctx* ctx, const float* dY, int64_t R, float ip_regularization);

/* ---- triplet sampler (host side, integer only).  Replaces VideoSampledShotsDataLayer's
 * DataLayerSetUp / AddSamplesToTop / InternalThreadEntry / AddToBuffer / RandomShuffleTopids
 * (src/caffe/layers/video_sampled_shots_data_layer.cpp:24-44,64-369,371-507,768-909) with the
 * feature copies replaced by table-row indices.  The draw order of the reference's libc rand()
 * stream (never seeded => glibc seed 1), include/caffe/util/rng.hpp:43-54's random_unique and
 * libstdc++'s std::random_shuffle are reproduced exactly, so indices are bit-identical.
 * One DB record = one video: record v has video_id[v], n_shots[v] frames whose features are table
 * rows row_base[v] .. row_base[v]+n_shots[v]-1, and shot ids shot_ids[shot_off..] (NULL = 0..n-1).
 * context_type: see VV_CONTEXT_* below. */
typedef struct vv_sampler vv_sampler;
typedef struct {
  int32_t batch_size, context_size, num_negative_samples;
  int32_t max_buffer_size, negative_swap_percentage, max_same_video_negs;
  int32_t max_tries_for_negs;      /* gflag --max_tries_for_negs, default 100 (...data_layer.cpp:20) */
  int32_t context_type;            /* VV_CONTEXT_* (VideoSampledShotsDataParameter.ContextType) */
  int32_t initial_cursor;          /* records skipped before anything else (rand_skip, ...data_layer.cpp:156-180): the DB
                                      cursor starts at this record (modulo the record count) */
  int32_t output_shot_distance;    /* PAIRWISE only (...data_layer.cpp:71, 407-418): label = frame distance, not video id */
  float   max_shot_distance;       /* ... clamped to this bound (caffe.proto:674, default 5) */
  int32_t rand_seed;               /* srand() argument of the draw stream.  0 or 1 = the reference's (it never seeds: glibc's seed
                                      1).  Other values (< 2^31 - 1) give each data-parallel rank its own stream (per-rank
                                      samplers); the stream is still glibc's rand() after srand(rand_seed) */
} vv_sampler_param;
/* WINDOW (...data_layer.cpp:425-507): target = the middle of C sorted random frames.  PAST (:510-596): target = the
 * last of C sorted random frames.  PAST_CONTINUOUS (:599-674): C equally spaced frames, random stride and start,
 * target = the last.  PAST_CONTINUOUS_FIXED (:677-757): the same with the largest stride minus one, ending at the
 * video's end.  PAIRWISE (:396-422): two distinct random frames in draw order (target, then the one context frame),
 * context_size forced to 2 (:200-201), records with one shot skipped; with output_shot_distance the label is their
 * distance.  All fill the same (B, C+Nn) layout. */
enum { VV_CONTEXT_WINDOW = 0, VV_CONTEXT_PAST = 1, VV_CONTEXT_PAST_CONTINUOUS = 2, VV_CONTEXT_PAST_CONTINUOUS_FIXED = 3,
       VV_CONTEXT_PAIRWISE = 4 };
void vv_sampler_param_default(vv_sampler_param* p);
int vv_sampler_create(const vv_sampler_param* p, int32_t n_videos, const int32_t* video_id,
                      const int32_t* n_shots, const int64_t* row_base, const int32_t* shot_ids,
                      vv_sampler** out);
/* The same with VideoSampledShotsDataParameter.negative_dataset (...data_layer.cpp:105-151, 253-286, 325-341): the
 * negative buffer starts as EVERY shot of the negative dataset's records taken in order (no rand() draw, the main
 * cursor stays at initial_cursor) and must come out exactly full -- the reference tests the fill level only after a
 * whole record and overruns its buffer otherwise, so anything but an exact fit is VV_ERR_ARG.  neg_row_base index
 * the same feature table as row_base.  neg_videos == 0 is vv_sampler_create. */
int vv_sampler_create_neg(const vv_sampler_param* p, int32_t n_videos, const int32_t* video_id,
                          const int32_t* n_shots, const int64_t* row_base, const int32_t* shot_ids,
                          int32_t neg_videos, const int32_t* neg_video_id, const int32_t* neg_n_shots,
                          const int64_t* neg_row_base, const int32_t* neg_shot_ids, vv_sampler** out);
/* One prefetch batch.  idx / last_src: int32 [batch_size][context_size + num_negative_samples]
 * (last_src: the row whose LAST feature the slot holds -- differs from idx only for same-video
 * negatives, which the reference copies without their last element, ...data_layer.cpp:492; -1 =
 * zero); label: int32 [batch_size] video ids (...:879).  Any output may be NULL. */
int vv_sampler_next(vv_sampler* s, int32_t* idx, int32_t* last_src, int32_t* label);
int vv_sampler_destroy(vv_sampler* s);

/* ---- prefetch.  Replaces BasePrefetchingDataLayer::{CreatePrefetchThread, JoinPrefetchThread}
 * (src/caffe/layers/base_data_layer.cpp:52-95) and InternalThread (src/caffe/internal_thread.cpp:14-37): the
 * reference starts one thread per batch that fills prefetch_data_ while the solver consumes the previous one.
 * Here background threads run `depth` batches ahead of the consumer; afterwards vv_sampler_next pops finished
 * batches in order (the index stream is exactly the one vv_sampler_next would have produced by itself).
 *   threads   1 = whole batches on one thread; 2, 3 or 4 = the item's three chains (stream walk + buffer swap-in /
 *             negative-slot draw / frame draw) as a pipeline of threads, the fourth generating the rand() stream a
 *             block ahead of the walk (same indices; only for samplers without same-video negatives whose
 *             (video_id, shot_id) keys name distinct rows, else 1 is used).
 *   shm_name  NULL = a private ring.  A name = the ring lives in a POSIX shared-memory object of that name, so that
 *             ONE sampler per node serves `consumers` processes (the data-parallel ranks: every rank takes its
 *             items of the same global batch) -- see vv_batch_ring_attach.
 *   consumers number of readers that must each take (a slice of) a batch before its buffer is reused. */
int vv_sampler_prefetch_start(vv_sampler* s, int32_t depth, int32_t threads, const char* shm_name, int32_t consumers);
int vv_sampler_prefetch_stop(vv_sampler* s);
/* Counters for tests and tuning: which = 0 swap-in walks that had to restart (a swap-in evicted a later shot of the same
 * video), 1 = the staged fast path is in use (0/1), 2 = producer threads of the running prefetch, 3 = time-stamp-counter
 * ticks since the pipeline started, 4 / 5 / 6 = ticks of them the walk / negative-slot / frame stage spent waiting,
 * 7 = the 512-bit forms of the walk and the slot draw are in use (0/1: the host has AVX-512 F/BW/DQ/VL/VBMI2 and
 * VV_SAMPLER_AVX512 is not 0; same indices either way), 8 = ticks the walk stage waited for the stream-generating
 * thread (four-thread pipeline), 9 = cores held for the stage threads (4: one each, claimed against other samplers
 * of the host through /dev/shm/vv_sampler_cpu_<n> locks; 0: the threads share the caller's group of CPUs).  -1 = unknown. */
int64_t vv_sampler_stat(vv_sampler* s, int32_t which);

/* A reader's view of a sampler's batch ring.  vv_sampler_ring: the producer process's own handle (owned by the
 * sampler).  vv_batch_ring_attach: map the named ring of another process (waits up to timeout_s for it to appear).
 * vv_batch_ring_next: wait for this consumer's next batch (timeout_s <= 0: forever; VV_ERR_STATE when the producer
 * has gone or the wait timed out), copy items [item_begin, item_begin + item_count) -- idx int32 [item_count][C+Nn],
 * label int32 [item_count], either may be NULL -- and release the batch for this consumer. */
typedef struct vv_batch_ring vv_batch_ring;
int vv_sampler_ring(vv_sampler* s, vv_batch_ring** out);
int vv_batch_ring_attach(const char* shm_name, double timeout_s, vv_batch_ring** out);
int vv_batch_ring_info(vv_batch_ring* r, int32_t* batch_size, int32_t* slots_per_item, int32_t* consumers, int32_t* depth);
int vv_batch_ring_next(vv_batch_ring* r, int32_t consumer, int32_t item_begin, int32_t item_count, int32_t* idx,
                       int32_t* label, double timeout_s);
int vv_batch_ring_detach(vv_batch_ring* r);
/* BasePrefetchingDataLayer::Forward_gpu (src/caffe/layers/base_data_layer.cu:7-21: join the prefetch thread, copy the
 * batch to the device) + Net::ForwardBackward: take this consumer's next batch out of the ring -- items
 * [item_begin, item_begin + cfg->B) of it -- send the indices to the device through a pinned staging buffer with an
 * asynchronous copy on the context's stream, and queue vv_forward_backward on them.  label_out: host int32 [B] or NULL. */
int vv_forward_backward_ring(vv_ctx* ctx, const vv_step_cfg* cfg, vv_batch_ring* ring, int32_t consumer,
                             int32_t item_begin, int32_t* label_out, double timeout_s);

/* Timing hook for the benchmark: average device time in ms of one named kernel ("fwd_gemm",
 * "score_loss", "wgrad_gemm", "reduce", "sgd") over the launches since the last reset, measured
 * with hipEvents on the context's stream (only while enabled; enabling adds two event records
 * per launch). */
/* on: 0 = off, 1 = every step, N > 1 = the N-th, 2N-th, ... step after the call (a step ends with vv_apply_update).  The events ride on the
 * kernels' own dispatch packets (hipExtLaunchKernelGGL); no extra packets are queued. */
int vv_profile_enable(vv_ctx* ctx, int on);
/* Restrict the timing to a comma-separated list of kernel names (NULL or "" = all).  A timed dispatch cannot be pipelined
 * behind its predecessor (~5 us each): the benchmark's timed leg times the two GEMMs only. */
int vv_profile_select(vv_ctx* ctx, const char* kernels);
int vv_profile_get(vv_ctx* ctx, const char* kernel, double* avg_ms, int64_t* launches);

/* Box calibration for the benchmark: what THIS device delivers at the moment of the call, so that step times measured on different boxes of
 * a pool can be compared (the step's GEMMs run at whatever clock the chip holds under them).  No reference counterpart: `caffe device_query`
 * (tools/caffe.cpp:108-121, Caffe::DeviceQuery, src/caffe/common.cpp:147-180) prints static properties only.  Two fixed probes on buffers of
 * their own (2.2 GB, released before the call returns), queued on the context's stream, ~10 ms of device time:
 *   gemm  the benchmark's forward instantiation (f16 operands, 192-row tiles, 216 workgroups) on CONTIGUOUS rows of a random table,
 *         20 736 x 4096 x 512, operands uniform in [-1, 1): TFLOP/s over 24 back-to-back launches behind 8 warm-up launches, and the shader
 *         clock held inside the kernel (s_memtime over s_memrealtime, median over its workgroups)
 *   copy  a 1 GiB device-to-device streaming copy: bytes read + written per second over 6 copies behind 2 */
typedef struct {
  double gemm_tflops, gemm_ms, gemm_clock_mhz;
  double copy_tbs, copy_ms;
  int32_t gemm_rows, gemm_k, gemm_n, gemm_launches;
  int64_t copy_bytes;
} vv_box_probe_result;
int vv_box_probe(vv_ctx* ctx, vv_box_probe_result* out);

#ifdef __cplusplus
}
#endif
#endif
