/*
 * vv_oracle.h -- CPU restatement of the reference's videovec_embedding training path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under videovector_amd/ (the product) may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker / the reported CPU baseline.
 *
 * Parity pin status (see DESIGN.md "Oracle"): the reference cannot be built in this image
 * (needs protoc-generated caffe.pb.h, glog, gflags, boost, cblas, lmdb, leveldb -- all absent),
 * and its own tests for this path hold no stored tensors, only properties and one known-answer
 * test.  The oracle is pinned against
 *   - the real glibc rand() and libstdc++ std::random_shuffle of this image (the two third-party
 *     algorithms the sampler delegates to; tests/test_oracle_rng.py),
 *   - every property / known-answer assertion of the reference's unit tests for the layers on the
 *     path (tests/test_oracle_reference_kats.py cites each test file:line).
 *   - the WIRING of the assembled graph (orc_forward_backward) against the reference's own project file: the TRAIN-phase topology of
 *     projects/videovec_embedding/mednet_embedding_train.prototxt (tests/golden/mednet_train_graph.json: 37 layers, their types,
 *     bottoms, tops and parameters) is executed layer by layer with the layer functions above and must give the assembled step's loss,
 *     scores, ip2, dW and db (tests/test_oracle_graph_topology.py).
 * The SAMPLER is NOT pinned by any reference fixture ("parity unpinned" for it: the reference has no test and no recorded batch for
 * it); it is cross-checked against an independent restatement driven by the real glibc rand() (tests/pyref.py).
 *
 * All citations are relative to /root/reference.
 */
#ifndef VV_ORACLE_H_
#define VV_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ RNG (third party) ------ */
/* glibc 2.35 random_r() TYPE_3 additive-feedback generator == rand().  The reference never calls
 * srand, so the stream is the seed-1 stream (video_sampled_shots_data_layer.cpp:27,29,306;
 * include/caffe/util/rng.hpp:48). */
typedef struct { int32_t r[31]; int f, b; } orc_rng;
void    orc_srand(orc_rng* g, unsigned seed);
int32_t orc_rand(orc_rng* g);
/* include/caffe/util/rng.hpp:43-54 */
void orc_random_unique(orc_rng* g, int32_t* a, int len, int n);
/* libstdc++ 11 bits/stl_algo.h std::random_shuffle(first,last) (two-argument form, rand()-based);
 * call site video_sampled_shots_data_layer.cpp:482 */
void orc_random_shuffle(orc_rng* g, int32_t* a, int len);

/* ------------------------------------------------------------------ sampler ---------------- */
/* One DB record == one video (proto VideoShots, src/caffe/proto/video_shot_sentences.proto:15-20):
 * video_id, shot_ids[], shot_words[] (one feature row each).  Feature rows live in a row table;
 * record v's shot j is table row row_base[v] + j. */
typedef struct {
  int32_t        n_videos;
  const int32_t* video_id;   /* [n_videos] */
  const int32_t* n_shots;    /* [n_videos] */
  const int64_t* row_base;   /* [n_videos] */
  const int32_t* shot_ids;   /* concatenated per record, or NULL => 0..n_shots-1 */
  const int64_t* shot_off;   /* [n_videos] offset of record v in shot_ids (ignored if NULL ids) */
} orc_dataset;

typedef struct {
  int32_t batch_size, context_size, num_negative_samples;
  int32_t max_buffer_size, negative_swap_percentage, max_same_video_negs;
  int32_t max_tries_for_negs;   /* gflag, default 100 (…data_layer.cpp:20) */
  int32_t context_type;         /* VideoSampledShotsDataParameter.ContextType */
  int32_t initial_cursor;       /* rand_skip (…data_layer.cpp:156-180): records skipped before the buffer is filled */
  int32_t output_shot_distance; /* PAIRWISE only (…data_layer.cpp:71, 407-418): the label is the frame distance */
  float   max_shot_distance;    /* ... clamped to this (caffe.proto:674, default 5) */
} orc_sampler_param;
/* PAIRWISE (…data_layer.cpp:396-422): two random frames of the record, context_size forced to 2 (:200-201) */
enum { ORC_CONTEXT_WINDOW = 0, ORC_CONTEXT_PAST = 1, ORC_CONTEXT_PAST_CONTINUOUS = 2, ORC_CONTEXT_PAST_CONTINUOUS_FIXED = 3,
       ORC_CONTEXT_PAIRWISE = 4 };

typedef struct orc_sampler orc_sampler;
/* DataLayerSetUp (…data_layer.cpp:64-369): fills the negative buffer from the cursor position.
 * Returns NULL if the reference would CHECK-fail (…:344, :207, :434). */
orc_sampler* orc_sampler_create(const orc_dataset* ds, const orc_sampler_param* p, unsigned seed);
/* The same with `negative_dataset` set (…data_layer.cpp:105-151, 253-286, 325-341): the buffer is filled with EVERY
 * shot of the negative dataset's records taken in order (no rand(), the main cursor stays where it is) and must
 * come out exactly full (:344).  neg's row_base index the same row table as ds's. */
orc_sampler* orc_sampler_create_neg(const orc_dataset* ds, const orc_dataset* neg, const orc_sampler_param* p,
                                    unsigned seed);
void         orc_sampler_destroy(orc_sampler* s);
/* InternalThreadEntry (…data_layer.cpp:768-909): one batch.
 *   idx      [B][C+Nn]  table row held by each prefetch slot (channel 0 target, 1..C-1 context in
 *                       temporal order, C.. negatives), -1 = slot never written (zeros)
 *   last_src [B][C+Nn]  table row whose LAST feature the slot holds (differs from idx only for
 *                       same-video negatives, which copy F-1 values: …:492 -- quirk Q1)
 *   label    [B]        video_id (…:879)
 * Any of the three may be NULL. */
void orc_sampler_next(orc_sampler* s, int32_t* idx, int32_t* last_src, int32_t* label);
/* inspection for tests: buffer slot -> table row, cursor, number of rand() calls so far */
const int32_t* orc_sampler_buffer_rows(const orc_sampler* s);
const int32_t* orc_sampler_buffer_ids(const orc_sampler* s);
int32_t        orc_sampler_cursor(const orc_sampler* s);
int64_t        orc_sampler_rand_calls(const orc_sampler* s);

/* ------------------------------------------------------------------ BLAS stand-in ---------- */
/* Row-major C = alpha*op(A)*op(B) + beta*C  (math_functions.cpp:12-21 delegates to cblas_sgemm;
 * which BLAS is a build choice of the reference, so summation order is not specified). */
void orc_sgemm(int transA, int transB, int M, int N, int K, float alpha,
               const float* A, const float* B, float beta, float* C);
/* optional external BLAS for orc_sgemm (shared object exporting cblas_sgemm); NULL / "" = the built-in kernel */
int orc_set_blas(const char* path);
int orc_has_blas(void);
void orc_set_threads(int n);   /* 0 = all cores */
int  orc_get_threads(void);

/* ------------------------------------------------------------------ layers ----------------- */
/* normalization_layer.cpp:29-61 / 63-112 */
void orc_normalize_fwd(int num, int dim, const float* x, float* y);
void orc_normalize_bwd(int num, int dim, const float* x, const float* dy, float* dx);
/* max_margin_loss_layer.cpp:53-127 / 129-214.  norm: 1 = L1, 2 = L2.  weight may be NULL
 * (third bottom, use_direct_weight). */
void orc_max_margin_fwd(int count, const float* s_true, const float* s_bogus, const float* weight,
                        float margin, int norm, float* loss, float* violations);
void orc_max_margin_bwd(int count, const float* s_true, const float* s_bogus, const float* weight,
                        float margin, int norm, float loss_weight, float* d_true, float* d_bogus);
/* sum_layer.cpp:31-54 / 56-82 */
void orc_sum_fwd(int num, int dim, int num_output, const float* x, float* y);
void orc_sum_bwd(int num, int dim, int num_output, const float* dy, float* dx);
/* The plain layers of the graph, one function per Forward_cpu / Backward_cpu; orc_forward_backward is assembled
 * from these, so the reference's per-layer tests restated in tests/test_oracle_reference_kats.py pin the code the
 * whole-step oracle runs. */
/* relu_layer.cpp:10-20 / 23-37 */
void orc_relu_fwd(int64_t n, const float* x, float slope, float* y);
void orc_relu_bwd(int64_t n, const float* x, const float* dy, float slope, float* dx);
/* dropout_layer.cpp:34-50 / 52-68; mask 1 = keep, train 0 = TEST phase (identity) */
void orc_dropout_fwd(int64_t n, const float* x, const uint8_t* mask, float ratio, int train, float* y);
void orc_dropout_bwd(int64_t n, const float* dy, const uint8_t* mask, float ratio, int train, float* dx);
/* eltwise_layer.cpp:53-105 / 108-159; op 0 PROD, 1 SUM, 2 MAX */
void orc_eltwise_fwd(int op, int64_t n, int nb, const float* const* bottom, const float* coeff, float* top);
void orc_eltwise_bwd(int op, int64_t n, int nb, const float* const* bottom, const float* coeff,
                     const float* top, const float* dtop, int which, int stable, float* dbottom);
/* slice_layer.cpp:79-135, concat_layer.cpp:45-117 (dims 0 and 1) */
void orc_split_pieces(int outer, int64_t inner, int npieces, const int32_t* width, const float* whole,
                      float* const* piece);
void orc_join_pieces(int outer, int64_t inner, int npieces, const int32_t* width, const float* const* piece,
                     float* whole);
/* split_layer.cpp:36-51 */
void orc_split_bwd(int64_t n, int ntop, const float* const* dtop, float* dbottom);
/* inner_product_layer.cpp:61-74 / 76-106; b, dW, db, dX may be NULL */
void orc_inner_product_fwd(int M, int N, int K, const float* X, const float* W, const float* b, float* Y);
void orc_inner_product_bwd(int M, int N, int K, const float* X, const float* W, const float* dY,
                           float* dW, float* db, float* dX);
/* solver.cpp:440-460 */
float orc_learning_rate(const char* policy, float base_lr, float gamma, float power, int stepsize,
                        int iter);
/* solver.cpp:502-531 + blob.cpp:112-136 for one parameter blob; reg: 2 = L2, 1 = L1 */
void orc_sgd_update(int64_t n, float* w, float* grad, float* hist, float rate, float lr_mult,
                    float momentum, float weight_decay, float decay_mult, int reg);
void orc_solver_update(int64_t n, float* w, float* grad, float* hist, float rate, float lr_mult,
                       float momentum, float weight_decay, float decay_mult, int reg, int solver_type,
                       float delta);

/* ------------------------------------------------------------------ whole training step ---- */
typedef struct {
  int32_t B, C, Nn, F, D;
  float   margin; int32_t norm;       /* MaxMarginLossParameter (caffe.proto:858-868) */
  float   loss_weight;                /* top[0] loss weight (layer.hpp:416-422) */
  const float* ctx_coeff;             /* [C-1] eltwise SUM coeffs (prototxt :260-278) */
  float   dropout_ratio;              /* 0 = no dropout layer */
  const uint8_t* dropout_mask;        /* [(C+Nn)*B][D] in channel-major row order, 1 = keep */
  float   relu_negative_slope;        /* 0 */
  float   ip_regularization;          /* InnerProductParameter.regularization (…:80-90) */
  int64_t global_count;               /* 0 => B*Nn; data-parallel shards pass the global B*Nn */
  const float* item_weight;           /* [B] or NULL: MAX_MARGIN_LOSS's 3rd bottom (the weight of item b, replicated
                                         over its Nn terms; max_margin_loss_layer.cpp:82-97, 152-161) */
} orc_step_cfg;

typedef struct {                      /* every pointer optional (NULL = not wanted) */
  float* Y;         /* [(C+Nn)*B][D] ip1_nonorm, channel-major rows (row = ch*B + b) */
  float* H;         /* same shape: ip2 after ReLU (+dropout) */
  float* ctx;       /* [B][D]  context_feature (normalised context mean) */
  float* posneg;    /* [(1+Nn)*B][D] normalised target / negative embeddings, (q*B + b) */
  float* s_true;    /* [B][Nn] target_score */
  float* s_bogus;   /* [B][Nn] negative_scores */
  float* dY;        /* [(C+Nn)*B][D] diff of ip1_nonorm */
  float* dW;        /* [D][F] */
  float* db;        /* [D] */
  float  loss, violations;
} orc_step_out;

/* Forward + backward of the TRAIN graph of projects/videovec_embedding/mednet_embedding_train.prototxt
 * layer by layer (materialised slice/concat copies, sgemm for fc7, ones-vector reductions), i.e.
 * Net::ForwardBackward (net.hpp:78-83).  table is [rows][F] fp32; idx/last_src as produced by the
 * sampler (last_src may be NULL). */
void orc_forward_backward(const orc_step_cfg* cfg, const float* table, const int32_t* idx,
                          const int32_t* last_src, const float* W, const float* b,
                          orc_step_out* out);

/* fc7 + ReLU (+ optional L2 normalise) of arbitrary table rows: the extract_features /
 * TEST-branch embedding (videovec_extraction.prototxt:179-205). */
void orc_embed(int n, int F, int D, const float* table, const int32_t* rows, const float* W,
               const float* b, int relu, int l2norm, float* out);

/* retrieval_stats_layer.cpp:104-141 (ComputeStats), 143-355 (Forward_cpu, per-shot retrieval):
 * distance = -2 X X^T, self set to -1e15, ascending sort, AP / hit@1 / hit@5 over the samples whose
 * class is >= 0.  std::sort leaves the order of equal distances unspecified; ties are broken here by
 * ascending index, which is the order that reproduces the reference's own known-answer test
 * (test_retrieval_stats_layer.cpp:34-39,82-84).  id->class pairs: map_ids / map_cls (ids absent from
 * the map read as class 0, as operator[] of the reference's unordered_map does). */
void orc_retrieval_stats(int n, int dim, const float* feat, const int32_t* video_ids,
                         const int32_t* map_ids, const int32_t* map_cls, int n_map,
                         int exclude_same_video, float* mean_ap, float* hit1, float* hit5);

#ifdef __cplusplus
}
#endif
#endif
