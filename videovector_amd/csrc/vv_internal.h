// vv_internal.h -- shared declarations of the HIP implementation behind include/videovec.h.
// gfx950 (MI355X, CDNA4) only.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace vv {

// Floats between two split-K slabs of the weight gradient.  Exactly D_p x F_p (8 MiB at 512 x 4096) puts the eight streams that k_reduce /
// k_reduce_sgd read side by side a power of two apart, and a reader of the slabs alone gains 11 % from 4 KiB of skew (3.93 -> 4.38 TB/s,
// tools/lab/slab_pitch_lab.hip) -- but in the step k_reduce_sgd's 0.9 us are given back by the weight-gradient GEMM's stores (+0.7 us),
// alternating builds on one box (profiles/r05_slab_pitch.txt): the pad stays 0; -DVV_SLAB_PAD=<floats> builds the other form.
#ifndef VV_SLAB_PAD
#define VV_SLAB_PAD 0
#endif
__host__ __device__ inline int64_t slab_pitch(int Dp, int Fp) { return (int64_t)Dp * Fp + VV_SLAB_PAD; }


// ---------------------------------------------------------------------------------------------
// GEMM tiling shared by the forward (gather-GEMM) and weight-gradient (gather-GEMM^T) kernels.
// One workgroup = 512 threads = 8 waves (2 along M x 4 along N), one 256x256 output tile,
// K advanced 64 at a time through a double-buffered LDS image filled by LDS-DMA
// (global_load_lds_dwordx4).  Each wave owns a 128x64 sub-tile = 8x4 MFMA 16x16x32 accumulators.
// ---------------------------------------------------------------------------------------------
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int GEMM_THREADS = 512;
constexpr int LDS_TILE_BYTES = 256 * 64 * 2;          // one operand tile: 32 KiB
constexpr int GEMM_LDS_BYTES = 4 * LDS_TILE_BYTES;    // {A,B} x 2 buffers = 128 KiB

// Row padding rules of the half-precision operand copies kept in HBM.
constexpr int F_ALIGN = 256;   // table / W row length (K of fwd, N of wgrad)
constexpr int D_ALIGN = 256;   // W rows (N of fwd), dY row length (M of wgrad)
constexpr int R_ALIGN = 256;   // batch rows (M of fwd, K of wgrad)
constexpr int SGD_BLOCKS = 1024;

inline __host__ __device__ int64_t round_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Device-resident scale block: power-of-two factors that keep f16 operands in range.
//   table_h = x * sx          (fixed at table load)
//   W_h     = W * sw_cur      (refreshed by the SGD kernel each step)
//   dY_h    = dY * sg         (sg chosen per step from the loss count)
struct Scales {
  float sx;            // table scale
  float sw_cur;        // scale the current W_h copy carries
  float sw_next;       // scale the next W -> W_h conversion will use
  unsigned wmax_bits;  // running max |W| (float bits) collected by the SGD kernel
};

// f16 gradient-scale guard.  The 16-bit gradient operand of the weight-gradient GEMM carries ONE power-of-two scale per
// step (the host's sg).  A value past f16's range must never reach the GEMM clipped, so every kernel that rounds
// gradients to 16 bits ("producer": k_score_loss / k_score_loss_reg, k_segsum, k_seg_bwd) records its per-block max |g|
// (before rounding) and raises flag[round] when one passes GG_LIMIT; the host then launches the same producers once
// more per guard round as conditional REPEATS: a repeat whose previous round raised no flag returns at once (the normal
// case: one near-empty launch), otherwise it folds the recorded maxima, takes the excess powers of two off the scale
// (max placed at 2^12) and produces its output again.  k_reduce undoes sg * mul.  Exact: powers of two only.
struct GradGuard {
  int32_t flag[4];           // flag[r] == seq: round r of step seq produced a value past GG_LIMIT
  int32_t shift[4];          // shift[r], r = 1, 2: powers of two taken off the host's scale from round r on (cumulative);
                             // shift[3]: the step's final shift
  float mul;                 // 2^-shift of the step's last round
  int32_t repeat_seq;        // the last step in which a conditional repeat really ran
  int32_t pad[2];
};
constexpr float GG_LIMIT = 32768.f;
struct GuardArgs {
  GradGuard* gg = nullptr;   // null: no guard (bf16 operands: fp32's exponent range)
  float* slots = nullptr;    // [3 rounds][2 producers][nslot] per-block max |g| in the units of that round's scale
  int nslot = 0;
  int producer = 0;          // which producer of the round this launch is (0, 1)
  int n_of[2] = {0, 0};      // blocks of producer 0 / 1 (0: that producer does not exist)
  int round = 0;             // 0: the step's normal pass; r >= 1: the r-th conditional repeat
  int last = 0;              // this launch closes the round (writes shift[round]) ...
  int final_round = 0;       // ... and the round is the step's last (publishes mul)
  int32_t seq = 0;
  // PROACTIVE form (the segment-wise backward: k_score_fwd / k_score_stream + k_seg_bwd): no repeat launch.  The score kernel
  // knows every instance's factored gradient (alpha, beta, the vector it multiplies) and bounds its largest element:
  // |alpha| max|Ah_d| + |beta| max|x_d| <= |alpha| + |beta| |x| for a target / negative instance, |alpha| 4 sA gsum /
  // (sA^1.5 + eps) for a context instance; it leaves the batch's maximum in *bound (tagged with the step's sequence number, so
  // that nothing has to reset it).  A distinct row's sum is at most (its instance count) x that, the grouping kernels leave the
  // largest instance count in *cnt_max: k_seg_bwd takes 2^-shift off the scale BEFORE it rounds anything whenever
  // cnt_max x bound could pass 2^15.  Rigorous (a bound, not an estimate), and loose by design: a shift that was not
  // needed costs nothing while the values stay normal f16 numbers.
  int proactive = 0;
  const unsigned long long* bound = nullptr;   // GG_BOUND_SLOTS words, one 128-B line each (an item adds to slot blockIdx % 64: a
  const int32_t* cnt_max = nullptr;            //   thousand atomics on ONE word cost the score kernel 4 us of serialisation)
};
constexpr int GG_BOUND_SLOTS = 64, GG_BOUND_STRIDE = 16;     // (stride in 8-byte words)
#ifdef __HIPCC__
// the batch's largest per-instance element bound: max over the slots that carry this step's sequence number (every lane alike)
__device__ __forceinline__ float gg_bound_fold(const unsigned long long* bound, int32_t seq) {
  const unsigned long long w = bound[(threadIdx.x & (GG_BOUND_SLOTS - 1)) * GG_BOUND_STRIDE];
  float m = (unsigned)(w >> 32) == (unsigned)seq ? __uint_as_float((unsigned)w) : 0.f;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}
#endif

struct FwdArgs {
  const uint16_t* table;   // [n_rows + 1][Fp], last row all zero
  const int32_t* rows;     // [Rp] table row per batch row (already mapped, padding -> zero row)
  const uint16_t* Wh;      // [Dp][Fp]
  const float* bias;       // [D] or null
  const Scales* scales;
  float* H;                // [R][D] fp32 output (ip2)
  int R, D, Fp;
  int32_t zero_row;        // table row of zeros (for rows past R inside the last tile)
  int relu;
  // dropout
  float drop_ratio;        // 0 = off
  const uint8_t* mask;     // explicit mask [(C+Nn)*B][D] in reference row order, or null
  uint64_t drop_seed;
  int B, CN;               // to map an internal row (b*CN+ch) to the reference row (ch*B+b)
  const int32_t* n_dev = nullptr;   // dedup mode: device count of valid rows (overrides R; the grid covers R)
  int R_hint = 0;              // dedup mode: expected row count, sizes the tiles (0 = R)
  int32_t* seq_host = nullptr; // host-mapped word that receives `seq` (the step's index batch has been consumed), or null
  int32_t seq = 0;
  int abl = 0;                 // timing studies only (VV_ABLATE): selects an ablated instantiation of the phase-staggered kernel
  // Data-parallel overlap (api.hip: vv_apply_update): the parameter update of the PREVIOUS step may still be arriving --
  // F-chunk by F-chunk, each chunk all-reduced and applied on the communication stream -- while this kernel runs.
  // gate[c * W_GATE_STRIDE] holds the sequence number of the last update whose chunk c (K-tiles [gate_kt[c], gate_kt[c+1])
  // of W) is complete; a workgroup waits for it to reach gate_seq before it issues its first load of a W tile of chunk c.
  // null = no gating.
  const int32_t* gate = nullptr;
  int32_t gate_seq = 0;
  int gate_n = 0;              // chunks (1 .. W_CHUNKS_MAX)
  int gate_kt[5] = {0, 0, 0, 0, 0};   // first K-tile of each chunk (even; gate_kt[0] = 0, gate_kt[gate_n] = all K-tiles)
  int32_t* gate_err = nullptr; // host-mapped: set when a wait gave up (bounded: a lost update must not hang the device)
  // H as 16-bit floats (round 6, option "h16"): the forward GEMM stores ip2 as f16 -- [R][D] halves in the same buffer, scale 1, values past
  // f16's range saturated at +-65504 -- for the kernels that then read it as such (k_score_fwd / k_score_stream / k_seg_bwd: ScoreArgs::h16).
  // Only the segment-wise path sets it (api.hip); everything else keeps fp32 rows.
  int h16 = 0;
};
constexpr int W_CHUNKS_MAX = 4; // F-chunks of the overlapped update at most (chunk-major gradient buffer, api.hip: chunk_plan)
constexpr int W_GATE_STRIDE = 32;   // ints between two chunk flags: a 128-B line each (the waiting workgroups poll them)

// Segment-wise backward (de-duplicated batches): the score kernel keeps the gradient of an instance in factored form
// -- dY[r] = [x_u > 0] (alpha_r V[vec_r] - beta_r x_u), where V holds the item's normalised context mean Ah_b (row 2b:
// target / negative instances, alpha = c n^2/den, beta = c t/den) and the gradient of its context mean dA_b (row 2b+1:
// context instances, alpha = c_j, beta = 0) -- and writes one 16-byte record per instance at its grouped position.
// k_seg_bwd then produces, per distinct row u, dYu[u] = [x_u > 0] (sum_r alpha_r V[vec_r] - (sum_r beta_r) x_u) from
// the records of its instances: the per-instance 16-bit gradient rows ((C+Nn) B D values written and read back) are
// never materialised.
// The dropout mask of the path as a function (drop2 sits behind fc7 + ReLU, mednet_embedding_train.prototxt:220-230): element (reference
// row r = ch B + b, column n).  mode 1: counter hash -- quad n / 4 of row r owns counters row_ctr(r) + 2 (n / 4) + {0, 1}, each 32-bit hash
// gives two 16-bit uniforms, keep <=> uniform >= thr; mode 2: the caller's explicit mask[r D + n].  The forward GEMM's epilogue (dense
// execution) and the score / segment kernels (de-duplicated execution: the projection is shared, the mask is per instance) evaluate the SAME
// function, so the two executions drop the same elements.
struct DropSpec {
  int mode = 0;                    // 0 none, 1 counter hash, 2 explicit mask
  uint32_t thr = 0, s32 = 0;       // mode 1: threshold on the 16-bit uniform, the step's stream
  float scale = 1.f;               // 1 / (1 - ratio)
  const uint8_t* mask = nullptr;   // mode 2
  int B = 0, CN = 0, D = 0;
};
__device__ __forceinline__ uint32_t drop_mix32(uint32_t x) {     // (two multiply-xorshift rounds: C. Wellons' "lowbias32" constants)
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_row_ctr(int64_t ref_row, int D, uint32_t s32) {
  const uint64_t r2 = (uint64_t)ref_row * (uint64_t)(2 * ((D + 3) >> 2));
  return (uint32_t)r2 + (uint32_t)(r2 >> 32) * 0x9E3779B9u + s32;
}
// keep bits of columns n .. n + 3 (n a multiple of 4) of reference row ref_row; bit j = column n + j
__device__ __forceinline__ uint32_t drop_keep4(const DropSpec& d, int64_t ref_row, uint32_t row_ctr, int n) {
  if (d.mode == 2) {
    const uint32_t m = *(const uint32_t*)(d.mask + (uint64_t)ref_row * d.D + n);       // (D a multiple of 4 on these paths)
    return ((m & 0xffu) != 0) | (((m >> 8) & 0xffu) != 0) << 1 | (((m >> 16) & 0xffu) != 0) << 2 | ((m >> 24) != 0) << 3;
  }
  const uint32_t c0 = row_ctr + (uint32_t)(n >> 1);
  const uint32_t h0 = drop_mix32(c0), h1 = drop_mix32(c0 + 1u);
  return ((h0 & 0xffffu) >= d.thr) | ((h0 >> 16) >= d.thr) << 1 | ((h1 & 0xffffu) >= d.thr) << 2 | ((h1 >> 16) >= d.thr) << 3;
}

struct SegRec { float alpha, beta; int32_t vec; int32_t pad; };   // pad: the instance index b (C+Nn) + ch (sets the summation order)
constexpr int SEGB_BLOCKS = 1024;     // persistent grid of k_seg_bwd = rows of its bias partials

struct ScoreArgs {
  const float* H;          // [R][D] item-major rows (b*CN + ch)
  uint16_t* dYh;           // [Rp][Dp] scaled half gradient of ip1_nonorm
  float* dbp;              // [B][D] per-item column sums of dY (fp32, unscaled)
  float* loss_part;        // [B]
  float* viol_part;        // [B]
  float* s_true;           // [B] (replicated Nn times by the reference's SUM layer)
  float* s_bogus;          // [B][Nn]
  const float* coeff;      // [C-1] device
  int B, C, Nn, D, Dp;
  float margin; int norm;
  float grad_scale;        // loss_weight / global_count
  float drop_scale;        // 1/(1-ratio) or 1
  float sg;                // half-precision gradient scale
  // row-dedup mode (both NULL otherwise): H holds one row per UNIQUE table row, instance r reads
  // H[map[r]] and writes its gradient row to dYh[seg_start[map[r]] + ord[r]] (instances of one unique row contiguous)
  const float* item_w = nullptr;     // [B] loss-term weight of each item (MAX_MARGIN_LOSS 3rd bottom) or null
  const int32_t* map = nullptr;      // [R]
  const int32_t* seg_start = nullptr;  // [U + 1]   pos[r] = seg_start[map[r]] + ord[r]
  const int32_t* ord = nullptr;      // [R]
  GuardArgs guard;                   // f16 gradient-scale guard (kernels that write dYh)
  int items_rr = 0;                  // (lab, KernelOpts::score_rr) workgroup -> item round-robin instead of by XCD ranges
  int32_t* gate_host = nullptr;      // host-mapped word that receives gate_seq when the kernel starts ("the forward GEMM
  int32_t gate_seq = 0;              //   of this step has finished": releases the host to queue a later step's grouping)
  // segment-wise backward (launch_score_fwd): outputs instead of dYh / dbp
  float* V = nullptr;                // [2B][D]
  SegRec* rec = nullptr;             // [R]
  unsigned long long* bound_out = nullptr;   // GuardArgs::bound: (seq << 32) | bits of the largest per-instance element bound
  int32_t bound_seq = 0;
  DropSpec drop;                     // de-duplicated execution with dropout (k_score_fwd): H holds the SHARED pre-dropout rows, every
                                     // instance applies its own mask as it reads its row
  int h16 = 0;                       // H holds f16 rows (FwdArgs::h16): launch_score_fwd's kernels only
  int v16 = 0;                       // V leaves as f16 (k_score_stream's V16 form; needs h16): SegBwdArgs::v16
  int prefetch = 0;                  // k_score_fwd, f16 rows: first-round workgroups touch the second round's rows into the XCD's L2 (KernelOpts::score_pf)
  int lab_hack = 0;                  // (lab builds, VV_LAB_SCORE_HACK: timing studies of k_score_fwd's row loads -- WRONG results) 1: rows 1 KiB apart (half the footprint, the same requests), 2: only the first half of every row is loaded (half the bytes and requests)
  uint32_t* lab_ts = nullptr;        // (lab builds, VV_LAB_SCORE_TS=1: 16 words per item -- shader-clock stamps of k_score_fwd's phases, 100 MHz
                                     //  real time of its start and end, the compute unit it ran on; the product never sets or reads it)
};

// Row de-duplication of one batch (kernels_dedup.hip).  The sampler draws the negatives of every item
// from one shared buffer (video_sampled_shots_data_layer.cpp:836-875), so a batch repeats table rows;
// the projection is computed once per unique row and the gradient rows of its instances are summed
// before the weight-gradient GEMM.
struct DedupArgs {
  int lds_bytes = 0;             // dynamic LDS the kernels ask for and never touch (placement, kernels_dedup.hip)
  const int32_t* idx;            // [R] batch indices as sampled (-1 = empty slot)
  int32_t* rows;                 // [Rp] instance -> table row, written by k_dd_claim (what k_map_rows writes)
  unsigned long long* key;       // [table rows + scratch] epoch-tagged leader election
  unsigned long long* agg;       // [2][agg_stride] epoch-tagged block aggregates of the two scans
  int agg_stride;
  int32_t* slot_of;              // [R] slot of the leader instances
  int32_t* uniq_rows;            // [Rp] slot -> table row (zero_row past U)
  int32_t* map;                  // [R] instance -> slot
  int32_t* ord;                  // [R] arrival order of the instance inside its segment
  int32_t* cnt;                  // [Rp] instances per slot
  int32_t* seg_start;            // [Rp + 1]
  int32_t* pos;                  // [R] instance -> row of the grouped gradient buffer
  int32_t* info;                 // device {U, ...}
  int32_t* tickets;              // device [2]: arrival counters of the two single-pass scans
  int32_t* u_host;               // host-mapped copy of U (read one or more steps late by the launcher)
  int R, Rp; int32_t zero_row; int32_t row_limit; uint32_t epoch;
};

struct SegBwdArgs {
  const float* H;                // [U][D] projection of the distinct rows (x_u)
  const float* V;                // [2B][D]
  const SegRec* rec;             // [R] grouped by slot: records of slot u are rec[seg_start[u] .. seg_start[u+1])
  const int32_t* seg_start;      // [U + 1]
  const int32_t* info;           // {U}
  uint16_t* dYu;                 // [Rp][Dp] (zero rows up to the next multiple of BK)
  float* dbp;                    // [SEGB_BLOCKS][D] column sums of the unrounded rows, unscaled
  int Rp, D, Dp;
  float inv_sg;
  GuardArgs guard;
  DropSpec drop;                 // as ScoreArgs::drop: dx_u = [x_u > 0] (sum_i m_i alpha_i V_i - x_u scale sum_i m_i beta_i), m_i the instance's mask
  int h16 = 0;                   // H holds f16 rows (FwdArgs::h16)
  int v16 = 0;                   // V holds f16 rows (ScoreArgs::v16: the one-sweep score kernel wrote them so)
};

struct SegsumArgs {
  const uint16_t* dYh;           // [R][Dp] instance gradient rows grouped by slot
  const int32_t* seg_start;      // [U + 1]
  const int32_t* info;           // {U}
  uint16_t* dYu;                 // [Rp][Dp] per-slot sums (zero rows up to the next multiple of BK)
  int Rp, Dp;
  GuardArgs guard;
};

// The solver's update applied in the weight-gradient GEMM's epilogue (one split of K: the tile in the accumulators IS the gradient;
// api.hip: vv_update_hint).  What k_reduce_sgd does per 16 bytes of W -- unscale, the solver's rule (solver.cpp:502-531 / 599-655 / 714-781),
// blob.cpp:112-136, the new 16-bit copy, max |w| for the next scale -- done on the 256 x 256 tile while it is in registers: the 4 D F
// bytes of dW are neither written nor read back.  Bias, loss and the guard's report stay with k_reduce_sgd's special workgroups.
struct WgradUpd {
  float* W; float* hW; uint16_t* Wh;
  Scales* scales;
  float* wmax_blocks;                    // one slot per workgroup of the GEMM (this update's per-block max |w|)
  const float* wmax_prev; int wmax_prev_n; int recompute_scale; int prec;      // as FusedUpdArgs
  int D, F;                              // the matrix (tiles are padded to Dp x Fp)
  float rate, momentum, weight_decay, lr_mult_w, decay_mult_w, delta;
  int reg, solver_type;
  float sg; const GradGuard* gg; float ip_scale;          // dW = acc * ip_scale / (sg * gg->mul * sx), as ReduceArgs
};

struct WgradArgs {
  const uint16_t* dYh;     // [Rp][Dp]
  const uint16_t* table;   // [n_rows+1][Fp]
  const int32_t* rows;     // [Rp]
  float* slabs;            // [S][Dp][Fp] fp32 partial products
  int Rp, Dp, Fp;
  int S;                   // split-K factor
  int ksteps_per_split;    // BK-steps per split
  const int32_t* n_dev = nullptr;    // dedup mode: device count of valid K rows (overrides Rp / ksteps_per_split)
  int32_t zero_row = 0;              // the table's all-zero row (K-tiles padded past a split's end read it)
  int tm_begin = 0, tm_count = 0;    // restrict the launch to M tiles [tm_begin, tm_begin + tm_count) (0 = all): the
                                     // data-parallel overlap all-reduces one row block of dW while the next is computed
  int abl = 0;                       // as FwdArgs::abl
  int fuse_upd = 0;                  // 1: S == 1 and the epilogue applies `upd` instead of storing the tile to the slab
  WgradUpd upd;
  // Split-K partial products as f16 (round 6, option "slab16"; k_wgrad_gemm_ph only): the slab buffer holds [S][Dp][Fp] HALVES, every
  // (split, 256 x 256 tile) carries one power-of-two factor -- the tile's largest magnitude placed in [2^14, 2^15): no overflow whatever the
  // gradient scale, 11 significant bits per partial product -- whose inverse goes to slab_sc[split * tiles + tm * tilesN + tn]
  int slab16 = 0;
  float* slab_sc = nullptr;
  // the LEAN instantiations of k_wgrad_gemm_ph (round 6, option "wgrad_lean"): gathered rows addressed as table + 32-bit byte offset --
  // set by api.hip when every row of the table (scratch rows included) lies below 4 GiB and its index below 2^24
  int lean = 0;
};

struct ReduceArgs {
  const float* slabs; int S, Dp, Fp;
  int slab16 = 0; const float* slab_sc = nullptr;   // WgradArgs::slab16: f16 partial products and their per-(split, tile) factors
  const float* dbp; int B;
  int db_rows = 0;         // rows of dbp (0 = B, one per item; the segment-wise backward writes SEGB_BLOCKS partials)
  const Scales* scales;
  float sg;
  const GradGuard* gg = nullptr;   // f16: the step's final scale is sg * gg->mul
  const float* sg_dev = nullptr;   // per-layer operator: the scale chosen on the device (overrides sg)
  // the loss workgroup also reports the step's max |dY| (unscaled) and final shift to the host: ring of 16 entries
  const float* gmax_slots = nullptr; int gmax_n0 = 0, gmax_n1 = 0, gmax_stride = 0;
  unsigned long long* gmax_host = nullptr; int32_t seq = 0;
  int guard_last_round = 0;        // the step's final guard round (its flag must not be up)
  const unsigned long long* gbound = nullptr; const int32_t* gcnt = nullptr;   // proactive path: GuardArgs::bound / cnt_max (reported too)
  float* grads;            // [D*F + D]
  int D, F;
  float ip_scale;          // 1 + regularization/2 (inner_product_layer.cpp:80-90), normally 1
  // the loss reduction rides in one extra workgroup: loss = loss_scale * sum(loss_part), violations = sum(viol_part)
  const float* loss_part; const float* viol_part; float loss_scale; float* loss_out;
  int d_begin = 0, d_count = 0;      // rows of dW this launch reduces (d_count 0 = all D)
  int f_begin = 0, f_count = 0;      // columns of dW this launch reduces (f_count 0 = all F); multiples of 4
  int n_chunks = 0;                  // > 0: dW is stored CHUNK-MAJOR -- n_chunks column blocks, block c = all D rows of columns
  int chunk_c0[5] = {0, 0, 0, 0, 0}; //   [chunk_c0[c], chunk_c0[c+1]) at float offset D chunk_c0[c], row stride = the block's width
                                     //   (boundaries multiples of 4) -- so that every F-chunk is one contiguous all-reduce buffer
  int shard_rows = 0;                // > 0: the gradient buffer is SHARD-MAJOR (the sharded update, api.hip): D / shard_rows shards, shard s =
                                     //   rows [s shard_rows, (s + 1) shard_rows) of dW (row-major) followed by their shard_rows entries of db --
                                     //   one reduce-scatter message per rank, and k_sgd sees its shard as a small [dW | db] buffer of its own
  int parts = 3;                     // bit 0: the dW rows, bit 1: db and the loss scalars
  Scales* scale_sc = nullptr;        // non-null: one more workgroup performs the W -> half scale update left pending by the
  const float* scale_wmax = nullptr; //   previous step's k_sgd (its per-block max |w| slots,
  int scale_n = 0;                   //   this many of them)
  int scale_prec = 0;
};

struct SgdArgs {
  float* W; float* b; float* hW; float* hb;
  const float* grads;      // [D*F + D]
  uint16_t* Wh; Scales* scales;
  float* wmax_blocks;      // [SGD_BLOCKS] per-block max |w| of this update
  int D, F, Dp, Fp;
  float rate, momentum, weight_decay;
  float lr_mult_w, lr_mult_b, decay_mult_w, decay_mult_b;
  int reg;
  int solver_type;         // 0 SGD, 1 Nesterov, 2 AdaGrad
  float delta;             // AdaGrad stability constant
  // one F-chunk of the update (data-parallel overlap): columns [f_begin, f_begin + f_count) of every row, read from the
  // chunk-major gradient buffer (ReduceArgs::n_chunks; f_count may be 0: nothing but, possibly, the bias);
  // chunked = 0: the whole matrix from the row-major buffer
  int chunked = 0, f_begin = 0, f_count = 0;
  int blk_off = 0, n_blk = 0;   // slots of wmax_blocks this launch writes = its grid (the chunks of an update share SGD_BLOCKS)
  int do_bias = 1;         // this launch also updates b (the last chunk: db is all-reduced with it)
  int set_scale = 1;       // this launch publishes the scale of the new half copy (the first chunk)
  // Publication from inside the kernel (chunked launches of the overlapped update; null = none): what the NEXT forward
  // GEMM reads while this very stream is still working -- the half copy, its scale, the bias -- is stored write-through
  // at agent scope (sc1); every wave drains its stores, the workgroup meets at a barrier and adds itself to pub_count;
  // the workgroup whose add is the last sets *pub_flag = pub_seq (and clears the counter for the next chunk).  The
  // consumer polls the flag and runs an agent-scope acquire (FwdArgs::gate).  No separate launch, no L2 write-back.
  int32_t* pub_flag = nullptr; int32_t* pub_count = nullptr; int32_t pub_seq = 0;
  // data-parallel, direct peer transport: != 0 once an exchange in front of this launch gave up (a rank is missing) -- the gradient is
  // not the sum over the ranks; the launch changes nothing (and publishes nothing: the failure is fatal for the context, comm.hip)
  const uint32_t* skip_if = nullptr;
};

// Reduction and update in one launch (k_reduce_sgd; api.hip: the lazy reduction).  r: what k_reduce would have been given (its
// scale_* fields unused), g: what k_sgd would have been given.  Every workgroup that updates parameters derives the scale of
// the new half copy for itself from the previous update's per-block maxima (wmax_prev: what k_scale_update folds) when
// recompute_scale is set, else takes Scales::sw_next as it stands; this update's maxima go to g.wmax_blocks (another buffer).
struct FusedUpdArgs {
  ReduceArgs r; SgdArgs g;
  int no_params = 0;         // 1: only the special workgroups (bias, loss, the guard's report): the parameter matrix was updated in the
                             //    weight-gradient GEMM's epilogue (WgradUpd)
  const float* wmax_prev = nullptr; int wmax_prev_n = 0; int recompute_scale = 0; int prec = 0;
  int store_grads = 0;       // also write dW to the gradient buffer (0: it stays in the slabs, api.hip materialises it on request)
};
constexpr int WMAX_SLOTS = 2048;      // slots of one per-block-maxima buffer (k_sgd writes SGD_BLOCKS of them, k_reduce_sgd RED_DW_BLOCKS)

// Which kernels a context runs (vv_ctx::ko).  The launchers read the options of the context whose entry point is running on this
// thread (g_ko, set by every ABI entry point next to hipSetDevice); nothing here is process-global, so two contexts of one process
// do not inherit each other's choices.  Product options are set by vv_set_option (their environment variables are read ONCE per
// context, in vv_create); the fields marked (lab) select ablated / experimental kernels whose results may be WRONG: they are
// settable only in a build with -DVV_LAB (make lab -> lib/libvideovec_lab.so, what tools/lab/* use), and the ablated
// instantiations are compiled only there.
struct KernelOpts {
  int fwd_lead = 1;        // "fwd_lead" / VV_FWD_LEAD: the forward GEMM's sibling lead (kernels_gemm_ph.hip)
  int fwd_merge = 0;       // "fwd_merge" / VV_FWD_MERGE: the forward GEMM with two phases per barrier pair (k_fwd_gemm_ph, MRG)
  int wgrad_tr = 1;        // "wgrad_tr" / VV_WGRAD_TR: transposed LDS reads in the weight-gradient GEMM (0: the round-1 kernel)
  int score_stream = 0;    // "score_stream" / VV_SCORE_STREAM: the one-sweep score kernel for every shape
  int score_pf = 1;        // "score_pf" / VV_SCORE_PF=0: k_score_fwd's first-round workgroups prefetch the second round's rows into L2 (ScoreArgs::prefetch; 23.4 -> 22.8 us)
  int wgrad_lean = 1;      // "wgrad_lean" / VV_WGRAD_LEAN=0: k_wgrad_gemm_ph's lean instantiations (fewer instructions in the LOAD segments: scalar bases + 32-bit lane offsets, LDS addresses as immediates, a K loop without end-of-stream tests); only for tables below 4 GiB
  int gemm_variant = 5;    // (lab) VV_GEMM_VARIANT: 5 = the phase-staggered kernels; 0 = the round-1 kernels; 6 / 7 / 8 mixtures
  int ablate = 0;          // (lab) VV_ABLATE: ablated instantiations of the dense-size GEMMs (results wrong)
  int lab_fwd_abl = 0;     // (lab) VV_LAB_FWD_ABL: ablations of the 192-row forward kernel at the de-duplicated size (results wrong)
  int lab_wg_abl = 0;      // (lab) VV_LAB_WG_ABL
  int fwd_ring10 = 0;      // (lab) VV_FWD_RING10: the ten-slot forward kernel
  int ph_mq = 0;           // (lab) VV_PH_MQ: force the forward tile (2, 3, 4 = 128 / 192 / 256 rows, 31 = 176 rows)
  int score_reg = 1;       // (lab) VV_SCORE_REG=0: the LDS-resident score kernel
  int score_waves = 8;     // (lab) VV_SCORE_WAVES=4
  int lab_score_pipe = 0;  // (lab) VV_LAB_SCORE_PIPE=1: the persistent, pipelined score kernel (profiles/attic/score_fwd_pipelined.hip.txt)
  int score_rr = 0;        // (lab) VV_SCORE_RR=1: the item-major kernels deal their items round-robin over the XCDs again (kernels_elem.hip: item_of_block)
};
extern thread_local const KernelOpts* g_ko;
inline const KernelOpts& ko() { static const KernelOpts dflt; return g_ko ? *g_ko : dflt; }

// Kernel timing without extra queue packets: when the ABI layer has armed a pair of events (vv_profile_enable),
// the launch goes through hipExtLaunchKernelGGL, which stamps the events from the dispatch packet's own
// start / completion signal.  (A hipEventRecord pair around every kernel costs ~3 us of queue time per record on
// this runtime -- 50 us per step over the seven timed kernels -- and would distort the very step being measured.)
struct ProfPair { hipEvent_t start = nullptr, stop = nullptr; };
extern thread_local ProfPair g_prof;
#define VV_LAUNCH_EV(kern, grid, block, lds, s, use_start, use_stop, ...)                                   \
  do {                                                                                                     \
    hipEvent_t e0_ = (use_start) ? vv::g_prof.start : nullptr, e1_ = (use_stop) ? vv::g_prof.stop : nullptr; \
    if (e0_ || e1_) {                                                                                      \
      hipExtLaunchKernelGGL(kern, grid, block, lds, s, e0_, e1_, 0, __VA_ARGS__);                           \
      if (use_start) vv::g_prof.start = nullptr;                                                           \
      if (use_stop) vv::g_prof.stop = nullptr;                                                             \
    } else hipLaunchKernelGGL(kern, grid, block, lds, s, __VA_ARGS__);                                      \
  } while (0)
#define VV_LAUNCH(kern, grid, block, lds, s, ...) VV_LAUNCH_EV(kern, grid, block, lds, s, true, true, __VA_ARGS__)
#define VV_LAUNCH_FIRST(kern, grid, block, lds, s, ...) VV_LAUNCH_EV(kern, grid, block, lds, s, true, false, __VA_ARGS__)
#define VV_LAUNCH_LAST(kern, grid, block, lds, s, ...) VV_LAUNCH_EV(kern, grid, block, lds, s, false, true, __VA_ARGS__)

// kernel launchers (defined in the .hip files); prec: 0 = f16, 1 = bf16
void launch_fwd_gemm(int prec, const FwdArgs& a, hipStream_t s);
bool fwd_gemm_can_gate(const FwdArgs& a);                     // the forward kernel has a gated form for these arguments (FwdArgs::gate)
long fwd_gemm_plan(int R, int R_hint, int D, int* mq_out);   // workgroups of the default forward GEMM that get a tile
void launch_wgrad_gemm(int prec, const WgradArgs& a, hipStream_t s);
bool wgrad_can_fuse_update();
void launch_score_loss(int prec, const ScoreArgs& a, hipStream_t s);
void launch_dedup(const DedupArgs& a, hipStream_t s);
void launch_dedup_groups(const DedupArgs& a, hipStream_t s);
void launch_dedup_pos(const DedupArgs& a, hipStream_t s);   // pos[] for the debug accessors only
void launch_segsum(int prec, const SegsumArgs& a, hipStream_t s);
bool score_fwd_dropout_supported(int D, int C, int Nn);
bool score_fwd_supported(const ScoreArgs& a);               // shapes the segment-wise pair is built for
void launch_score_fwd(const ScoreArgs& a, hipStream_t s);   // forward + factored backward records (dedup mode)
void launch_seg_bwd(int prec, const SegBwdArgs& a, hipStream_t s);
// (h16: src holds f16 rows -- FwdArgs::h16; map == nullptr: row r is its own source)
void launch_gather_rows_dropout(const float* src, const int32_t* map, int R, int D, const DropSpec& dr, float* dst, hipStream_t s, int h16 = 0);
void launch_gather_rows_f32(const float* src, const int32_t* map, int R, int D, float* dst, hipStream_t s, int h16 = 0);
void launch_gather_rows_u16(const uint16_t* src, const int32_t* pos, int R, int Dp, uint16_t* dst, hipStream_t s);
void launch_reduce(const ReduceArgs& a, hipStream_t s);
void launch_sgd(int prec, const SgdArgs& a, hipStream_t s);
void launch_publish(int32_t* flag, int32_t seq, hipStream_t s);
void launch_delay(int us, hipStream_t s);                             // test hook: occupies s for `us` microseconds       // flag <- seq (agent scope), behind everything queued on s
void launch_scale_update(int prec, Scales* sc, const float* wmax_blocks, int n_blocks, hipStream_t s);
int launch_reduce_sgd(const FusedUpdArgs& a, hipStream_t s);   // -> the number of per-block maxima it writes
void launch_table_convert(int prec, const float* src, uint16_t* dst, int64_t n_rows, int F, int Fp,
                          float sx, hipStream_t s);
void launch_table_synth(int prec, uint16_t* dst, uint64_t seed, int64_t n_rows, int F, int Fp,
                        float sx, hipStream_t s);
void launch_table_read(int prec, const uint16_t* table, const int32_t* rows, int64_t n, int F,
                       int Fp, float inv_sx, float* out, hipStream_t s);
void launch_w_convert(int prec, const float* W, uint16_t* Wh, int D, int F, int Dp, int Fp,
                      Scales* sc, hipStream_t s);
void launch_map_rows(const int32_t* idx, int32_t* rows, int R, int Rp, int32_t zero_row, int32_t row_limit,
                     hipStream_t s);
void launch_final_loss(const float* loss_part, const float* viol_part, int B, float scale,
                       float* out2, hipStream_t s);
void launch_dyh_to_float(int prec, const uint16_t* dYh, int R, int D, int Dp, float inv_sg,
                         float* out, hipStream_t s);
void launch_row_normalize(float* x, int n, int D, hipStream_t s);
void launch_absmax(const float* x, int64_t n, unsigned* out_bits, hipStream_t s);
void launch_mean_rows(int prec, uint16_t* table, const int32_t* rows, int64_t n, int k, const float* coeff,
                      int64_t first_row, int Fp, hipStream_t s);
void launch_gram(const float* x, int n, int dim, float alpha, float* out, hipStream_t s);
void launch_patch_rows(uint16_t* table, const int32_t* desc, int64_t n_patch, int64_t first_row, int F,
                       int Fp, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// 16-bit operand types
// ---------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));

struct F16 {
  static constexpr int id = 0;
  static __device__ __forceinline__ f32x4 mfma(i16x8 a, i16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint16_t from_float(float x) {
    x = fminf(fmaxf(x, -65504.f), 65504.f);   // saturate instead of producing inf
    return __builtin_bit_cast(uint16_t, (_Float16)x);
  }
  static __device__ __forceinline__ float to_float(uint16_t v) {
    return (float)__builtin_bit_cast(_Float16, v);
  }
};

struct BF16 {
  static constexpr int id = 1;
  static __device__ __forceinline__ f32x4 mfma(i16x8 a, i16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint16_t from_float(float x) {
    return __builtin_bit_cast(uint16_t, (__bf16)x);
  }
  static __device__ __forceinline__ float to_float(uint16_t v) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
  }
};

#ifdef __HIPCC__
// ---- f16 gradient-scale guard, device side (GradGuard above).  Every thread of the block calls these.
// max over the block, valid in every thread; smem: >= 16 floats of LDS not otherwise in use around the call
__device__ __forceinline__ float gg_block_max(float v, float* smem) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = v;
  __syncthreads();
  float m = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, smem[w]);
  return m;
}
// Start of a producer launch.  false: a repeat whose previous round raised no flag -- nothing to do, the caller returns
// (all threads alike).  sg_mul: the power of two to put on top of the host's scale in this round.
__device__ __forceinline__ bool gg_begin(const GuardArgs& g, float* smem, float& sg_mul) {
  sg_mul = 1.f;
  if (!g.gg || g.round == 0) return true;
  const int prev = g.round > 1 ? g.gg->shift[g.round - 1] : 0;
  int shift = prev;
  const bool active = g.gg->flag[g.round - 1] == g.seq;
  if (active) {
    float m = 0.f;
    const float* sl = g.slots + (size_t)(g.round - 1) * 2 * g.nslot;
    for (int p = 0; p < 2; ++p)
      for (int i = threadIdx.x; i < g.n_of[p]; i += blockDim.x) m = fmaxf(m, sl[(size_t)p * g.nslot + i]);
    m = gg_block_max(m, smem);
    int e = 0;
    (void)frexpf(m, &e);                       // m = f 2^e, f in [0.5, 1): afterwards the largest value lies in [2^11, 2^12)
    shift = prev + (e - 12 > 1 ? e - 12 : 1);
    __syncthreads();                           // smem may be reused by the caller
  }
  if (g.last && blockIdx.x == 0 && threadIdx.x == 0) {
    if (active) g.gg->repeat_seq = g.seq;
    g.gg->shift[g.round] = shift;
    if (g.final_round) { g.gg->mul = ldexpf(1.f, -shift); g.gg->shift[3] = shift; }     // [3]: the step's final shift
  }
  sg_mul = ldexpf(1.f, -shift);
  return active;
}
// End of a producer launch: block_max = this block's max |g| in the round's scaled units, before rounding (thread 0's
// value is used).
__device__ __forceinline__ void gg_end(const GuardArgs& g, float block_max) {
  if (!g.gg || threadIdx.x != 0) return;
  g.slots[((size_t)g.round * 2 + g.producer) * g.nslot + blockIdx.x] = block_max;
  if (block_max > GG_LIMIT) __hip_atomic_store(&g.gg->flag[g.round], g.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// splitmix64 finaliser, identical to videovector_amd/synth.py:mix64
__host__ __device__ inline uint64_t mix64(uint64_t seed, uint64_t x) {
  uint64_t z = seed * 0x9E3779B97F4A7C15ull + x;
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

}  // namespace vv
