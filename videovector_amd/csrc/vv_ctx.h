// vv_ctx.h -- the context object behind include/videovec.h's vv_ctx, shared by api.hip (the fused training step)
// and ops.hip (the per-layer operators).  Internal.
#pragma once
#include <cstdarg>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/videovec.h"
#include "vv_internal.h"
#include "vv_comm.h"

extern thread_local char vv_g_err[512];
int vv_fail(int code, const char* fmt, ...);
#define fail vv_fail
#define HIPCHK(x)                                                                              \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess)                                                                      \
      return vv_fail(VV_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, \
                     __LINE__);                                                                \
  } while (0)

// A device allocation that lives for one API call: released on every return path, the early error returns of HIPCHK
// included.
template <typename T>
struct DevTmp {
  T* p = nullptr;
  DevTmp() = default;
  DevTmp(const DevTmp&) = delete;
  DevTmp& operator=(const DevTmp&) = delete;
  ~DevTmp() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t count) { return hipMalloc((void**)&p, count * sizeof(T)); }
  operator T*() const { return p; }
};

struct ProfEntry { std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };

struct vv_ctx {
  int device = 0, prec = 0;
  hipStream_t stream = nullptr, own_stream = nullptr;

  // feature table
  uint16_t* table = nullptr; int64_t n_rows = 0; int F = 0, Fp = 0; float sx = 1.f;
  int64_t patch_cap = 0;            // scratch rows after the zero row (quirk Q1 composites)
  int32_t* patch_desc = nullptr; int64_t patch_desc_cap = 0;
  // parameters
  int D = 0, Dp = 0;
  float *W = nullptr, *b = nullptr, *hW = nullptr, *hb = nullptr;
  uint16_t* Wh = nullptr; vv::Scales* scales = nullptr; float* wmax_blocks = nullptr;
  int n_cu = 256;                           // compute units of the device
  bool scale_pending = false;               // k_sgd ran, its scale update has not (it rides in the next k_reduce)
  // wmax_blocks holds two buffers of WMAX_SLOTS per-block maxima: an update writes the one the previous update did not
  int wmax_cur = 0, wmax_n = 0;             // the buffer the latest update wrote, and how many slots of it
  bool wmax_seed_live = false;              // Scales::wmax_bits still carries vv_params_set's seed (the next scale update consumes it)
  // Lazy reduction: vv_forward_backward leaves dW in the split-K slabs (and db / the loss in their partials) when nothing
  // but an update is likely to want them; vv_apply_update then reduces and updates in ONE launch (k_reduce_sgd), and
  // anything that reads the gradient or the loss first runs the plain k_reduce (reduce_now in api.hip).
  bool red_lazy = false; vv::ReduceArgs red_args;
  bool grads_exposed = false;               // vv_grads_device handed the buffer out: its holder may read it any time, so the reduction is eager from then on
  bool grads_stale = false;                 // the fused update ran without writing dW to the gradient buffer: it is still in the slabs
  float* grads = nullptr;           // [D*F + D] (own buffer, or the bound external one)
  float* grads_own = nullptr;
  // per-batch buffers
  int B = 0, C = 0, Nn = 0, R = 0, Rp = 0;
  int32_t *idx_dev = nullptr, *rows = nullptr;
  float* H = nullptr; uint16_t* dYh = nullptr; float* dbp = nullptr;
  float *loss_part = nullptr, *viol_part = nullptr, *s_true = nullptr, *s_bogus = nullptr;
  float* coeff = nullptr; std::vector<float> coeff_host;
  float* item_w = nullptr;          // [B] loss-term weights of the current batch
  uint8_t* mask = nullptr; size_t mask_bytes = 0;
  float* slabs = nullptr; size_t slab_bytes = 0; int S = 1, kps = 0;
  bool drop_dedup = true;                   // option "drop_dedup" (VV_DROP_DEDUP): dropout rides the de-duplicated path where the kernels carry the masks
  vv::DropSpec last_drop;                   // the last step's dropout on the de-duplicated path (mode 0: none): the blob accessors re-apply it
  float* loss2 = nullptr;           // {loss, violations}
  float sg = 1.f; float last_loss_weight = 1.f;
  uint64_t iter = 0;
  bool have_fwd = false;
  // row de-duplication (kernels_dedup.hip)
  int dedup = 1;                    // 1 = on whenever dropout is off (VV_DEDUP / vv_set_dedup)
  bool last_dedup = false;          // what the last forward/backward pass used
  // f16 gradient-scale guard (vv_internal.h: GradGuard)
  vv::GradGuard* gg = nullptr; float* gg_slots = nullptr; int gg_nslot = 0;
  unsigned long long* gg_bound = nullptr;      // GuardArgs::bound: the score kernel's (seq, largest per-instance element bound)
  unsigned long long* gmax_host = nullptr;     // pinned + mapped: 16 entries of {seq | bits(max |dY|) << 32, final shift}
  unsigned long long* gmax_host_dev = nullptr;
  int sg_adj = 0;                   // powers of two on top of the count-based default scale (follows the reported maxima)
  int32_t gg_seq0 = 0;              // first step since the guard's state was reset (no reports older than that)
  int64_t gg_repeats = 0;           // steps whose gradients had to be produced again at a smaller scale
  float gg_last_mul = 1.f;          // (debug accessors) final scale multiplier of the last step, read back on demand
  unsigned long long* dd_key = nullptr; int64_t dd_key_cap = 0;
  unsigned long long* dd_agg = nullptr; int dd_agg_stride = 0;
  int32_t *dd_slot_of = nullptr, *dd_uniq = nullptr, *dd_map = nullptr, *dd_ord = nullptr, *dd_cnt = nullptr,
          *dd_seg = nullptr, *dd_pos = nullptr, *dd_info = nullptr;
  uint16_t* dYu = nullptr;
  // segment-wise backward (vv_internal.h: SegRec)
  float* segV = nullptr; vv::SegRec* seg_rec = nullptr; float* seg_dbp = nullptr;
  bool seg_bwd = true;             // env VV_SEG_BWD=0: the per-instance gradient rows + k_segsum instead
  bool last_seg_bwd = false;
  bool slab16 = true;              // option "slab16" (VV_SLAB16=0: fp32): the weight gradient's split-K partial products as f16 x a power of two per (split, tile)
  float* slab_sc = nullptr;        // ... their inverse factors, [8][tiles]
  bool v16 = true;                 // option "v16" (VV_V16=0: fp32): the per-item vectors of the one-sweep score kernel (D = 1024) as f16, with h16
  bool h16 = true;                 // option "h16" (VV_H16=0: fp32 rows): ip2 as f16 between the forward GEMM and the segment-wise pair (FwdArgs::h16)
  bool last_h16 = false;           // ... and whether the last forward pass stored it that way (the accessors read H accordingly)
  vv::ScoreArgs last_score;            // to rebuild the per-instance gradient rows for vv_blobs_get(ip1_diff)
  int32_t* U_host = nullptr;        // pinned + mapped: k_dd_leaders stores U here every step, the launcher reads it late
  int32_t* U_host_dev = nullptr;    // device alias of U_host
  uint32_t dd_epoch = 0;
  // The grouping kernels of step k+1 run on a stream of their own while step k is still executing (they need only the
  // indices): four sets of their output arrays rotate (dd_* above point at the set of the step being issued), the forward
  // GEMM's sequence stamp says when the step that read a set is over, an event per set when its grouping is done.
  static constexpr int kDdSets = 4;
  struct DdSet {
    int32_t *rows = nullptr, *slot_of = nullptr, *uniq = nullptr, *map = nullptr, *ord = nullptr, *cnt = nullptr, *seg = nullptr,
            *info = nullptr;
    hipEvent_t done = nullptr; int32_t used_seq = 0;      // sequence number of the step that last read the set
  } dd_set[kDdSets];
  int32_t* dd_info_all = nullptr;
  int32_t* dd_rows = nullptr;      // instance -> table row of the current set (k_dd_claim's output)
  hipStream_t dd_stream = nullptr;
  uint64_t dd_step = 0;
  double dd_spin_us = 60.0;        // how long the host watches the grouping's event before queueing a stream wait (VV_DEDUP_SPIN_US)
  bool dd_async = true;            // env VV_DEDUP_ASYNC=0: the grouping kernels in the step's own stream
  int32_t step_seq = 0;
  // staging of index batches taken from a sampler's prefetch ring (vv_forward_backward_ring)
  static constexpr int kStage = 8;
  int32_t* stage_host[kStage] = {};         // pinned, mapped into the device's address space
  int32_t* stage_dev[kStage] = {};          // the device's alias of the same memory
  int32_t stage_seq[kStage] = {};           // sequence number of the step that last read the slot
  size_t stage_bytes = 0; int32_t stage_next = 0;
  int32_t* seq_host = nullptr;              // pinned + mapped, two words.  [0]: the forward GEMM stores the step's sequence
  int32_t* seq_host_dev = nullptr;          //   number here when it starts, i.e. "the kernels that read this step's index
                                            //   batch have finished"; [1]: the score kernel does when IT starts, i.e. "this
                                            //   step's forward GEMM has finished" (the gate of the asynchronous grouping)
  // data-parallel gradient exchange (comm.hip)
  vv::Comm* comm = nullptr;
  bool comm_overlap = false;        // the update runs F-chunk by F-chunk on the communication stream (all-reduce, SGD, publish) and
                                    // the NEXT step's forward GEMM waits per chunk inside the kernel (FwdArgs::gate)
  bool grads_pending = false;       // a backward pass has produced gradients that have not been all-reduced yet
  bool grads_chunked = false;       // the gradient buffer of the last backward pass is laid out chunk-major (ReduceArgs::n_chunks)
  // per-context switches (vv_set_option; environment read once, in vv_create).  Nothing of this is process-global.
  vv::KernelOpts ko;                // which kernels the launchers pick (vv_internal.h)
  bool fuse_update = true;          // "fuse_update" / VV_FUSE_UPDATE=0: reduce at the end of every backward pass, update apart
  bool comm_gate = true;            // "comm_gate" / VV_COMM_GATE=0: the overlapped update never gates the forward GEMM (the stream joins)
  // vv_update_hint: the next vv_forward_backward* is followed by vv_apply_update with these solver parameters and nothing reads the gradient
  // in between -- with one split of K the weight-gradient GEMM then applies the update itself (WgradUpd)
  bool upd_hint = false;
  bool wgrad_update = true;                 // option "wgrad_update" (VV_WGRAD_UPDATE): a hinted step with one split of K applies its update in the weight-gradient
                                            // GEMM's epilogue.  ON since the epilogue walks its row groups in an order that depends on the tile (k_wgrad_gemm_ph, UPD):
                                            // before that, two processes in sixteen ran it at half speed (182-196 us instead of 100-125), decided by where the buffers
                                            // happened to lie in physical memory; with it forty processes in forty at 95-107 us: 0.211-0.224 ms per iteration at the
                                            // shipped shape against 0.229-0.236 with the update as its own launch (profiles/r05_shipped_update.txt).
  vv_step_cfg upd_cfg;
  bool upd_in_wgrad = false;                // this step's parameter matrix is already updated; vv_apply_update finishes the step
  int upd_wgrad_blocks = 0;                 // per-block max |w| slots that launch wrote
  bool grads_lost = false;                  // the last step's dW was consumed in registers (vv_grads_* refuse until the next backward pass)
  uint32_t* lab_score_ts = nullptr;         // (lab builds) device buffer of k_score_fwd's phase stamps, api.hip
  bool comm_inline = true;                  // option "comm_inline" (VV_COMM_INLINE): the sharded update queued on the compute stream itself (no second stream, no gate)
  int comm_test_delay_us = 0;       // "comm_test_delay_us" / VV_COMM_TEST_DELAY_US: TEST HOOK -- the communication stream held this long per chunk
  // (lab) -- settable in a -DVV_LAB build only
  bool guard_proactive = true;      // VV_GUARD_PROACTIVE=0: the repeat form of the gradient-scale guard on the segment-wise path too
  bool fuse_keep_grads = false;     // VV_FUSE_KEEP_GRADS=1: the fused update also writes dW out
  bool comm_skip_ar1 = false;       // VV_COMM_SKIP_AR1=1: no collective call at world 1 (diagnosis)
  int dd_gate_word = 0;             // VV_DEDUP_GATE=1: the grouping is released by the score kernel's stamp
  int dd_lds_kb = -1;               // VV_DEDUP_LDS_KB: dynamic LDS the grouping kernels ask for (placement)
  double trace_host_ms = -1.0;      // VV_TRACE_HOST: report ABI calls that keep the host longer than this
  bool trace_waits = false;         // VV_TRACE_WAITS: where the host waits
  double wait_ms[5] = {0, 0, 0, 0, 0}; long wait_calls = 0;
  // SHARDED update (vv_comm_schedule 2): fp32 reduce-scatter of the shard-major [dW rows | db entries] buffer, the solver's rule on this
  // rank's D / world rows of W / history / bias, all-gather of the 16-bit copy + the bias + the per-block maxima that every rank's next
  // forward pass reads.  The fp32 master W and the history are complete only on their owner until vv_params_get gathers them.
  bool comm_sharded = false;
  bool grads_sharded = false;       // the gradient buffer of the last backward pass is laid out shard-major (ReduceArgs::shard_rows)
  bool params_partial = false;      // W / hW / hb rows of the other ranks' shards are stale (a sharded update ran since the last gather)
  bool upd_inflight = false;        // an overlapped update is on the communication stream and the compute stream has not joined it
  bool upd_unjoined = false;        // ... the gated forward GEMM has consumed it, the compute stream has still not waited for its end
  int32_t upd_seq = 0;              // sequence number of the last overlapped update (what w_gate[c] reaches when chunk c is done)
  int32_t* w_gate = nullptr;        // device [W_CHUNKS_MAX * W_GATE_STRIDE]: one flag per 128-B line
  int n_chunks = 3;                 // F-chunks of the overlapped update (env VV_COMM_CHUNKS, 1 .. 4)
  int chunk_kt[5] = {0, 0, 0, 0, 0};    // first K-tile of each chunk for the current Fp (chunk_plan)
  int32_t* pub_count = nullptr;     // device: arrival counter of the publishing SGD kernels (behind the flags)
  int32_t* pub_count0 = nullptr;    // ... of the first chunk's kernel when that one runs on the compute stream (overlap_first_inline)
  bool overlap_first_inline = false; // VV_COMM_FIRST_INLINE=1: the overlapped update's first F-chunk (exchange + SGD) in the compute stream, the rest on the communication
                                    // stream (round 5: built, measured, NOT the default -- one rank over real RCCL 0.2490-0.2501 ms against 0.2401-0.2430 with every
                                    // chunk on the communication stream: chunk 0 loses its head start behind its own reduction launch, profiles/r05_overlap_cost.txt)
  hipEvent_t ev_chunk0 = nullptr; bool chunk0_event = false;     // the first F-chunk's reduction is done (recorded by fb_impl)
  int32_t* gate_err = nullptr; int32_t* gate_err_dev = nullptr;     // pinned + mapped: a gated forward gave up waiting
  hipEvent_t ev_chunk = nullptr;
  hipEvent_t ev_idx = nullptr;      // orders the grouping stream behind caller-produced device indices (idx_on_device = 1)
  // profiling
  bool prof = false;
  int prof_every = 1;               // record every prof_every-th forward/backward + update (vv_profile_enable's argument)
  uint64_t prof_calls = 0;
  std::map<std::string, ProfEntry> prof_map;
  std::string prof_only;            // vv_profile_select: ",name,name," -- only these kernels are timed (empty = all)
  std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;    // events are reused across profiling sessions
};


// ops.hip keeps per-context scratch for the per-layer INNER_PRODUCT operators; vv_destroy releases it
void vv_ops_release(vv_ctx* c);
// api.hip: orders the compute stream behind an overlapped parameter update still on the communication stream
int vv_comm_join(vv_ctx* c);
