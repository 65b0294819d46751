// kernels_dedup.hip -- row de-duplication of one batch (gfx950 only).
//
// Why: VideoSampledShotsDataLayer draws the Nn negatives of every batch item from ONE shared ring
// buffer of at most max_buffer_size frames (video_sampled_shots_data_layer.cpp:836-875), so the
// (C+Nn)*B rows of a batch name far fewer distinct frames (cfg 2: 56 320 rows, ~20 700 distinct).
// The reference copies every repeat and multiplies it again.  Here the fc projection runs once per
// distinct table row, every instance reads the shared output row, and the gradient rows of the
// instances are summed per distinct row before the weight-gradient GEMM:
//     dW = sum_r dY[r]^T X[row(r)] = sum_u (sum_{r -> u} dY[r])^T X[u].
// Same results (the projection of equal inputs is equal; the gradient sum is reassociated).
//
//   k_dd_claim    : index -> table row (empty slots -> the zero row), then leader election, atomicMax of (epoch, ~r) per table row -> the smallest instance
//                   index of every distinct row wins; epoch tags make a reset pass unnecessary.
//   k_dd_leaders  : slot = rank of the leader among leaders (single-pass scan: every workgroup takes a ticket,
//                   publishes its count as an epoch-tagged word and sums the words of the earlier tickets).
//                   Slots are ordered by first appearance, so the order is deterministic.
//   k_dd_map      : instance -> slot, per-slot instance count, arrival order inside the slot.
//   k_dd_segstart : exclusive scan of the counts (same single-pass scan).
//   k_dd_pos      : instance -> row of the grouped gradient buffer (debug accessor only; the score kernel
//                   computes seg_start[map[r]] + ord[r] inline).
//   k_segsum      : per slot, sum of its instances' 16-bit gradient rows.  f16: accumulated in f64,
//                   which is EXACT for up to 2^13 f16 addends, so the arrival order (atomics) cannot
//                   change the result.  bf16: f64 as well (order-independent unless the addends span
//                   more than ~40 binades).
#include "vv_internal.h"

namespace vv {

constexpr int DD_BLOCK = 1024;

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void k_dd_claim(DedupArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r == 0) { a.tickets[0] = 0; a.info[1] = 0; }   // k_dd_leaders' ticket counter, k_dd_segstart's max count (kernel boundaries order the resets)
  if (r >= a.Rp) return;
  int row = a.zero_row;
  if (r < a.R) { const int i = a.idx[r]; if (i >= 0 && i < a.row_limit) row = i; }   // as k_map_rows
  a.rows[r] = row;                     // what k_map_rows does on the dense path
  a.uniq_rows[r] = a.zero_row; a.cnt[r] = 0;
  if (r < a.R)
    atomicMax(&a.key[row], ((unsigned long long)a.epoch << 32) | (0xFFFFFFFFu - (unsigned)r));
}

// exclusive block scan of one int per thread (DD_BLOCK threads); returns the prefix, *total = block sum
__device__ __forceinline__ int block_excl_scan(int v, int* total, int* sm /* >= 17 ints */) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
  if (lane == 63) sm[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    int w = lane < DD_BLOCK / 64 ? sm[lane] : 0;
    int winc = w;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) { const int t = __shfl_up(winc, o, 64); if (lane >= o) winc += t; }
    if (lane < DD_BLOCK / 64) sm[lane] = winc - w;
    if (lane == DD_BLOCK / 64 - 1) sm[16] = winc;
  }
  __syncthreads();
  const int res = sm[wave] + inc - v;
  *total = sm[16];
  __syncthreads();
  return res;
}

// Logical workgroup index = arrival order (a ticket), not blockIdx: a workgroup only ever waits for tickets
// smaller than its own, whose owners have already started, so the scan cannot deadlock whatever order the
// hardware dispatches workgroups in.
__device__ __forceinline__ int take_ticket(int32_t* counter, int* sm) {
  if (threadIdx.x == 0) sm[19] = atomicAdd(counter, 1);
  __syncthreads();
  const int t = sm[19];
  __syncthreads();
  return t;
}

// sum of the aggregates of logical workgroups [0, bid): spin on each word until it carries this epoch
__device__ __forceinline__ int lookback_sum(const unsigned long long* agg, unsigned epoch, int bid, int* sm) {
  int s = 0;
  for (int j = threadIdx.x; j < bid; j += DD_BLOCK) {
    unsigned long long w;
    do { w = ld_agent(agg + j); } while ((unsigned)(w >> 32) != epoch);
    s += (int)(unsigned)w;
  }
  int tot;
  block_excl_scan(s, &tot, sm);
  return tot;
}

__global__ __launch_bounds__(DD_BLOCK) void k_dd_leaders(DedupArgs a) {
  __shared__ int sm[24];
  const int bid = take_ticket(a.tickets, sm);
  const int r = bid * DD_BLOCK + threadIdx.x;
  int row = 0, flag = 0;
  if (r < a.R) {
    row = a.rows[r];
    flag = (unsigned)a.key[row] == 0xFFFFFFFFu - (unsigned)r;
  }
  int bt;
  const int lp = block_excl_scan(flag, &bt, sm);
  if (threadIdx.x == 0) st_agent(a.agg + bid, ((unsigned long long)a.epoch << 32) | (unsigned)bt);
  const int off = lookback_sum(a.agg, a.epoch, bid, sm);
  if (flag) {
    a.slot_of[r] = off + lp;
    a.uniq_rows[off + lp] = row;
  }
  if (bid == (int)gridDim.x - 1 && threadIdx.x == 0) {
    a.info[0] = off + bt;
    __hip_atomic_store(a.u_host, off + bt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void k_dd_map(DedupArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r == 0) a.tickets[1] = 0;        // k_dd_segstart's ticket counter
  if (r >= a.R) return;
  const unsigned lead = 0xFFFFFFFFu - (unsigned)a.key[a.rows[r]];
  const int s = a.slot_of[lead];
  a.map[r] = s;
  a.ord[r] = atomicAdd(&a.cnt[s], 1);
}

__global__ __launch_bounds__(DD_BLOCK) void k_dd_segstart(DedupArgs a) {
  __shared__ int sm[24];
  const int U = a.info[0];
  const int bid = take_ticket(a.tickets + 1, sm);
  const int u = bid * DD_BLOCK + threadIdx.x;
  const int v = u < U ? a.cnt[u] : 0;
  {                                                  // info[1] = the largest instance count of a distinct row (GuardArgs::cnt_max)
    int m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(a.info + 1, m);
  }
  int bt;
  const int lp = block_excl_scan(v, &bt, sm);
  unsigned long long* agg = a.agg + a.agg_stride;
  if (threadIdx.x == 0) st_agent(agg + bid, ((unsigned long long)a.epoch << 32) | (unsigned)bt);
  if (bid * DD_BLOCK > U) return;                    // nothing to write (block-uniform)
  const int off = lookback_sum(agg, a.epoch, bid, sm);
  if (u <= U) a.seg_start[u] = off + lp;             // u == U: the total (= R)
}

__global__ __launch_bounds__(256) void k_dd_pos(DedupArgs a) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r < a.R) a.pos[r] = a.seg_start[a.map[r]] + a.ord[r];
}

// (The four passes as ONE launch with grid-wide barriers -- an arrival counter in device memory, every cross-workgroup
// value passed through agent-scope atomics because the eight XCDs' L2s are not coherent inside a kernel -- was built and
// measured: 26.9 us against 24.7 us for the four launches; with release / acquire fences at agent scope instead of
// atomics 120 us.  A barrier across 55 workgroups costs about what a dependent launch does on this part.)

// DedupArgs.lds_bytes: the grouping kernels of a later step run on the context's second stream BESIDE the forward GEMM
// (api.hip, the async block).  That GEMM is one persistent 512-thread workgroup per CU holding 128 of the CU's 160 KiB of
// LDS, on 216 of the 256 CUs at cfg 2; a grouping workgroup placed on one of those CUs slows that CU's GEMM workgroup and
// the whole GEMM waits for it (0.085-0.088 ms against 0.080 alone).  Asking for 36 KiB of dynamic LDS they never touch
// makes the grouping workgroups too big for the 32 KiB a GEMM workgroup leaves free: the dispatcher can place them only on
// the CUs the GEMM does not use.  Forward GEMM 0.083 ms, step 3-5 us shorter (profiles/r02_step_ablations.txt, 4.).  The
// caller asks for it only when the GEMM leaves CUs idle: beside a GEMM of more than one round of workgroups (cfg 5) a
// grouping workgroup that cannot share a CU takes a whole CU away from the GEMM for its lifetime (forward 0.47 -> 0.50 ms).

// part 1: what the forward GEMM needs (distinct rows and their count)
void launch_dedup(const DedupArgs& a, hipStream_t s) {
  const int g256 = (a.Rp + 255) / 256;
  VV_LAUNCH_FIRST(k_dd_claim, dim3(g256), dim3(256), a.lds_bytes, s, a);
  hipLaunchKernelGGL(k_dd_leaders, dim3((a.R + DD_BLOCK - 1) / DD_BLOCK), dim3(DD_BLOCK), a.lds_bytes, s, a);
}
// part 2: what the score kernel needs (instance -> slot / grouped gradient row); independent of the
// forward GEMM, so the caller may run it on a second stream beside it
void launch_dedup_groups(const DedupArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_dd_map, dim3((a.R + 255) / 256), dim3(256), a.lds_bytes, s, a);
  VV_LAUNCH_LAST(k_dd_segstart, dim3(a.R / DD_BLOCK + 1), dim3(DD_BLOCK), a.lds_bytes, s, a);
}
// instance -> grouped gradient row as an array; the score kernel computes the same value inline, only the
// debug accessor (vv_blobs_get ip1_diff) needs it materialised
void launch_dedup_pos(const DedupArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_dd_pos, dim3((a.R + 255) / 256), dim3(256), 0, s, a);
}

// One wave per slot; a lane owns 8 consecutive columns (16-B loads) of every 512-column chunk.
template <typename T>
__global__ __launch_bounds__(256) void k_segsum(SegsumArgs a) {
  // f16: the sum of a row repeated thousands of times, with gradients far above their usual scaled size, could pass
  // 65504: the f16 gradient-scale guard (vv_internal.h: GradGuard) sees the sums before they are rounded.  A repeat finds
  // the instance rows already rewritten at the reduced scale by the score kernel's repeat and simply sums them again.
  __shared__ float ggs[16];
  float sgm;
  if (!gg_begin(a.guard, ggs, sgm)) return;
  const int U = a.info[0];
  const int Uk = (U + BK - 1) / BK * BK;             // the wgrad K loop reads whole BK-row steps
  const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const bool live = u < Uk && u < a.Rp;
  double gmx = 0.0;
  int b = 0, e = 0;
  if (u < U) { b = a.seg_start[u]; e = a.seg_start[u + 1]; }
  for (int c0 = lane * 8; live && c0 < a.Dp; c0 += 512) {
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int i = b; i < e; ++i) {
      const uint4 v = *(const uint4*)(a.dYh + (int64_t)i * a.Dp + c0);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += (double)T::to_float((uint16_t)(w[j] & 0xFFFFu));
        acc[2 * j + 1] += (double)T::to_float((uint16_t)(w[j] >> 16));
      }
    }
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o[j] = T::from_float((float)acc[2 * j]) | ((uint32_t)T::from_float((float)acc[2 * j + 1]) << 16);
#pragma unroll
    for (int j = 0; j < 8; ++j) gmx = fmax(gmx, fabs(acc[j]));
    *(uint4*)(a.dYu + (int64_t)u * a.Dp + c0) = make_uint4(o[0], o[1], o[2], o[3]);
  }
  if (a.guard.gg) gg_end(a.guard, gg_block_max((float)fmin(gmx, 3.0e38), ggs));
}

void launch_segsum(int prec, const SegsumArgs& a, hipStream_t s) {
  const dim3 grid((a.Rp + 3) / 4), block(256);
  if (prec == 0) VV_LAUNCH(k_segsum<F16>, grid, block, 0, s, a);
  else VV_LAUNCH(k_segsum<BF16>, grid, block, 0, s, a);
}

// debug / parity accessors: expand per-slot rows back to per-instance rows (H16: the source rows are f16, FwdArgs::h16)
template <bool H16>
__global__ __launch_bounds__(256) void k_gather_rows_f32(const float* src, const int32_t* map, int R, int D, float* dst) {
  const int r = blockIdx.x;
  const int64_t so = (int64_t)(map ? map[r] : r) * D;
  for (int d = threadIdx.x; d < D; d += 256) dst[(int64_t)r * D + d] = H16 ? (float)((const _Float16*)src)[so + d] : src[so + d];
}
// ... with the instance's dropout mask applied (de-duplicated execution with dropout: src holds the shared pre-dropout rows); instance
// r = b CN + ch, its mask is that of the reference's row ch B + b (DropSpec)
template <bool H16>
__global__ __launch_bounds__(256) void k_gather_rows_dropout(const float* src, const int32_t* map, int R, int D, DropSpec dr, float* dst) {
  const int r = blockIdx.x;
  const int bb = r / dr.CN, ch = r - bb * dr.CN;
  const int64_t rr = (int64_t)ch * dr.B + bb;
  const uint32_t rc = drop_row_ctr(rr, D, dr.s32);
  const int64_t so = (int64_t)(map ? map[r] : r) * D;
  for (int d = threadIdx.x * 4; d < D; d += 1024) {
    const uint32_t kp = drop_keep4(dr, rr, rc, d);
    for (int j = 0; j < 4 && d + j < D; ++j) {
      const float x = H16 ? (float)((const _Float16*)src)[so + d + j] : src[so + d + j];
      dst[(int64_t)r * D + d + j] = ((kp >> j) & 1u) ? x * dr.scale : 0.f;
    }
  }
}
void launch_gather_rows_dropout(const float* src, const int32_t* map, int R, int D, const DropSpec& dr, float* dst, hipStream_t s, int h16) {
  if (R <= 0) return;
  if (h16) hipLaunchKernelGGL(k_gather_rows_dropout<true>, dim3(R), dim3(256), 0, s, src, map, R, D, dr, dst);
  else hipLaunchKernelGGL(k_gather_rows_dropout<false>, dim3(R), dim3(256), 0, s, src, map, R, D, dr, dst);
}
void launch_gather_rows_f32(const float* src, const int32_t* map, int R, int D, float* dst, hipStream_t s, int h16) {
  if (R <= 0) return;
  if (h16) hipLaunchKernelGGL(k_gather_rows_f32<true>, dim3(R), dim3(256), 0, s, src, map, R, D, dst);
  else hipLaunchKernelGGL(k_gather_rows_f32<false>, dim3(R), dim3(256), 0, s, src, map, R, D, dst);
}
__global__ __launch_bounds__(256) void k_gather_rows_u16(const uint16_t* src, const int32_t* pos, int R, int Dp, uint16_t* dst) {
  const int r = blockIdx.x;
  const uint16_t* sp = src + (int64_t)pos[r] * Dp;
  for (int d = threadIdx.x; d < Dp; d += 256) dst[(int64_t)r * Dp + d] = sp[d];
}
void launch_gather_rows_u16(const uint16_t* src, const int32_t* pos, int R, int Dp, uint16_t* dst, hipStream_t s) {
  if (R > 0) hipLaunchKernelGGL(k_gather_rows_u16, dim3(R), dim3(256), 0, s, src, pos, R, Dp, dst);
}

}  // namespace vv
