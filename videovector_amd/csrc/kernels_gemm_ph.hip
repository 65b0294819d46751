// kernels_gemm_ph.hip -- phase-staggered MFMA kernels of the videovec training step (gfx950 only).
//
// Same products as kernels_gemm.hip (k_fwd_gemm: ip2 = ReLU(X W^T + b), X gathered through the triplet index;
// k_wgrad_gemm: dW = dY^T X, split-K slabs) on a different schedule.  The first kernels load a whole K-step, wait for it
// (vmcnt(0)) and run 64 MFMAs per wave between two barriers: the two waves of a SIMD do the same thing at the same
// time, so the matrix pipe idles while both read LDS and the LDS idles while both multiply (rocprofv3: matrix pipe busy
// 33-47 %, half of every wave's life parked at the K-step's wait + barrier).  Here:
//
//   * operand tiles are cut into HALF-TILES of 16 KiB -- A_lo, B_lo, B_hi, A_hi per 64-deep K-tile -- that travel
//     HBM/L2 -> LDS by LDS-DMA through a ring of 8 slots, issued 6 half-tiles (1.5 K-tiles) ahead of their use and
//     waited for with COUNTED s_waitcnt vmcnt(8): four half-tiles (64 KiB per CU) stay in flight across every barrier;
//   * a K-tile is 4 PHASES of 16*MQ/4 MFMAs per wave (one quadrant of the wave's output x K = 64); every wave owns
//     rows from BOTH A halves and columns from BOTH B halves, so a phase needs only the half-tiles staged first:
//         phase 0: A_lo x B_lo    phase 1: A_lo x B_hi    phase 2: A_hi x B_hi    phase 3: A_hi x B_lo (B_lo kept in registers)
//     and a slot is free for the stream again as soon as its quadrant has been read (A_lo, B_lo after phase 0, ...);
//   * a phase is a LOAD segment (this phase's ds_reads + 2 LDS-DMA instructions) and an MFMA segment with a workgroup
//     barrier after each, and waves 4-7 run ONE SEGMENT BEHIND waves 0-3 (they pass one extra barrier before the
//     loop): on every SIMD one wave multiplies while its partner reads LDS and issues loads.
//
// Hazards (LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader
// has passed; a slot is restaged at least two segments after its last ds_read was issued):
//   RAW  the wait at the end of LOAD(P) covers everything LOAD(P+1) reads; between it and LOAD(P+1) of either group
//        lies a barrier that every wave passes after its own wait.
//   WAR  the half-tile issued in LOAD(t,p) replaces the one consumed in phase (t-1,p+2) / (t,p-2): at least two
//        phases (four barriers) earlier.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "vv_internal.h"

namespace vv {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ void ph_glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}
__device__ __forceinline__ int ph_xcd_remap(int bid, int nblk) {
  const int x = bid & 7, q = nblk >> 3, rem = nblk & 7;
  return x * q + (x < rem ? x : rem) + (bid >> 3);
}

constexpr int PH_SLOT = 16384;                 // one half-tile: 128 rows x 64 halves
constexpr int PH_LDS_BYTES = 8 * PH_SLOT;      // ring of 8 slots = 128 KiB
#define PH_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// ------------------------------------------------------------------------------- forward ------
// Half-tile LDS image: [128 rows][64 halves] = 128-B rows of 8 16-B chunks, chunk' = chunk ^ (row & 7)
// (conflict-free ds_read_b128 fragment reads, as in k_fwd_gemm).  One LDS-DMA wave-instruction = 8 rows.
// MQ = 16-row MFMA tiles per wave and A half: the tile is (64*MQ) x 256 (MQ 4: 256 rows, 3: 192, 2: 128).
template <typename T, bool DROP, bool VEC, int MQ>
__global__ __launch_bounds__(GEMM_THREADS) void k_fwd_gemm_ph(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HROWS = 32 * MQ;               // live rows of an A half-tile
  constexpr int BMT = 2 * HROWS;               // rows of the output tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const int tilesN = Dp / BN;
  const int R = a.n_dev ? *a.n_dev : a.R;
  const int nact = a.n_dev ? ((R + BMT - 1) / BMT) * tilesN : (int)gridDim.x;
  if (a.seq_host && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.seq_host, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((int)blockIdx.x >= nact) return;
  const int L = ph_xcd_remap(blockIdx.x, nact);
  const int m0 = (L / tilesN) * BMT, n0 = (L % tilesN) * BN;
  const int Fp = a.Fp;

  // staging sources: LDS-DMA instruction i (0, 1) of this wave fills rows (i*8 + wave)*8 .. +7 of a half-tile
  const uint16_t* srcA[2][2];
  const uint16_t* srcB[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 8 + (lane >> 3), lc = (lane & 7) ^ (row & 7);
      const int grow = m0 + hf * HROWS + row;
      const int trow = (row < HROWS && grow < R) ? a.rows[grow] : a.zero_row;
      srcA[hf][i] = a.table + (int64_t)trow * Fp + lc * 8;
      srcB[hf][i] = a.Wh + (int64_t)(n0 + hf * 128 + row) * Fp + lc * 8;
    }

  f32x4 acc[2][MQ][2][2];
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mi = 0; mi < MQ; ++mi)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mh][mi][nh][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = Fp / BK;                      // K-tiles (Fp is a multiple of 256: nk % 4 == 0)
  const int H = 4 * nk;                        // half-tiles of this workgroup's stream
  // half-tile h = 4*kt + q, q: 0 A_lo, 1 B_lo, 2 B_hi, 3 A_hi; slot = h & 7
  auto issue = [&](int kt, int q, int slot) {
    const uint16_t* const* src = q == 0 ? srcA[0] : q == 1 ? srcB[0] : q == 2 ? srcB[1] : srcA[1];
    unsigned char* dst = smem + slot * PH_SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) ph_glds16(src[i] + kt * BK, dst + (i * 8 + wave) * 1024);
  };

  // prologue: half-tiles 0 .. 5
  issue(0, 0, 0); issue(0, 1, 1); issue(0, 2, 2); issue(0, 3, 3); issue(1, 0, 4); issue(1, 1, 5);
  PH_WAIT(8);                                  // half-tiles 0, 1 (A_lo, B_lo of K-tile 0) have landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one segment behind
  if (wm == 1) __builtin_amdgcn_s_setprio(1);  // the younger half loses VALU arbitration otherwise

  const int frow = lane & 15, fq = lane >> 4;
  const int a_off = (wm * 16 * MQ + frow) * 128;          // + mi*16*128, chunk by kk
  const int b_off = (wn * 32 + frow) * 128;               // + ni*16*128
  const int sw = frow & 7;
  i16x8 af[MQ][2], b0[2][2], b1[2][2];

  // one phase: LOAD segment (reads + stream + wait), barrier, MFMA segment, barrier
#define PH_LOAD_A(slot)                                                                              \
  _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
    af[mi][kk] = *(const i16x8*)(smem + (slot) * PH_SLOT + a_off + mi * 2048 + (((kk * 4 + fq) ^ sw) << 4));
#define PH_LOAD_B(dst, slot)                                                                         \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)  \
    dst[ni][kk] = *(const i16x8*)(smem + (slot) * PH_SLOT + b_off + ni * 2048 + (((kk * 4 + fq) ^ sw) << 4));
#define PH_MFMA(mh, nh, bfr)                                                                         \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                 \
      acc[mh][mi][nh][ni] = T::mfma(bfr[ni][kk], af[mi][kk], acc[mh][mi][nh][ni]);                   \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);
  // stream step of phase (t, p): half-tile 4t + p + 6, then the counted wait for what the next phase reads
#define PH_STREAM(tpar, t, p, wait)                                                                  \
  {                                                                                                  \
    const int h = 4 * (t) + (p) + 6;                                                                 \
    if (h < H) { issue(h >> 2, ((p) + 2) & 3, 4 * (((tpar) + (((p) + 6) >> 2)) & 1) + (((p) + 2) & 3)); if (wait) PH_WAIT(8); } \
    else if (wait) PH_WAIT(0);                                                                       \
  }

  for (int t = 0; t < nk; t += 2) {
    // ---- K-tile t (even): slots 0..3
    PH_LOAD_A(0) PH_LOAD_B(b0, 1) PH_STREAM(0, t, 0, true) PH_MFMA(0, 0, b0)
    PH_LOAD_B(b1, 2) PH_STREAM(0, t, 1, true) PH_MFMA(0, 1, b1)
    PH_LOAD_A(3) PH_STREAM(0, t, 2, false) PH_MFMA(1, 1, b1)
    PH_STREAM(0, t, 3, true) PH_MFMA(1, 0, b0)
    // ---- K-tile t + 1 (odd): slots 4..7
    PH_LOAD_A(4) PH_LOAD_B(b0, 5) PH_STREAM(1, t + 1, 0, true) PH_MFMA(0, 0, b0)
    PH_LOAD_B(b1, 6) PH_STREAM(1, t + 1, 1, true) PH_MFMA(0, 1, b1)
    PH_LOAD_A(7) PH_STREAM(1, t + 1, 2, false) PH_MFMA(1, 1, b1)
    PH_STREAM(1, t + 1, 3, true) PH_MFMA(1, 0, b0)
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // waves 0-3 catch the extra barrier of waves 4-7
#undef PH_LOAD_A
#undef PH_LOAD_B
#undef PH_MFMA
#undef PH_STREAM

  // Epilogue: descale, bias, ReLU, dropout.  The MFMA was issued with the operands swapped (D' = W_tile X_tile^T):
  // the lane's column is the batch row m and its 4 registers are 4 consecutive outputs n -> one 16-B store.
  const float descale = 1.0f / (a.scales->sx * a.scales->sw_cur);
  const float dscale = DROP ? 1.0f / (1.0f - a.drop_ratio) : 1.0f;
  const float lo = a.relu ? 0.f : -INFINITY;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mi = 0; mi < MQ; ++mi) {
      const int m = m0 + mh * HROWS + wm * 16 * MQ + mi * 16 + frow;
      if (m >= R) continue;
      int64_t ref_row = 0;
      if (DROP) {
        const int bb = m / a.CN, ch = m - bb * a.CN;
        ref_row = (int64_t)ch * a.B + bb;
      }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = n0 + nh * 128 + wn * 32 + ni * 16 + fq * 4;
          if (n >= a.D) continue;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float bj = (n + j < a.D) ? a.bias[n + j] : 0.f;
            v[j] = fmaxf(acc[mh][mi][nh][ni][j] * descale + bj, lo);
            if (DROP) {
              const uint64_t e = (uint64_t)(ref_row * a.D + n + j);
              bool keep;
              if (a.mask) keep = (n + j < a.D) && a.mask[e] != 0;
              else keep = (float)(mix64(a.drop_seed, e) >> 40) * (1.0f / 16777216.0f) >= a.drop_ratio;
              v[j] = keep ? v[j] * dscale : 0.f;
            }
          }
          float* dst = a.H + (int64_t)m * a.D + n;
          if (VEC) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          else
            for (int j = 0; j < 4; ++j) if (n + j < a.D) dst[j] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------- launchers ----
template <typename T, bool DROP, bool VEC, int MQ>
static void launch_fwd_ph_q(const FwdArgs& a, hipStream_t s) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES), true);
  (void)once;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const dim3 grid(((a.R + 64 * MQ - 1) / (64 * MQ)) * (Dp / BN)), block(GEMM_THREADS);
  VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ>), grid, block, PH_LDS_BYTES, s, a);
}

static int g_ph_mq = 0;                   // VV_PH_MQ: force the tile height (2, 3, 4); 0 = automatic
void set_ph_mq(int v) { g_ph_mq = v; }

template <typename T, bool DROP, bool VEC>
static void launch_fwd_ph_t(const FwdArgs& a, hipStream_t s) {
  const int Dp = (int)round_up(a.D, D_ALIGN);
  // tile height 64*MQ with the least (rounds of 256 workgroups) x (cost of one K-tile of that height); the cost of a
  // K-tile is not proportional to MQ: the LDS-DMA stream and the barriers do not shrink with the tile
  const int Rh = a.n_dev && a.R_hint > 0 ? (int)std::min<long>(a.R, a.R_hint + a.R_hint / 32 + 64) : a.R;
  static const int kCost[5] = {0, 0, 70, 85, 100};
  int best = 4; long best_cost = -1;
  for (int mq = 4; mq >= 2; --mq) {
    const long tiles = ((Rh + 64 * mq - 1) / (64 * mq)) * (long)(Dp / BN);
    const long cost = ((tiles + 255) / 256) * kCost[mq];
    if (best_cost < 0 || cost < best_cost) { best = mq; best_cost = cost; }
  }
  if (g_ph_mq >= 2 && g_ph_mq <= 4) best = g_ph_mq;
  if (best == 4) launch_fwd_ph_q<T, DROP, VEC, 4>(a, s);
  else if (best == 3) launch_fwd_ph_q<T, DROP, VEC, 3>(a, s);
  else launch_fwd_ph_q<T, DROP, VEC, 2>(a, s);
}

template <typename T>
static void launch_fwd_ph_p(const FwdArgs& a, hipStream_t s) {
  const bool drop = a.drop_ratio > 0.f, vec = a.D % 4 == 0;
  if (drop) { if (vec) launch_fwd_ph_t<T, true, true>(a, s); else launch_fwd_ph_t<T, true, false>(a, s); }
  else { if (vec) launch_fwd_ph_t<T, false, true>(a, s); else launch_fwd_ph_t<T, false, false>(a, s); }
}

void launch_fwd_gemm_ph(int prec, const FwdArgs& a, hipStream_t s) {
  if (prec == 0) launch_fwd_ph_p<F16>(a, s); else launch_fwd_ph_p<BF16>(a, s);
}

}  // namespace vv
