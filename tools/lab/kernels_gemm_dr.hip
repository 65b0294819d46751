// kernels_gemm_dr.hip -- forward projection ip2 = ReLU(X W^T + b) with the L2-resident operand loaded STRAIGHT INTO REGISTERS
// (gfx950 only).  Math: /root/reference/src/caffe/layers/inner_product_layer.cpp:61-74 (+ relu_layer.cpp:13-22).
//
// k_fwd_gemm_ph (kernels_gemm_ph.hip) moves BOTH operands HBM/L2 -> LDS by LDS-DMA and back out by ds_read: the weight
// matrix, 4 MB and L2-hot, is 57 % of the bytes that travel through the workgroup's fill path, its fragment reads are 40 %
// of the LDS reads, and its slots need the workgroup barriers (8 per K-tile) that the schedule is built around
// (profiles/r03_step_ablations.txt 2e: the stream alone 81 us, everything but the stream 61 us).  Here:
//
//   * the 16-bit copy of W is kept in MFMA OPERAND ORDER (Wq, `wq_index` below): the 16 x 32 fragment of one MFMA is 1 KiB
//     of consecutive bytes, lane l's 16 bytes at l * 16, and the four fragments a wave needs per K-tile are 4 KiB in a row.
//     A wave loads them with global_load_dwordx4 -- fully coalesced, no LDS write, no LDS read, no barrier -- P K-tiles
//     before it multiplies them (P x 16 registers);
//   * every wave owns 32 of the tile's 256 columns and ALL of its rows (the waves are 1 x 8, not 2 x 4): no W fragment is
//     loaded twice, and the gathered rows -- the operand that has to be shared -- are the only thing in LDS: an image of
//     16 MT rows x 128 B per K-tile (24 KiB at 192 rows: no padding rows), P + 1 of them in a ring, filled by LDS-DMA P
//     K-tiles ahead (P = 4: 96 KiB of gathered lines in flight per CU against 32 KiB in k_fwd_gemm_ph);
//   * ONE workgroup barrier per K-tile: in the second half of K-tile t every wave waits for its own pieces of A(t+1)
//     (counted vmcnt) and for its fragment reads of A(t), meets the others, and goes on with reads of A(t+1);
//   * a wave's instruction stream is uniform -- per 16-row MFMA tile one ds_read_b128 and two MFMAs -- with a ring of four
//     fragments read ahead, so there are no load / multiply segments to stagger.
//
// Accumulation order per output element: K-blocks of 32 in ascending order, one MFMA each, W as the first operand -- the
// order of k_fwd_gemm_ph: outputs are bit for bit the same.
//
// vmcnt bookkeeping (all vector-memory operations of a wave complete in issue order): K-tile step t issues, at the end of
// its first half, B(t+P, kk 0) [2 loads] and N0 pieces of A(t+P), at the end of its second half B(t+P, kk 1) [2] and the other
// N1 pieces.  The wait in the second half of step t needs everything issued in step t + 1 - P: younger than that are P - 2 whole
// steps and the first half of step t = (P - 2)(4 + NPW) + 2 + N0 operations.
// Slots: A(t+P) goes where A(t-1) was (NS = P + 1); its pieces are issued after the barrier of step t - 1, in front of which
// every wave has waited for its last fragment reads of A(t-1).
#include <algorithm>
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include "../../videovector_amd/csrc/vv_internal.h"

namespace vv {

#define DR_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// element (n, k) of W in the operand-ordered copy: [n / 32][k / 64][kk = k / 32 % 2][ni = n / 16 % 2][lane = (k / 8 % 4) 16 + n % 16][k % 8]
__host__ __device__ inline int64_t wq_index(int n, int k, int nk) {
  return ((((int64_t)(n >> 5) * nk + (k >> 6)) * 4 + ((k >> 5) & 1) * 2 + ((n >> 4) & 1)) * 512) + ((((k >> 3) & 3) * 16 + (n & 15)) * 8) + (k & 7);
}

__device__ __forceinline__ int dr_xcd_remap(int bid, int nblk) {
  const int x = bid & 7, q = nblk >> 3, rem = nblk & 7;
  return x * q + (x < rem ? x : rem) + (bid >> 3);
}
// one LDS-DMA piece: 64 lanes x 16 B from sbase + off[lane] to LDS [lds_addr, +1 KiB)
template <int CP = 0>
__device__ __forceinline__ void dr_glds16(unsigned off, const void* sbase, unsigned lds_addr) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
  if (CP == 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m0v), "v"(off), "s"(sbase) : "m0", "memory");
  else if (CP == 1) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" :: "s"(m0v), "v"(off), "s"(sbase) : "m0", "memory");
  else if (CP == 2) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 nt" :: "s"(m0v), "v"(off), "s"(sbase) : "m0", "memory");
  else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 sc0 sc1" :: "s"(m0v), "v"(off), "s"(sbase) : "m0", "memory");
}
template <int IMM>
__device__ __forceinline__ void dr_gload16(i16x8& dst, unsigned voff, const void* sbase) {
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
#define DR_LDSRD(addr) (*(const __attribute__((address_space(3))) i16x8*)(uintptr_t)(addr))

// ABL (lab, results wrong): 1 no A stream in the loop, 2 no MFMA, 4 no fragment reads, 8 every gathered row the zero row, 16 no W loads,
// 64 no stores
// TD (0 = off, 4, 8): TOUCH PREFETCH.  Once per four K-tiles every lane asks for ONE dword of one 128-B line of the gathered rows
// -- the four lines (512 B in a row) that K-tiles [t + TD, t + TD + 4) will stream -- into a scratch word of LDS: one wave
// instruction names 64 lines.  The lines arrive in L2 / the Infinity Cache long before the LDS-DMA asks for them, and HBM sees 512-B
// runs instead of single lines.  SIB: the tile has a sibling (the other column half of the same rows, same XCD): each touches
// half of the rows, every line is fetched from HBM on behalf of ONE of them.
// OPT bit 0: every wave waits for its fragment reads (lgkmcnt(0)) in front of the barrier; (lab) bit 1: sibling column tiles on different XCDs;
// bits 2-3: cache policy of the gathered rows' LDS-DMA (1 sc1, 2 nt, 3 sc0 sc1)
template <typename T, int MT, int P, int ABL = 0, int TD = 0, bool SIB = true, int OPT = 0>
__global__ __launch_bounds__(GEMM_THREADS) void k_fwd_gemm_dr(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BMT = 16 * MT, SLOTB = BMT * 128, NPW = 2 * MT / 8, NS = P + 1, AD = 4;
  constexpr int N1 = NPW / 2, N0 = NPW - N1, GRP = 4 + NPW;
  constexpr int TROWS = SIB ? BMT / 2 : BMT, NTI = TD ? (TROWS * 4 + 511) / 512 : 0;
  static_assert(TD == 0 || (P == 4 && (TD == 4 || TD == 8)), "touch prefetch");
  static_assert((2 * MT) % 8 == 0 && MT > AD && (P == 2 || P == 4), "tile");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const int tilesN = Dp / BN;
  const int R = a.n_dev ? *a.n_dev : a.R;
  const int nact = a.n_dev ? ((R + BMT - 1) / BMT) * tilesN : (int)gridDim.x;
  if (a.seq_host && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.seq_host, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((int)blockIdx.x >= nact) return;
  int L = dr_xcd_remap(blockIdx.x, nact);
  if (OPT & 2) {                               // (lab) the column halves of a row tile on DIFFERENT XCDs: XCD x works on column tile x % tilesN only
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int rt = x / tilesN + (8 / tilesN) * j;
    if (rt * BMT >= R) return;
    L = rt * tilesN + x % tilesN;
  }
  const int m0 = (L / tilesN) * BMT, n0 = (L % tilesN) * BN;
  const int Fp = a.Fp, nk = Fp / BK;

  // gathered operand: piece (i, wave) = rows (i*8 + wave)*8 .. +7 of the image, lane -> (row, 16-B chunk ^ row & 7)
  unsigned aoff[NPW];
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int row = (i * 8 + wave) * 8 + (lane >> 3), grow = m0 + row;
    const int trow = (grow < R && !(ABL & 8)) ? a.rows[grow] : a.zero_row;
    aoff[i] = (unsigned)((int64_t)trow * Fp * 2) + (((lane & 7) ^ (lane >> 3)) << 4);
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)DR_LDS_PTR(smem));
  // touch prefetch: lane g of the workgroup -> (row, line) pairs g, g + 512 of this sibling's rows
  unsigned toff[NTI > 0 ? NTI : 1];
  if constexpr (TD > 0) {
    const int sib = SIB ? (L % tilesN) & 1 : 0;
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      const int idx = j * 512 + tid, row = idx >> 2, grow = m0 + sib * TROWS + row;
      const int trow = (row < TROWS && grow < R && !(ABL & 8)) ? a.rows[grow] : a.zero_row;
      toff[j] = (unsigned)((int64_t)trow * Fp * 2) + ((idx & 3) << 7);
    }
  }
  const unsigned lds_scratch = lds0 + NS * SLOTB;      // 256 B per wave behind the ring
  auto touch = [&](int kt4) {                  // the lines of K-tiles [kt4, kt4 + 4) (clamped: the count of operations is what the waits assume)
    const int k = kt4 + 4 <= nk ? kt4 : nk - 4;
#pragma unroll
    for (int j = 0; j < NTI; ++j) {
      const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_scratch + wave * 256);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, %2" :: "s"(m0v), "v"(toff[j]), "s"(a.table + (int64_t)k * BK) : "m0", "memory");
    }
  };
  // W: this wave's run of 4-KiB blocks
  const uint16_t* wbase = a.Wh + ((int64_t)(n0 / 32 + wave) * nk) * 2048;
  const unsigned voff = lane * 16;

  f32x4 acc[MT][2];
#pragma unroll
  for (int mi = 0; mi < MT; ++mi) { acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  i16x8 bq[P][2][2];                           // [K-tile % P][kk][ni]
  i16x8 af[AD];

  // (lab, OPT bit 4, stream timing only -- the LDS image is wrong): a piece = 4 rows x 256 B instead of 8 rows x 128 B; even K-tiles fetch
  // the lower half of the rows for TWO K-tiles, odd ones the upper half: the same bytes, in 256-byte runs per row visit
  // RUNG = row groups = bytes per row visit / 128 (OPT bits 7-8 on top of bit 4: 2 -> 256 B, 4 -> 512 B, 8 -> 1 KiB)
  constexpr int RUNG = !(OPT & 16) ? 1 : (((OPT >> 7) & 3) == 0 ? 2 : ((OPT >> 7) & 3) == 1 ? 4 : 8);
  unsigned aoff2[RUNG][NPW];
  if (OPT & 16) {
#pragma unroll
    for (int hf = 0; hf < RUNG; ++hf)
#pragma unroll
      for (int i = 0; i < NPW; ++i) {
        constexpr int LPR = 8 * RUNG;            // lanes per row (16 B each)
        const int row = hf * (BMT / RUNG) + (i * 8 + wave) * (64 / LPR) + lane / LPR, grow = m0 + row;
        const int trow = (grow < R && !(ABL & 8)) ? a.rows[grow] : a.zero_row;
        aoff2[hf][i] = (unsigned)((int64_t)trow * Fp * 2) + ((lane % LPR) << 4);
      }
  }
  // (lab, OPT bits 5-6: K ROTATION) odd column tiles run their K loop 1 / 2 / 4 tiles ahead (tile (t + rot) mod nk at step t): a row tile's
  // two siblings never ask for the same gathered lines at the same time -- the odd one is every line's first toucher, the even one finds it
  // in L2 rot steps later -- with no extra LDS.  The accumulation order of the odd tiles is rotated with it (not bit-identical to k_fwd_gemm_ph).
  // (lab, OPT bit 11: K SPREAD) every ROW tile starts its K loop at another tile (19 * row tile mod nk; the column siblings together): at any
  // moment the workgroups of the chip ask for different 128-byte offsets of their 8-KiB rows, i.e. for different L2 / memory channels, instead
  // of marching through the offsets in step.
  const int rot = (OPT & 2048) ? (int)(((unsigned)(L / tilesN) * 19u) % (unsigned)nk)
                : ((OPT >> 5) & 3) == 0 ? 0 : (((L % tilesN) & 1) ? (1 << (((OPT >> 5) & 3) - 1)) : 0);
  auto rotk = [&](int tt) { const int k = tt + rot; return k >= nk ? k - nk : k; };
  // (lab, OPT bits 9-10, STREAM TIMING ONLY: the LDS image is wrong) BALANCED lead: every workgroup asks for the half of its rows that matches its
  // column parity 1 / 2 / 4 K-tiles early -- each sibling is the first toucher of half of the lines and finds the other half in L2
  const int blead = ((OPT >> 9) & 3) == 0 ? 0 : (1 << (((OPT >> 9) & 3) - 1));
  const int sibp = (L % tilesN) & 1;
  // (lab, r05, STREAM TIMING ONLY -- the LDS image is incomplete and the counted waits see fewer operations):
  //   OPT bit 12: only the EVEN column tile of a row tile asks for the gathered rows (108 pullers x 1.5 MB, no duplicate requests);
  //   OPT bit 13: every sibling asks only for ITS half of the rows (216 pullers x 0.75 MB, no duplicate requests)
  auto a_issue = [&](int tt0, int slot, int i) {
    if ((OPT & 4096) && sibp && tt0 >= P) return;
    if ((OPT & 8192) && ((((i * 8 + wave) * 8 >= BMT / 2) ? 1 : 0) != sibp) && tt0 >= P) return;
    int tt = rotk(tt0);
    if (blead && (((i * 8 + wave) * 8 >= BMT / 2) ? 1 : 0) == sibp) { tt += blead; if (tt >= nk) tt -= nk; }
    if ((OPT & 16) && (!(ABL & 1) || tt0 < P)) { dr_glds16<(OPT >> 2) & 3>(aoff2[tt % RUNG][i], a.table + (int64_t)(tt - tt % RUNG) * BK, lds0 + slot * SLOTB + (i * 8 + wave) * 1024); return; }
    if (!(ABL & 1) || tt0 < P) dr_glds16<(OPT >> 2) & 3>(aoff[i], a.table + (int64_t)tt * BK, lds0 + slot * SLOTB + (i * 8 + wave) * 1024);
  };
  // issue group of K-tile tt, first / second part (the same order in the prologue and in the loop)
#define DR_ISSUE0(tt, J, slot)                                                                        \
  { const uint16_t* wp_ = wbase + (int64_t)rotk(tt) * 2048;                                           \
    if (!(ABL & 16)) { dr_gload16<0>(bq[J][0][0], voff, wp_); dr_gload16<1024>(bq[J][0][1], voff, wp_); }  \
    _Pragma("unroll") for (int i_ = 0; i_ < N0; ++i_) a_issue(tt, slot, i_); }
#define DR_ISSUE1(tt, J, slot)                                                                        \
  { const uint16_t* wp_ = wbase + (int64_t)rotk(tt) * 2048;                                           \
    if (!(ABL & 16)) { dr_gload16<2048>(bq[J][1][0], voff, wp_); dr_gload16<3072>(bq[J][1][1], voff, wp_); } \
    _Pragma("unroll") for (int i_ = N0; i_ < NPW; ++i_) a_issue(tt, slot, i_); }

  if (ABL & 16) {
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
      for (int x = 0; x < 4; ++x) bq[j][x >> 1][x & 1] = i16x8{1, 2, 3, 4, 5, 6, 7, (short)(j + x)};
  }
  // prologue: K-tiles 0 .. P-1 into slots 0 .. P-1
  DR_ISSUE0(0, 0, 0) DR_ISSUE1(0, 0, 0) DR_ISSUE0(1, 1, 1) DR_ISSUE1(1, 1, 1)
  if constexpr (P == 4) { DR_ISSUE0(2, 2, 2) DR_ISSUE1(2, 2, 2) DR_ISSUE0(3, 3, 3) DR_ISSUE1(3, 3, 3) }
  if constexpr (TD == 8) touch(4);
  // fragment addresses: row (mi 16 + frow), chunk (kk 4 + fq) ^ (frow & 7)
  const int frow = lane & 15, fq = lane >> 4, sw = frow & 7;
  const unsigned rl0 = lds0 + frow * 128 + ((fq ^ sw) << 4), rl1 = lds0 + frow * 128 + (((4 + fq) ^ sw) << 4);
  asm volatile("s_waitcnt vmcnt(%[cnt])"
               : "+v"(bq[0][0][0]), "+v"(bq[0][0][1]), "+v"(bq[0][1][0]), "+v"(bq[0][1][1]) : [cnt] "n"((P - 1) * GRP + (TD == 8 ? NTI : 0)) : "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int x = 0; x < AD; ++x) {
    if (!(ABL & 4)) af[x] = DR_LDSRD(rl0 + x * 2048); else af[x] = i16x8{1, 2, 3, 4, 5, 6, 7, (short)x};
  }

  int sl_r = 0;                                // slot of K-tile t
  int sl_i = P;                                // slot of K-tile t + P
#define DR_MM(mi, B)                                                                                  \
  if (!(ABL & 2)) { acc[mi][0] = T::mfma(B[0], af[(mi) % AD], acc[mi][0]); acc[mi][1] = T::mfma(B[1], af[(mi) % AD], acc[mi][1]); }
#define DR_RD(mi, addr, smi) if (!(ABL & 4)) af[(mi) % AD] = DR_LDSRD((addr) + (smi) * 2048);
#define DR_PIN(n)                                                                                     \
  _Pragma("unroll") for (int g_ = 0; g_ < (n); ++g_) {                                                \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
  // one K-tile step.  J = t % P (register buffer of B); TAIL = -1: steady state, TAIL = j >= 0: step nk - P + j (nothing left to issue)
  auto step = [&](int t, auto j_c, auto tail_c) {
    constexpr int J = decltype(j_c)::value, TAIL = decltype(tail_c)::value;
    constexpr int JN = (J + 1) % P;
    const unsigned rc0 = rl0 + sl_r * SLOTB, rc1 = rl1 + sl_r * SLOTB;
    const int sl_n = sl_r + 1 == NS ? 0 : sl_r + 1;
    const unsigned rn0 = rl0 + sl_n * SLOTB;
    // ---- first half: kk = 0
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      DR_MM(mi, bq[J][0])
      if (mi + AD < MT) { DR_RD(mi, rc0, mi + AD) } else { DR_RD(mi, rc1, mi + AD - MT) }
    }
    DR_PIN(MT)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL < 0 && TD > 0 && J == 0) touch(t + TD);
    if constexpr (TAIL < 0) DR_ISSUE0(t + P, J, sl_i)
    __builtin_amdgcn_sched_barrier(0);
    // ---- second half: kk = 1
#pragma unroll
    for (int mi = 0; mi < MT - AD; ++mi) { DR_MM(mi, bq[J][1]) DR_RD(mi, rc1, mi + AD) }
    DR_PIN(MT - AD)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL != P - 1) {
      // everything issued in step t + 1 - P has landed: this wave's pieces of A(t+1), its B(t+1)
      constexpr int CNT = TAIL < 0 ? (P - 2) * GRP + 2 + N0 + (TD > 0 && J != 3 ? NTI : 0) : (P - 2 - TAIL > 0 ? (P - 2 - TAIL) * GRP : 0);
      if constexpr (OPT & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(%[cnt])"
                   : "+v"(bq[JN][0][0]), "+v"(bq[JN][0][1]), "+v"(bq[JN][1][0]), "+v"(bq[JN][1][1]) : [cnt] "n"(CNT) : "memory");
      __builtin_amdgcn_s_barrier();            // (hipcc waits for this wave's outstanding fragment reads in front of it: A(t) is free)
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = MT - AD; mi < MT; ++mi) { DR_MM(mi, bq[J][1]) if constexpr (TAIL != P - 1) { DR_RD(mi, rn0, mi + AD - MT) } }
    DR_PIN(AD)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TAIL < 0) DR_ISSUE1(t + P, J, sl_i)
    __builtin_amdgcn_sched_barrier(0);
    sl_r = sl_n;
    sl_i = sl_i + 1 == NS ? 0 : sl_i + 1;
  };
  using IM = std::integral_constant<int, -1>;
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  int t = 0;
  if constexpr (P == 4) {
    for (; t < nk - 4; t += 4) { step(t, I0{}, IM{}); step(t + 1, I1{}, IM{}); step(t + 2, I2{}, IM{}); step(t + 3, I3{}, IM{}); }
    step(t, I0{}, I0{}); step(t + 1, I1{}, I1{}); step(t + 2, I2{}, I2{}); step(t + 3, I3{}, I3{});
  } else {
    for (; t < nk - 2; t += 2) { step(t, I0{}, IM{}); step(t + 1, I1{}, IM{}); }
    step(t, I0{}, I0{}); step(t + 1, I1{}, I1{});
  }
#undef DR_MM
#undef DR_RD
#undef DR_PIN
#undef DR_ISSUE0
#undef DR_ISSUE1

  // Epilogue: descale, bias, ReLU.  The MFMA was issued with the operands swapped (D' = W_tile X_tile^T): the lane's
  // column is the batch row m and its 4 registers are 4 consecutive outputs n -> one 16-B store.
  const float descale = 1.0f / (a.scales->sx * a.scales->sw_cur);
  const float lo = a.relu ? 0.f : -INFINITY;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = n0 + wave * 32 + ni * 16 + fq * 4;
    if (n >= a.D) continue;
    float bj[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bj[j] = (a.bias && n + j < a.D) ? a.bias[n + j] : 0.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
      const int m = m0 + mi * 16 + frow;
      if (m >= R) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[mi][ni][j] * descale + bj[j], lo);
      if ((ABL & 64) && v[0] != 12345.f) continue;
      *(float4*)(a.H + (int64_t)m * a.D + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

}  // namespace vv
