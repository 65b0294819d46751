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
#include <type_traits>
#include <cstdio>
#include <cstdlib>
#include "vv_internal.h"

namespace vv {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// LDS-DMA issue as inline assembly.  Through __builtin_amdgcn_global_load_lds the compiler knows the instruction
// writes LDS and, unable to tell the ring's slots apart, puts `s_waitcnt vmcnt(0)` in front of every later ds_read: the
// whole stream was drained once per phase and the counted waits below never had anything left to count (seen in the
// disassembly of both kernels; the lookahead the schedule is built on did not exist).  As assembly the instruction is
// opaque: ordering against the fragment reads is exactly what the schedule states -- this wave's counted vmcnt wait,
// then a workgroup barrier, before any wave reads a slot; a slot restaged only after its readers have passed a barrier
// with their reads complete.  M0 carries the wave-uniform LDS base (the builtin sets it the same way).
__device__ __forceinline__ void ph_glds16(const void* gsrc, void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_wave_base));
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(gsrc) : "m0", "memory");
}
// the same with the LDS address as a number (a wave-uniform byte offset into LDS: nothing to cast or null-check per use)
__device__ __forceinline__ void ph_glds16_at(const void* gsrc, unsigned lds_addr) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(gsrc) : "m0", "memory");
}
// ... with the non-temporal hint (a stream the L2 need not keep)
__device__ __forceinline__ void ph_glds16_at_nt(const void* gsrc, unsigned lds_addr) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off nt" :: "s"(m0v), "v"(gsrc) : "m0", "memory");
}
// the same with the source 256 bytes further on (the upper 128-column half of a k-major operand): an immediate
// offset of the instruction instead of a second address register pair.  The hardware adds the instruction offset to
// the LDS address as well as to the global one (LDS_ADDR = M0 base + inst_offset + lane * size), so the LDS base is
// handed over 256 bytes low.  Only used for slots >= 2: the adjusted base never falls below the start of the LDS.
__device__ __forceinline__ void ph_glds16_hi(const void* gsrc, unsigned char* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_wave_base - 256));
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off offset:256" :: "s"(m0v), "v"(gsrc) : "m0", "memory");
}
// The LEAN form (round 6, k_wgrad_gemm_ph): the source as a wave-uniform base in scalar registers + a 32-bit byte offset per lane, the LDS
// address as (the wave's LDS base in a scalar register) + an immediate, added straight into M0: two instructions per issue, no address
// arithmetic on the vector unit beyond the offset itself.  ldsimm and goff are literals; with goff the hardware moves the LDS address too
// (see ph_glds16_hi): the caller takes it off ldsimm.
#define PH_GLDS_S(lds_w, voff, sbase, ldsimm, goff)                                                    \
  asm volatile("s_add_u32 m0, %0, %3\n\tglobal_load_lds_dwordx4 %1, %2 offset:" #goff                 \
               :: "s"(lds_w), "v"(voff), "s"(sbase), "n"(ldsimm) : "m0", "scc", "memory")
__device__ __forceinline__ int ph_xcd_remap(int bid, int nblk) {
  const int x = bid & 7, q = nblk >> 3, rem = nblk & 7;
  return x * q + (x < rem ? x : rem) + (bid >> 3);
}

// (drop_mix32, the dropout mask's counter hash: vv_internal.h, DropSpec)
constexpr int PH_SLOT = 16384;                 // one half-tile: 128 rows x 64 halves
constexpr int PH_LDS_BYTES = 8 * PH_SLOT;      // ring of 8 slots = 128 KiB
#define PH_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// ------------------------------------------------------------------------------- forward ------
// Half-tile LDS image: [128 rows][64 halves] = 128-B rows of 8 16-B chunks, chunk' = chunk ^ (row & 7)
// (conflict-free ds_read_b128 fragment reads, as in k_fwd_gemm).  One LDS-DMA wave-instruction = 8 rows.
// MQ = 16-row MFMA tiles per wave and A half: the tile is (64*MQ) x 256 (MQ 4: 256 rows, 3: 192, 2: 128).
// GATE: the W operand may still be arriving chunk by chunk (FwdArgs::gate).  Every wave checks all chunk flags once at
// its start -- nothing of its stream is in flight yet, the check costs one round trip -- and when all are set (one GPU,
// or an all-reduce that finished early: the usual case) the loop runs exactly as without gating.  Otherwise the
// workgroup waits in front of its first B_lo issue of each chunk (B_lo(kt) is the first W access of K-tile kt: phase 3 of
// K-tile kt - 2): WAVE 0 polls the chunk's flag (agent scope, s_sleep between polls: one poller per workgroup -- every
// wave of 216 workgroups polling one line slowed the very kernels that were to set it) and runs the agent-scope acquire
// (this CU's L1 may not serve the chunk from before the update); a workgroup barrier then releases the other waves.  The
// barrier is one more in every wave's sequence: waves 4-7, one barrier behind, pass it one segment later, still behind
// wave 0's poll.  The poll is a vector load: it drains wave 0's own stream once per chunk (three times per kernel, and
// only while an update is really still in flight).
// DEAD (0 / 1): the LAST 16-row MFMA tile of the upper A half (waves 4-7's) is left out -- the tile covers 64 MQ - 16 rows.
// With MQ = 3 that is 176 rows: 236 instead of 216 workgroups for the benchmark's ~20 650 distinct rows, on 256 CUs.
// (Rows of the 128-row half-tile image that no wave reads are still staged, from the L2-hot zero row: dropping those
// LDS-DMA instructions -- the second one of waves 4-7 at 96 live rows -- with per-wave counted waits was measured and made
// the 192-row kernel 4.6 us SLOWER, profiles/r03_step_ablations.txt.)
// LEAD (0 / 1): the workgroups of one row tile (its column tiles: two at D = 512) ask for the same gathered lines at the
// same time, so each line costs BOTH of them a miss -- the second request finds the first still in flight and waits for HBM
// just as long (tools/lab/fwd_stream_lab.hip: the stream alone 104 us as issued today, 140 us when the siblings read
// different rows, 74 us with every row L2-hot).  With LEAD the even column tiles ask for their A_lo half-tile ONE K-TILE
// EARLY and the odd ones for their A_hi half-tile: every line's missing request comes from one sibling, and the other's,
// a K-tile later, is an L2 hit (lab: 92 us; two or more K-tiles of lead are slower again).  The stream keeps its order and
// its counts -- only the address of the leading half-tile's instructions moves a K-tile on -- so the counted waits hold
// as they are.  Nothing of it may cost the LOAD segment anything (a first build with run-time ring positions lost more there
// than the stream gained): in the code the leading half is ALWAYS A_lo -- odd column tiles swap which rows are their lower and
// upper half instead --, it lives one K-tile longer in a ring of FOUR slots (its two natural ones + a ninth and a tenth: all
// 160 KiB of LDS), and the loop is unrolled by four K-tiles so that every ring position is a constant.
// DROP: 0 no dropout, 1 the counter-based mask, 2 an explicit mask (FwdArgs::mask) -- separate instantiations: with the choice made per
// element at run time the epilogue was 700 branches and 50 KB of code, and the dense-size kernel took 17 us longer for it
// MRG (0 / 1; round 5): TWO phases per barrier pair.  The stamps (profiles/r05_fwd_stamps.txt) put a phase at 192 clocks of matrix pipe
// plus ~45 of barrier release plus what the partner's load segment overhangs: 512 intervals of ~250.  The fragments of a K-tile already
// live in registers together (A half 24 + B_lo 16 + B_hi 16 VGPRs), so the four phases pair up without a register more:
//   M0: read A_lo, B_lo, B_hi | stream steps 0, 1 | barrier | A_lo x B_lo, A_lo x B_hi (24 MFMAs) | barrier
//   M1: read A_hi             | stream steps 2, 3 | barrier | A_hi x B_hi, A_hi x B_lo (24 MFMAs) | barrier
// 256 intervals of ~430.  The stream keeps its order and its six half-tiles of distance; the counted waits move: before M1 everything
// but the four youngest half-tiles (A_hi(t) is the fifth youngest), before M0 everything but the THREE youngest (B_hi(t + 1), issued in
// M0(t), is the fourth youngest: the one half-tile whose flight is half a K-tile instead of a whole one -- it is a W tile, an L2 hit).
// O16 (round 6; FwdArgs::h16): ip2 leaves as f16 -- half the bytes of the kernel's store-bound epilogue (42 -> 21 MB at the benchmark's size)
// and of every later read of it.  A lane's two quads of a 32-column group (ni = 0, 1: columns fq 4 .. + 3 and 16 + fq 4 .. + 3, four halves
// = two registers each) are regrouped by two v_permlane16_swap into EIGHT consecutive columns per lane -- one 16-byte store where the fp32
// form issues two: half the store instructions as well (the tail of this epilogue is store-ISSUE-bound, cdna_hip_programming.md T21).
template <typename T, int DROP, bool VEC, int MQ, int ABL = 0, bool GATE = false, int DEAD = 0, int LEAD = 0, int MRG = 0, bool O16 = false>
__global__ __launch_bounds__(GEMM_THREADS) void k_fwd_gemm_ph(FwdArgs a) {
  static_assert(!O16 || (VEC && DROP == 0), "the f16 output form exists for the plain forward (D % 8 == 0, no dropout in the epilogue)");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HROWS = 32 * MQ;               // rows of the lower A half-tile (upper: HROWS - 16 DEAD)
  constexpr int BMT = 2 * HROWS - 16 * DEAD;   // rows of the output tile
  if (ABL & (512 | 2048)) asm volatile("s_memtime s[84:85]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s101, s84" ::: "s84", "s85", "s101");   // (lab: time stamps, below)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const int tilesN = Dp / BN;
  const int R = a.n_dev ? *a.n_dev : a.R;
  const int nact = a.n_dev ? ((R + BMT - 1) / BMT) * tilesN : (int)gridDim.x;
  if (a.seq_host && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.seq_host, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // (lab, ABL bit 13: the CEILING of a K-cut, timing only.  Every tile's workgroup runs 7/8 of its K-tiles (56 of 64: 13 824 K-tile units
  // over 256 CUs are 54 each) and 40 further workgroups -- the ones a K-cut would put on the CUs this launch leaves idle -- run the same 56
  // K-tiles on some tile and store nothing: no partial accumulators, no fix-up, no seam.  What the step gains with it is what a K-cut could
  // gain at most; the grouping kernels of the next step lose their 40 CUs as they would.)
  constexpr bool KCUT = (ABL & 8192) != 0;
  const bool kc_helper = KCUT && (int)blockIdx.x >= nact && (int)blockIdx.x < nact + 40;
  if ((int)blockIdx.x >= nact && !kc_helper) return;
  const int L = kc_helper ? (((int)blockIdx.x - nact) * 5) % nact : ph_xcd_remap(blockIdx.x, nact);
  const int m0 = (L / tilesN) * BMT, n0 = (L % tilesN) * BN;
  const int Fp = a.Fp;
  const int Rst = kc_helper ? 0 : R;            // rows this workgroup stores

  // staging sources: LDS-DMA instruction i (0, 1) of this wave fills rows (i*8 + wave)*8 .. +7 of a half-tile
  // (LEAD: odd column tiles hold the tile's UPPER rows in their "lower" half, the one that leads)
  const int hswap = LEAD ? ((L % tilesN) & 1) : 0;
  const uint16_t* srcA[2][2];
  const uint16_t* srcB[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 8 + (lane >> 3), lc = (lane & 7) ^ (row & 7);
      const int grow = m0 + (hf ^ hswap) * HROWS + row;
      const int trow = (row < HROWS - 16 * DEAD * hf && grow < R && !(ABL & 8)) ? a.rows[grow] : a.zero_row;
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

  const int nk = KCUT ? (Fp / BK) / 8 * 7 : Fp / BK;   // K-tiles (Fp is a multiple of 256: nk % 4 == 0)
  const int H = 4 * nk;                        // half-tiles of this workgroup's stream
  // half-tile h = 4*kt + q, q: 0 A_lo, 1 B_lo, 2 B_hi, 3 A_hi; slot = h & 7
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem)) + wave * 1024;   // this wave's first piece of slot 0
  // LEAD: A_lo of K-tile k lives in ring position k & 3 = slot 0, 4, 8, 9 (PH_RING); its place in the stream carries A_lo of the
  // NEXT K-tile (the last K-tile's place re-stages K-tile nk - 1 into the position that is free: counts stay exact)
#define PH_RING(r) (((r) & 3) == 0 ? 0 : ((r) & 3) == 1 ? 4 : ((r) & 3) == 2 ? 8 : 9)
  auto issue = [&](int kt, int q, int slot) {
    int k = kt;
    if (LEAD && q == 0) k = kt + 1 < nk ? kt + 1 : nk - 1;          // (slot: the caller's PH_RING position of K-tile kt + 1)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint16_t* src = q == 0 ? srcA[0][i] : q == 1 ? srcB[0][i] : q == 2 ? srcB[1][i] : srcA[1][i];
      if ((ABL & 128) && (q == 0 || q == 3)) ph_glds16_at_nt(src + k * BK, lds_wave + slot * PH_SLOT + i * 8192);   // (lab, ABL 128: gathered rows non-temporal)
      else if ((ABL & 256) && (q == 1 || q == 2)) ph_glds16_at_nt(src + k * BK, lds_wave + slot * PH_SLOT + i * 8192);   // (lab, ABL 256: W non-temporal)
      else ph_glds16_at(src + k * BK, lds_wave + slot * PH_SLOT + i * 8192);
    }
  };
#define PH_WAITQ() PH_WAIT(8)                  /* everything but the four youngest half-tiles has landed */

  // chunk gates (GATE only)
  bool gated = false;
  int g_next = 1;                              // the next chunk whose gate lies inside the K loop
  int gate_at = -1;                            // ... and the K-tile in front of whose B_lo issue it sits (-1: none; one compare per K-tile in the loop)
  auto gate_wait = [&](int chunk) {
    if (wave == 0) {
      unsigned spins = 0;
      while ((int32_t)(__hip_atomic_load(a.gate + chunk * W_GATE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.gate_seq) < 0) {
        __builtin_amdgcn_s_sleep(64);
        if (++spins > 12000000u) {             // ~half a minute: the update never came (a failed rank, a bug): flag it and go on
          if (lane == 0) __hip_atomic_store(a.gate_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __builtin_amdgcn_s_barrier();
  };
  if (GATE) {
    // (every wave reads the flags itself here: `gated` must be the same in all eight, and a flag only ever grows --
    // a wave that sees all set while another does not is excluded by re-reading after a barrier)
    __shared__ int s_behind;
    if (tid == 0) {
      int behind = 0;
#pragma unroll
      for (int cch = 0; cch < W_CHUNKS_MAX; ++cch)
        if (cch < a.gate_n) behind |= (int32_t)(__hip_atomic_load(a.gate + cch * W_GATE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.gate_seq) < 0;
      s_behind = behind;
    }
    __syncthreads();
    gated = s_behind != 0;
    if (gated) {
      // chunk 0 in front of the prologue; with it every chunk that starts before K-tile 4 (the prologue's loads reach K-tile 1)
      gate_wait(0);
      while (g_next < a.gate_n && a.gate_kt[g_next] < 4) { gate_wait(g_next); ++g_next; }
      if (g_next >= a.gate_n) gated = false;
      gate_at = gated ? a.gate_kt[g_next] : -1;
    } else if (wave == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the flags were read after the kernel began: order the W loads behind them
  }

  // prologue: half-tiles 0 .. 5 (LEAD: A_lo of K-tile 0 in front of them, and every A_lo place carries the next K-tile's)
  if (LEAD) {
#pragma unroll
    for (int i = 0; i < 2; ++i) ph_glds16_at(srcA[0][i], lds_wave + PH_RING(0) * PH_SLOT + i * 8192);
    issue(0, 0, PH_RING(1)); issue(0, 1, 1); issue(0, 2, 2); issue(0, 3, 3); issue(1, 0, PH_RING(2)); issue(1, 1, 5);
  } else {
    issue(0, 0, 0); issue(0, 1, 1); issue(0, 2, 2); issue(0, 3, 3); issue(1, 0, 4); issue(1, 1, 5);
  }
  if (MRG) PH_WAIT(6);                         // MRG: A_lo, B_lo AND B_hi of K-tile 0 (all but the three youngest half-tiles)
  else PH_WAITQ();                             // half-tiles 0, 1 (A_lo, B_lo of K-tile 0) have landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();   // waves 4-7 run one segment behind
  if (wm == 1) __builtin_amdgcn_s_setprio(1);  // the younger half loses VALU arbitration otherwise

  const int frow = lane & 15, fq = lane >> 4;
  const int a_off = (wm * 16 * MQ + frow) * 128;          // + mi*16*128, chunk by kk
  const int b_off = (wn * 32 + frow) * 128;               // + ni*16*128
  const int sw = frow & 7;
  i16x8 af[MQ][2], b0[2][2], b1[2][2];

  // one phase: LOAD segment (reads + stream + wait), barrier, MFMA segment, barrier
  const bool dead_hi = DEAD && wm == 1;        // this wave's last tile of the upper A half does not exist
#define PH_LOAD_A(slot)                                                                              \
  if (!abl_rd) {                                                                                     \
    constexpr int a_slot = (LEAD && ((slot) & 3) == 0) ? PH_RING(J) : (slot);                        \
    _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
    if (!(DEAD && ((slot) & 3) == 3 && mi == MQ - 1 && dead_hi))                                     \
    af[mi][kk] = *(const i16x8*)(smem + a_slot * PH_SLOT + a_off + mi * 2048 + (((kk * 4 + fq) ^ sw) << 4)); \
  }
#define PH_LOAD_B(dst, slot)                                                                         \
  if (!abl_rd) _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)  \
    dst[ni][kk] = *(const i16x8*)(smem + (slot) * PH_SLOT + b_off + ni * 2048 + (((kk * 4 + fq) ^ sw) << 4));
#define PH_MFMA(mh, nh, bfr)                                                                         \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  PH_TS("s[90:91]")                                                                                        \
  if (!abl_mm) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) \
    if (!(DEAD && (mh) == 1 && mi == MQ - 1 && dead_hi))                                             \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                 \
      acc[mh][mi][nh][ni] = T::mfma(bfr[ni][kk], af[mi][kk], acc[mh][mi][nh][ni]);                   \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  if (TS) {           /* the stamps consumed here were taken at least one MFMA segment ago */       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                          \
                 "s_sub_u32 s100, s92, s94\n\ts_add_u32 s98, s98, s100\n\t"   /* mfma += d - previous c */ \
                 "s_sub_u32 s100, s84, s92\n\ts_add_u32 s99, s99, s100\n\t"   /* bar2 += e - d */          \
                 "s_sub_u32 s100, s86, s84\n\ts_add_u32 s95, s95, s100\n\t"   /* load += b0 - e */         \
                 "s_sub_u32 s100, s88, s86\n\ts_add_u32 s96, s96, s100\n\t"   /* vmwait += b - b0 */       \
                 "s_sub_u32 s100, s90, s88\n\ts_add_u32 s97, s97, s100\n\t"   /* bar1 += c - b */          \
                 "s_mov_b32 s94, s90\n\ts_memtime s[92:93]" ::: PH_TS_CLOB);                                  \
    __builtin_amdgcn_sched_barrier(0);                                                               \
  }                                                                                                  \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  PH_TS("s[84:85]")
  // stream step of phase (t, p): half-tile 4t + p + 6, then the counted wait for what the next phase reads
#define PH_STREAM(tpar, t, p, wait)                                                                  \
  {                                                                                                  \
    const int h = 4 * (t) + (p) + 6;                                                                 \
    constexpr int q_ = ((p) + 2) & 3;                                                                \
    constexpr int slot_ = (LEAD && q_ == 0) ? PH_RING(J + 3) : 4 * (((tpar) + (((p) + 6) >> 2)) & 1) + q_;   /* A_lo's place (p = 2) carries K-tile t + 3 */ \
    if ((!CHK || h < H) && !abl_st) { issue(h >> 2, q_, slot_); PH_TS("s[86:87]") if (wait) PH_WAITQ(); }  \
    else { PH_TS("s[86:87]") if (wait) PH_WAIT(0); }                                                    \
    PH_TS("s[88:89]")                                                                              \
  }
  // MRG: the two MFMA blocks of a merged phase between ONE pair of barriers
#define PH_MFMA2(mhA, nhA, bfrA, mhB, nhB, bfrB)                                                     \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  if (!abl_mm) {                                                                                     \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) \
      if (!(DEAD && (mhA) == 1 && mi == MQ - 1 && dead_hi))                                          \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                               \
        acc[mhA][mi][nhA][ni] = T::mfma(bfrA[ni][kk], af[mi][kk], acc[mhA][mi][nhA][ni]);            \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) \
      if (!(DEAD && (mhB) == 1 && mi == MQ - 1 && dead_hi))                                          \
      _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                               \
        acc[mhB][mi][nhB][ni] = T::mfma(bfrB[ni][kk], af[mi][kk], acc[mhB][mi][nhB][ni]);            \
  }                                                                                                  \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);
  // stream step with the merged schedule's wait in front of M0 (all but the three youngest half-tiles)
#define PH_STREAM6(tpar, t, p)                                                                       \
  {                                                                                                  \
    const int h = 4 * (t) + (p) + 6;                                                                 \
    constexpr int q_ = ((p) + 2) & 3;                                                                \
    constexpr int slot_ = (LEAD && q_ == 0) ? PH_RING(J + 3) : 4 * (((tpar) + (((p) + 6) >> 2)) & 1) + q_; \
    if ((!CHK || h < H) && !abl_st) { issue(h >> 2, q_, slot_); PH_WAIT(6); }                        \
    else { PH_WAIT(0); }                                                                             \
  }
  // ABL: timing studies only (VV_ABLATE, results wrong): 1 no LDS-DMA stream in the loop, 2 no MFMA, 4 no fragment
  // reads, 8 every gathered row is the (L2-hot) zero row
  constexpr bool abl_st = ABL & 1, abl_mm = ABL & 2, abl_rd = ABL & 4;
  // (lab, ABL 512: TIME STAMPS.  Every wave reads the shader clock (s_memtime) right after each of the two barriers of a phase, after its
  // stream issue, after its counted wait and after its last MFMA issue, and keeps five sums -- load segment up to the wait / the counted
  // vmcnt wait / barrier behind the load segment / MFMA segment / barrier behind it; lane 0 of every wave writes them, the kernel's cycles
  // and its 100 MHz real time to FwdArgs::mask (as uint32[12] per wave) at the end.  tools/lab/fwd_dr_lab.hip prints them.
  // Stamps and sums live in FIXED scalar registers s78..s101 that only these assembly blocks name (the kernel itself needs ~62 SGPRs and
  // the allocator hands them out from s0 upwards; as compiler-visible values every stamp was spilled to a VGPR lane on the spot -- 700
  // v_writelane / v_readlane in the loop -- and as `s_memtime` builtins each one was followed by s_waitcnt lgkmcnt(0), draining the wave's
  // fragment reads).  A stamp is read only behind the explicit lgkmcnt(0) of the consuming block, which sits behind the phase's MFMAs --
  // where the fragments have been waited for anyway.
  //   s[84:85] e (after the barrier behind the MFMA segment = start of the load segment)   s[86:87] b0 (stream issued)
  //   s[88:89] b (counted wait passed)   s[90:91] c (after the barrier behind the load segment)   s[92:93] d (last MFMA issued)
  //   s94 previous c   s95..s99 sums: load, vmwait, bar1, mfma, bar2   s100 scratch   s101 kernel start   s83 loop start   s[78:79] real time)
  constexpr bool TS = (ABL & 512) != 0;
  constexpr bool TSL = (ABL & 2048) != 0;      // (lab: only the kernel's four marks -- start, loop start, loop end, last store -- and the real time: no per-phase stamps)
#define PH_TS_CLOB "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "scc"
#define PH_TS(pair) if (TS) { asm volatile("s_memtime " pair ::: PH_TS_CLOB); __builtin_amdgcn_sched_barrier(0); }
  if (TS || TSL) {
    asm volatile("s_memrealtime s[78:79]\n\ts_memtime s[84:85]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 s83, s84\n\ts_mov_b32 s86, s84\n\ts_mov_b32 s88, s84\n\ts_mov_b32 s90, s84\n\ts_mov_b32 s92, s84\n\ts_mov_b32 s94, s84\n\t"
                 "s_mov_b32 s95, 0\n\ts_mov_b32 s96, 0\n\ts_mov_b32 s97, 0\n\ts_mov_b32 s98, 0\n\ts_mov_b32 s99, 0" ::: PH_TS_CLOB);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (abl_rd) {
#pragma unroll
    for (int mi = 0; mi < MQ; ++mi) { af[mi][0] = i16x8{1, 2, 3, 4, 5, 6, 7, (short)mi}; af[mi][1] = af[mi][0]; }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) { b0[ni][0] = b0[ni][1] = b1[ni][0] = b1[ni][1] = i16x8{1, 2, 3, 4, 5, 6, 7, (short)ni}; }
  }

  // one K-tile of parity PAR (slots 4 PAR ..), number J of its group of four (LEAD's ring positions); CHK: the stream may have
  // run out (only the last K-tiles can see that: a K-tile's issues reach two K-tiles on)
  auto ktile = [&](int t, auto par_c, auto j_c, auto chk) {
    constexpr int PAR = decltype(par_c)::value, J = decltype(j_c)::value;
    constexpr bool CHK = decltype(chk)::value;
    (void)J;
    if constexpr ((ABL & 1024) != 0) {               // (lab, ABL bit 10: the phase's LDS-DMA instructions in front of its fragment reads)
#define PH_WAITSF(p) if ((!CHK || 4 * t + (p) + 6 < H) && !abl_st) { PH_WAITQ(); } else { PH_WAIT(0); }
      PH_STREAM(PAR, t, 0, false) PH_LOAD_A(4 * PAR + 0) PH_LOAD_B(b0, 4 * PAR + 1) PH_WAITSF(0) PH_MFMA(0, 0, b0)
      PH_STREAM(PAR, t, 1, false) PH_LOAD_B(b1, 4 * PAR + 2) PH_WAITSF(1) PH_MFMA(0, 1, b1)
#undef PH_WAITSF
      PH_STREAM(PAR, t, 2, false) PH_LOAD_A(4 * PAR + 3) PH_MFMA(1, 1, b1)
    } else if constexpr (MRG != 0) {
      PH_LOAD_A(4 * PAR + 0) PH_LOAD_B(b0, 4 * PAR + 1) PH_LOAD_B(b1, 4 * PAR + 2)
      PH_STREAM(PAR, t, 0, false) PH_STREAM(PAR, t, 1, true)
      PH_MFMA2(0, 0, b0, 0, 1, b1)
      PH_LOAD_A(4 * PAR + 3) PH_STREAM(PAR, t, 2, false)
      if (GATE && PAR == 0 && t + 2 == gate_at) {                        // B_lo(t + 2) opens a chunk
        gate_wait(g_next);
        if (++g_next >= a.gate_n) gated = false;
        gate_at = gated ? a.gate_kt[g_next] : -1;
      }
      PH_STREAM6(PAR, t, 3)
      PH_MFMA2(1, 1, b1, 1, 0, b0)
      return;
    } else {
    PH_LOAD_A(4 * PAR + 0) PH_LOAD_B(b0, 4 * PAR + 1) PH_STREAM(PAR, t, 0, true) PH_MFMA(0, 0, b0)
    PH_LOAD_B(b1, 4 * PAR + 2) PH_STREAM(PAR, t, 1, true) PH_MFMA(0, 1, b1)
    PH_LOAD_A(4 * PAR + 3) PH_STREAM(PAR, t, 2, false) PH_MFMA(1, 1, b1)
    }
    if (GATE && PAR == 0 && t + 2 == gate_at) {                          // B_lo(t + 2) opens a chunk
      gate_wait(g_next);
      if (++g_next >= a.gate_n) gated = false;
      gate_at = gated ? a.gate_kt[g_next] : -1;
    }
    PH_STREAM(PAR, t, 3, true) PH_MFMA(1, 0, b0)
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  if (LEAD) {                                  // groups of four K-tiles (nk % 4 == 0)
    int t = 0;
    for (; t < nk - 4; t += 4) {
      ktile(t, I0{}, I0{}, std::false_type{}); ktile(t + 1, I1{}, I1{}, std::false_type{});
      ktile(t + 2, I0{}, I2{}, std::false_type{}); ktile(t + 3, I1{}, I3{}, std::false_type{});
    }
    ktile(t, I0{}, I0{}, std::true_type{}); ktile(t + 1, I1{}, I1{}, std::true_type{});
    ktile(t + 2, I0{}, I2{}, std::true_type{}); ktile(t + 3, I1{}, I3{}, std::true_type{});
  } else {
    int t = 0;
    for (; t < nk - 2; t += 2) { ktile(t, I0{}, I0{}, std::false_type{}); ktile(t + 1, I1{}, I1{}, std::false_type{}); }
    ktile(t, I0{}, I0{}, std::true_type{}); ktile(t + 1, I1{}, I1{}, std::true_type{});
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();   // waves 0-3 catch the extra barrier of waves 4-7
  if (TS || TSL) asm volatile("s_memtime s[80:81]\n\ts_waitcnt lgkmcnt(0)" ::: PH_TS_CLOB);      // s80: end of the K loop
#undef PH_LOAD_A
#undef PH_LOAD_B
#undef PH_MFMA
#undef PH_MFMA2
#undef PH_STREAM
#undef PH_STREAM6
#undef PH_WAITQ
#undef PH_RING

  // Epilogue: descale, bias, ReLU, dropout.  The MFMA was issued with the operands swapped (D' = W_tile X_tile^T):
  // the lane's column is the batch row m and its 4 registers are 4 consecutive outputs n -> one 16-B store.
  // (gated: the scale of the half copy was rewritten while this kernel ran -- an agent-scope load, not the scalar cache's copy)
  const float sw_now = GATE ? __hip_atomic_load(&a.scales->sw_cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.scales->sw_cur;
  const float descale = 1.0f / (a.scales->sx * sw_now);
  const float dscale = DROP ? 1.0f / (1.0f - a.drop_ratio) : 1.0f;
  const float descale_d = descale * dscale;
  const float rcp_cn = DROP ? 1.0f / (float)a.CN : 0.f;
  const uint32_t drop_thr = DROP ? (uint32_t)(a.drop_ratio * 65536.f + 0.5f) : 0u;      // keep <=> 16-bit uniform >= thr
  const uint32_t drop_s32 = DROP ? (uint32_t)(mix64(a.drop_seed, 0x5eedull) >> 32) : 0u;  // the step's stream
  const float lo = a.relu ? 0.f : -INFINITY;
  // The lane's 16 bias values (its four output quads), loaded ONCE and all together.  Round 5, from the kernel's own time stamps
  // (profiles/r05_fwd_stamps.txt): with the loads inside the store loop hipcc put four scalar-dword loads and an s_waitcnt vmcnt(0)
  // in front of EVERY one of the 24 stores -- vmcnt counts stores too, so each quad waited for its bias round trip AND for the previous
  // quad's store to be acknowledged: 24 serial round trips, 23 000 clocks (11 us) from the last MFMA to the last store.
  float bq[2][2][4];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + nh * 128 + wn * 32 + ni * 16 + fq * 4;
      if (VEC) {                               // (D % 4 == 0 and n % 4 == 0: the quad is inside [0, D) or outside as a whole)
        const float4 b4 = n < a.D ? *(const float4*)(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
        bq[nh][ni][0] = b4.x * dscale; bq[nh][ni][1] = b4.y * dscale; bq[nh][ni][2] = b4.z * dscale; bq[nh][ni][3] = b4.w * dscale;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) bq[nh][ni][j] = (n + j < a.D ? a.bias[n + j] : 0.f) * dscale;
      }
    }
  if constexpr (O16) {
    uint16_t* H16 = (uint16_t*)a.H;
    const float lo16 = fmaxf(lo, -65504.f);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int mi = 0; mi < MQ; ++mi) {
        if (DEAD && mh == 1 && mi == MQ - 1 && dead_hi) continue;        // (wave-uniform)
        const int m = m0 + (mh ^ hswap) * HROWS + wm * 16 * MQ + mi * 16 + frow;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh) {
          uint32_t p[2][2];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
            _Float16 h[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              h[j] = (_Float16)fminf(fmaxf(acc[mh][mi][nh][ni][j] * descale_d + bq[nh][ni][j], lo16), 65504.f);   // (saturated: an inf here would be a NaN loss)
            p[ni][0] = (uint32_t)__builtin_bit_cast(uint16_t, h[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[1]) << 16);
            p[ni][1] = (uint32_t)__builtin_bit_cast(uint16_t, h[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[3]) << 16);
          }
          // rows of 16 lanes: (first, second)' = ([f0 s0 f2 s2], [f1 s1 f3 s3]) -- the lane of row r then holds the eight columns
          // (r & 1) 16 + (r >> 1) 8 .. + 7 of the group: first' its lower four, second' its upper four
          const auto s0 = __builtin_amdgcn_permlane16_swap(p[0][0], p[1][0], false, false);
          const auto s1 = __builtin_amdgcn_permlane16_swap(p[0][1], p[1][1], false, false);
          const int n = n0 + nh * 128 + wn * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
          if ((ABL & 64) && s0[0] != 12345u) continue;                    // (lab, ABL 64: no stores)
          if (m < Rst && n < a.D) *(uint4*)(H16 + (int64_t)m * a.D + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
  } else
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mi = 0; mi < MQ; ++mi) {
      if (DEAD && mh == 1 && mi == MQ - 1 && dead_hi) continue;
      const int m = m0 + (mh ^ hswap) * HROWS + wm * 16 * MQ + mi * 16 + frow;
      if (m >= Rst) continue;
      int64_t ref_row = 0;
      uint32_t row_ctr = 0;                    // counter of the row's first quad of outputs (two counters per quad)
      if (DROP) {
        // m = bb CN + ch without an integer division (m < 2^24: exact in fp32 up to the correction step)
        int bb = (int)((float)m * rcp_cn), ch = m - bb * a.CN;
        if (ch < 0) { --bb; ch += a.CN; } else if (ch >= a.CN) { ++bb; ch -= a.CN; }
        ref_row = (int64_t)ch * a.B + bb;
        row_ctr = drop_row_ctr(ref_row, a.D, drop_s32);
      }
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = n0 + nh * 128 + wn * 32 + ni * 16 + fq * 4;
          if (n >= a.D) continue;
          float v[4];
          // counter-based mask: ONE 32-bit hash serves two outputs (16-bit uniforms: the drop probability is ratio to 2^-16), two hashes
          // the lane's four -- instead of a 64-bit splitmix per output (+30 us on the 0.19 ms dense-size kernel)
          uint32_t u16x2[2] = {0u, 0u};
          if (DROP == 1) {
            const uint32_t c0 = row_ctr + (uint32_t)(n >> 1);                 // (n is a multiple of 4: quad n / 4, counters 2 (n / 4), + 1)
            u16x2[0] = drop_mix32(c0); u16x2[1] = drop_mix32(c0 + 1u);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            // (dropout: the 1 / (1 - ratio) of the kept values rides in the descale and the bias -- ReLU commutes with a positive factor)
            v[j] = fmaxf(acc[mh][mi][nh][ni][j] * descale_d + bq[nh][ni][j], lo);
            if (DROP) {
              bool keep;
              if (DROP == 2) keep = (n + j < a.D) && a.mask[(uint64_t)(ref_row * a.D + n + j)] != 0;
              else keep = ((u16x2[j >> 1] >> (16 * (j & 1))) & 0xffffu) >= drop_thr;
              v[j] = keep ? v[j] : 0.f;
            }
          }
          float* dst = a.H + (int64_t)m * a.D + n;
          if ((ABL & 64) && v[0] != 12345.f) continue;                    // (lab, ABL 64: no stores)
          if (VEC) *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
          else
            for (int j = 0; j < 4; ++j) if (n + j < a.D) dst[j] = v[j];
        }
    }
  if (TS || TSL) {
    uint32_t o_[12];
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime s[84:85]\n\ts_memrealtime s[86:87]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 %0, s95\n\ts_mov_b32 %1, s96\n\ts_mov_b32 %2, s97\n\ts_mov_b32 %3, s98\n\ts_mov_b32 %4, s99\n\t"
                 "s_sub_u32 %5, s83, s101\n\t"        /* up to the loop's first phase (prologue: row ids, first six half-tiles, first wait) */
                 "s_sub_u32 %6, s80, s101\n\t"        /* ... to the end of the K loop */
                 "s_sub_u32 %7, s84, s101\n\t"        /* ... to the last store's completion */
                 "s_sub_u32 %8, s86, s78\n\t"         /* 100 MHz ticks from the loop's start to the end */
                 "s_sub_u32 %9, s84, s83"              /* shader clocks over the same span */
                 : "=s"(o_[0]), "=s"(o_[1]), "=s"(o_[2]), "=s"(o_[3]), "=s"(o_[4]), "=s"(o_[5]), "=s"(o_[6]), "=s"(o_[7]), "=s"(o_[8]), "=s"(o_[9])
                 :: PH_TS_CLOB, "memory");
    if (lane == 0) {
      uint32_t* o = (uint32_t*)a.mask + ((size_t)blockIdx.x * 8 + wave) * 12;
#pragma unroll
      for (int j = 0; j < 10; ++j) o[j] = o_[j];
      o[10] = 0; o[11] = 0;
    }
  }
#undef PH_TS
#undef PH_TS_CLOB
}

#ifdef VV_LAB
// ------------------------------------------------------------------------------- forward, ten-slot ring (experiment) ----
// The same kernel with the WHOLE LDS as its ring: ten slots of 16 KiB, half-tiles issued EIGHT ahead of their use (two full
// K-tiles), counted wait vmcnt(12): six half-tiles (96 KiB per CU) stay in flight across every barrier instead of four.
// The ablations say the kernel sits on its LDS-DMA stream; if that stream is latency-bound (bytes in flight / round trip),
// half as many bytes again in flight should show.  Slots are runtime values here (half-tile h lives in slot h mod 10: the
// pattern repeats every five K-tiles, the loop is unrolled by two).  VV_GEMM_VARIANT=10 selects it (plain forward only).
template <typename T, int MQ>
__global__ __launch_bounds__(GEMM_THREADS) void k_fwd_gemm_ph10(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NS = 10;
  constexpr int HROWS = 32 * MQ, BMT = 2 * HROWS;
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
  const int nk = Fp / BK;
  const int H = 4 * nk;
  auto issue = [&](int kt, int q, int slot) {
    const uint16_t* const* src = q == 0 ? srcA[0] : q == 1 ? srcB[0] : q == 2 ? srcB[1] : srcA[1];
    unsigned char* dst = smem + slot * PH_SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) ph_glds16(src[i] + kt * BK, dst + (i * 8 + wave) * 1024);
  };
  // prologue: half-tiles 0 .. 7 (K-tiles 0 and 1) into slots 0 .. 7
#pragma unroll
  for (int h = 0; h < 8; ++h) issue(h >> 2, h & 3, h);
  PH_WAIT(12);                                 // half-tiles 0, 1 have landed
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_setprio(1);
  const int frow = lane & 15, fq = lane >> 4;
  const int a_off = (wm * 16 * MQ + frow) * 128;
  const int b_off = (wn * 32 + frow) * 128;
  const int sw = frow & 7;
  i16x8 af[MQ][2], b0[2][2], b1[2][2];
  int base = 0;                                // slot of half-tile 4 t
  auto slot_of = [&](int j) { const int x = base + j; return x >= NS ? (x >= 2 * NS ? x - 2 * NS : x - NS) : x; };   // j < 16
#define P10_LOAD_A(j)                                                                                \
  { const unsigned char* sp_ = smem + slot_of(j) * PH_SLOT;                                          \
    _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) \
      af[mi][kk] = *(const i16x8*)(sp_ + a_off + mi * 2048 + (((kk * 4 + fq) ^ sw) << 4)); }
#define P10_LOAD_B(dst, j)                                                                           \
  { const unsigned char* sp_ = smem + slot_of(j) * PH_SLOT;                                          \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)  \
      dst[ni][kk] = *(const i16x8*)(sp_ + b_off + ni * 2048 + (((kk * 4 + fq) ^ sw) << 4)); }
#define P10_MFMA(mh, nh, bfr)                                                                        \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int mi = 0; mi < MQ; ++mi) \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                 \
      acc[mh][mi][nh][ni] = T::mfma(bfr[ni][kk], af[mi][kk], acc[mh][mi][nh][ni]);                   \
  __builtin_amdgcn_sched_barrier(0);                                                                 \
  __builtin_amdgcn_s_barrier();                                                                      \
  __builtin_amdgcn_sched_barrier(0);
  // stream step of phase (t, p): half-tile 4 t + p + 8 = quadrant p of K-tile t + 2, into the slot of half-tile 4 t + p - 2
#define P10_STREAM(t, j, p, wait)                                                                    \
  {                                                                                                  \
    const int h = 4 * (t) + (p) + 8;                                                                 \
    if (h < H) { issue((t) + 2, (p), slot_of((j) + 8)); if (wait) PH_WAIT(12); }                     \
    else if (wait) PH_WAIT(0);                                                                       \
  }
  for (int t = 0; t < nk; t += 2) {
    // K-tile t: half-tiles base + 0 .. 3
    P10_LOAD_A(0) P10_LOAD_B(b0, 1) P10_STREAM(t, 0, 0, true) P10_MFMA(0, 0, b0)
    P10_LOAD_B(b1, 2) P10_STREAM(t, 1, 1, true) P10_MFMA(0, 1, b1)
    P10_LOAD_A(3) P10_STREAM(t, 2, 2, false) P10_MFMA(1, 1, b1)
    P10_STREAM(t, 3, 3, true) P10_MFMA(1, 0, b0)
    // K-tile t + 1: half-tiles base + 4 .. 7
    P10_LOAD_A(4) P10_LOAD_B(b0, 5) P10_STREAM(t + 1, 4, 0, true) P10_MFMA(0, 0, b0)
    P10_LOAD_B(b1, 6) P10_STREAM(t + 1, 5, 1, true) P10_MFMA(0, 1, b1)
    P10_LOAD_A(7) P10_STREAM(t + 1, 6, 2, false) P10_MFMA(1, 1, b1)
    P10_STREAM(t + 1, 7, 3, true) P10_MFMA(1, 0, b0)
    base += 8; if (base >= NS) base -= NS;
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
#undef P10_LOAD_A
#undef P10_LOAD_B
#undef P10_MFMA
#undef P10_STREAM
  const float descale = 1.0f / (a.scales->sx * a.scales->sw_cur);
  const float lo = a.relu ? 0.f : -INFINITY;
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mi = 0; mi < MQ; ++mi) {
      const int m = m0 + mh * HROWS + wm * 16 * MQ + mi * 16 + frow;
      if (m >= R) continue;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int n = n0 + nh * 128 + wn * 32 + ni * 16 + fq * 4;
          if (n >= a.D) continue;
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[mh][mi][nh][ni][j] * descale + a.bias[n + j], lo);
          *(float4*)(a.H + (int64_t)m * a.D + n) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

#endif  // VV_LAB
// ------------------------------------------------------------------------------- wgrad --------
// dW = dY^T X, one split of K per workgroup.  Both operands are k-major in HBM (dY rows / gathered feature rows), so a
// half-tile is 64 k-rows x 128 columns: 256-B LDS rows of 16 chunks, chunk' = chunk ^ (h(row) << 1) with
// h = (row & 3) | ((row >> 3) & 1) << 2, read with ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane group,
// delivered column-major = the MFMA operand layout): the 8 row segments of a 32-lane half fall on disjoint banks.
// One LDS-DMA wave-instruction = 4 k-rows.
//
// Schedule (one fragment set per operand: 176 of the 256 VGPRs go to accumulators and fragments).  X = the gathered
// feature rows (n side, 4 fragment tiles per wave and half), Y = dY (m side, 2 tiles per wave and half):
//     phase 0: X_lo x Y_lo   phase 1: X_lo x Y_hi   phase 2: X_hi x Y_hi   phase 3: X_hi x Y_lo (Y_lo read again)
// ring slots per K-tile parity: 0 X_lo, 1 Y_lo, 2 Y_hi, 3 X_hi.  The stream issues, in phase p of K-tile t,
//     p0: X_hi(t+1)   p1: Y_lo(t+1)   p2: X_lo(t+2)   p3: Y_hi(t+2)
// so the gathered operand (HBM) flies 6 phases, dY (L2 / Infinity Cache) at least 3, and ONE counted wait per K-tile
// (vmcnt(4) after p3: everything up to Y_lo(t+1) has landed, X_lo(t+2) and Y_hi(t+2) stay in flight) covers all reads
// of K-tile t+1.  Every slot is restaged at least two segments after its last read was issued.
// The table-row ids of the K range are copied to LDS first (the last 32 KiB of the 160 KiB), PH_WG_IDS rows at a time,
// so the loop has no ordinary global load (hipcc would wait vmcnt(0) for it and drain the stream); ids past the
// split's end name the table's zero row, so a padded K-tile contributes nothing.
constexpr int PH_WG_IDS = 8192;
constexpr int PH_WG_LDS_BYTES = PH_LDS_BYTES + PH_WG_IDS * 4;      // 160 KiB

__device__ __forceinline__ int ph_h(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ i16x4 ph_tr(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p));
}

// (the same read at an LDS address held as a number: nothing for the compiler to add per use but the instruction's own offset field)
__device__ __forceinline__ i16x4 ph_tr_at(unsigned lds_addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(uintptr_t)lds_addr);
}

// UPD: one split of K (S == 1) and the solver's update applied to the tile where it stands (WgradUpd, vv_internal.h) instead of the slab store
// S16 (WgradArgs::slab16): the tile leaves as f16 x one power of two per (split, tile); stores widened as in the forward kernel's O16 form
// LEAN (round 6; WgradArgs::lean): the LOAD segment of a phase is bound by the NUMBER of instructions the loading wave gets through the
// issue port beside its partner's MFMA stream (tools/lab/lds_issue_lab.hip: ~15 clocks per LDS read whatever its width; by the ISA the
// loop held 126 instructions per K-tile in its four LOAD segments, 362 clocks each, against 256 of MFMA).  The lean form issues the same
// reads and the same stream with fewer instructions around them: sources as scalar base + 32-bit lane offset (one v_mad_u32_u24 per
// gathered row instead of a 64-bit multiply and a 64-bit add; nothing per dY row), LDS addresses as scalar base + immediate into M0 (one
// instruction instead of a null-checked address-space cast), and a K loop whose body carries no end-of-stream tests (the last K-tiles
// run a checked copy) -- and FEWER reads: two Y fragment buffers of alternating roles, Y_lo read once per K-tile (the schedule: below, at
// PW_TILEL).  80 instructions per K-tile; stamps (profiles/r06_wgrad_stamps.txt): LOAD 352 -> 247 clocks per phase, below the MFMA
// segment's 290.  Needs a table below 4 GiB and row ids below 2^24 (the host decides); bit-identical results.
template <typename T, int ABL = 0, bool UPD = false, bool S16 = false, bool LEAN = false>
__global__ __launch_bounds__(GEMM_THREADS) void k_wgrad_gemm_ph(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int32_t* ids = (int32_t*)(smem + PH_LDS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tilesM = a.Dp / BM, tilesN = a.Fp / BN;
  const int L = ph_xcd_remap(blockIdx.x, gridDim.x);
  const int tmc = a.tm_count > 0 ? a.tm_count : tilesM;           // M tiles of this launch
  const int tm = a.tm_begin + L % tmc, tn = (L / tmc) % tilesN, sp = L / (tmc * tilesN);
  const int m0 = tm * BM, n0 = tn * BN;
  int total_steps = a.Rp / BK, kps = a.ksteps_per_split;
  if (a.n_dev) {                      // dedup mode: live K extent in device memory
    total_steps = (*a.n_dev + BK - 1) / BK;
    kps = (total_steps + a.S - 1) / a.S;
  }
  const int k_begin = sp * kps;
  int k_end = k_begin + kps;
  if (k_end > total_steps) k_end = total_steps;
  const int nk_all = k_end > k_begin ? k_end - k_begin : 0;

  f32x4 acc[2][4][2][2];               // [X half][n tile][Y half][m tile]
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[nh][ni][mh][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (lab, ABL bit 12: the same 128 accumulator registers as eight 32 x 32 tiles -- a timing study of the other MFMA shape, results wrong)
  constexpr bool M32T = (ABL & 4096) != 0;
  typedef float f32x16_t __attribute__((ext_vector_type(16)));
  f32x16_t acc32[2][2][2];
  if (M32T) {
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc32[x >> 2][(x >> 1) & 1][x & 1][j] = 0.f;
  }

  // staging: instruction i (0, 1) of this wave fills k-rows (i*8 + wave)*4 .. +3 of a half-tile; lane -> (row, chunk).
  // dY rows advance by a constant stride, so its two source pointers are simply incremented; the upper column half of
  // either operand is the same address + 256 B (immediate offset).  dYh / dYu carry BK rows of slack past Rp, so a
  // padded K-tile reads in bounds (its feature operand is the zero row).
  const int srow0 = wave * 4 + (lane >> 4), srow1 = srow0 + 32;
  const int scol0 = ((lane & 15) ^ (ph_h(srow0) << 1)) * 8;   // source column (halves) inside the 128-column half
  const int scol1 = ((lane & 15) ^ (ph_h(srow1) << 1)) * 8;
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, pp = li & 3;
  const uint16_t* tb0 = a.table + n0 + scol0;
  const uint16_t* tb1 = a.table + n0 + scol1;
  // (LEAN) wave-uniform bases and per-lane byte offsets
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem)) + wave * 1024;
  const uint16_t* xs = a.table + n0;
  const unsigned Fp2 = (unsigned)a.Fp * 2u, xl0 = (unsigned)scol0 * 2u, xl1 = (unsigned)scol1 * 2u;
  const unsigned yl0 = (unsigned)(srow0 * a.Dp + scol0) * 2u, yl1 = (unsigned)(srow1 * a.Dp + scol1) * 2u;
  // fragment reads: h(row1) does not depend on kk (bits 0, 1, 3 of the row), so kk, the +4-row partner and the slot are
  // immediate offsets of one address per fragment tile
  const int hx = ph_h(8 * g + q4) << 1;
  const int rd = (8 * g + q4) * 256 + (pp & 1) * 8;
  int xa[4], ya[2];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) xa[ni] = rd + (((wm * 8 + ni * 2 + (pp >> 1)) ^ hx) << 4);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) ya[mi] = rd + (((wn * 4 + mi * 2 + (pp >> 1)) ^ hx) << 4);

  // (lab, ABL bit 9: TIME STAMPS, as in k_fwd_gemm_ph -- fixed scalar registers s78..s101 that only these assembly blocks name;
  // per phase: [load segment incl. its counted wait | barrier | MFMA segment (16 MFMAs = 256 clocks of matrix pipe) | barrier])
  constexpr bool WTSON = (ABL & 512) != 0;
#define WTS_CLOB "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "s100", "s101", "scc"
#define WTS(pair) if (WTSON) { asm volatile("s_memtime " pair ::: WTS_CLOB); __builtin_amdgcn_sched_barrier(0); }
  if (WTSON) asm volatile("s_memtime s[84:85]\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 s101, s84\n\ts_mov_b32 s83, 0\n\t"
                          "s_mov_b32 s95, 0\n\ts_mov_b32 s96, 0\n\ts_mov_b32 s97, 0\n\ts_mov_b32 s98, 0\n\ts_mov_b32 s99, 0" ::: WTS_CLOB);
  for (int c0 = 0; c0 < nk_all; c0 += PH_WG_IDS / BK) {      // super-chunks of at most 128 K-tiles
    const int nk_c = nk_all - c0 < PH_WG_IDS / BK ? nk_all - c0 : PH_WG_IDS / BK;
    // (round 6: an odd count runs its last K-tile alone -- the padded, all-zero partner tile of rounds 2-5 cost one of the benchmark's 42
    // iterations per split; the row ids are still filled for an even count: the prologue stages K-tile 1 unconditionally)
    const int nk = nk_c;
    const int nk_ids = (nk_c + 1) & ~1;
    const int64_t kg0 = (int64_t)(k_begin + c0) * BK;
    const int live = nk_c * BK;
    __syncthreads();                                          // the previous chunk's stream is drained (tail waits)
    {
      // all of a thread's ids requested together, then stored (round 5: as a plain loop hipcc waited vmcnt(0) for every id before the
      // next load went out -- five to six serial L2 / HBM round trips in front of the first K-tile at the benchmark's split length)
      constexpr int NID = PH_WG_IDS / GEMM_THREADS;
      int32_t idv[NID];
      int tid_c = tid;                                        // (opaque: hipcc otherwise computes the sixteen LDS addresses of this fill in front of
      asm volatile("" : "+v"(tid_c));                         //  the chunk loop and carries -- or spills -- them through the K loop)
#pragma unroll
      for (int j = 0; j < NID; ++j) {
        const int i = tid_c + j * GEMM_THREADS;
        idv[j] = (i < live && !(ABL & 8)) ? a.rows[kg0 + i] : a.zero_row;
      }
#pragma unroll
      for (int j = 0; j < NID; ++j) {
        const int i = tid_c + j * GEMM_THREADS;
        if (i < nk_ids * BK) ids[i] = idv[j];
      }
    }
    __syncthreads();
    const uint16_t* pa0 = a.dYh + (kg0 + srow0) * a.Dp + m0 + scol0;
    const uint16_t* pa1 = a.dYh + (kg0 + srow1) * a.Dp + m0 + scol1;
    const int64_t a_step = (int64_t)BK * a.Dp;
    auto issue_y = [&](int kt, bool hi, int slot) {
      unsigned char* dst = smem + slot * PH_SLOT + wave * 1024;
      if (!hi) { ph_glds16(pa0 + kt * a_step, dst); ph_glds16(pa1 + kt * a_step, dst + 8192); }
      else { ph_glds16_hi(pa0 + kt * a_step, dst); ph_glds16_hi(pa1 + kt * a_step, dst + 8192); }
    };
    auto issue_x = [&](int id0, int id1, bool hi, int slot) {
      unsigned char* dst = smem + slot * PH_SLOT + wave * 1024;
      const int64_t r0 = (int64_t)id0 * a.Fp, r1 = (int64_t)id1 * a.Fp;
      if (!hi) { ph_glds16(tb0 + r0, dst); ph_glds16(tb1 + r1, dst + 8192); }
      else { ph_glds16_hi(tb0 + r0, dst); ph_glds16_hi(tb1 + r1, dst + 8192); }
    };
    const uint16_t* ys = a.dYh + kg0 * a.Dp + m0;             // (LEAN) dY rows of K-tile kt: ys + kt a_step, wave-uniform
    // hi, slot: literals (hi issues go to slots 2, 3, 6, 7: the LDS immediate stays positive)
#define PW_ISSUE_X(i0, i1, hi, slot)                                                                   \
    { if constexpr (LEAN) {                                                                            \
      const unsigned o0_ = __umul24((unsigned)(i0), Fp2) + xl0, o1_ = __umul24((unsigned)(i1), Fp2) + xl1; \
      if (!(hi)) { PH_GLDS_S(lds_w, o0_, xs, (slot) * PH_SLOT, 0); PH_GLDS_S(lds_w, o1_, xs, (slot) * PH_SLOT + 8192, 0); } \
      else { PH_GLDS_S(lds_w, o0_, xs, (slot) * PH_SLOT - 256, 256); PH_GLDS_S(lds_w, o1_, xs, (slot) * PH_SLOT + 8192 - 256, 256); } \
    } else issue_x(i0, i1, hi, slot); }
#define PW_ISSUE_Y(kt, hi, slot)                                                                       \
    { if constexpr (LEAN) {                                                                            \
      const uint16_t* yk_ = ys + (int64_t)(kt) * a_step;                                               \
      if (!(hi)) { PH_GLDS_S(lds_w, yl0, yk_, (slot) * PH_SLOT, 0); PH_GLDS_S(lds_w, yl1, yk_, (slot) * PH_SLOT + 8192, 0); } \
      else { PH_GLDS_S(lds_w, yl0, yk_, (slot) * PH_SLOT - 256, 256); PH_GLDS_S(lds_w, yl1, yk_, (slot) * PH_SLOT + 8192 - 256, 256); } \
    } else issue_y(kt, hi, slot); }
    // prologue, in stream order: X_lo(0), Y_hi(0), X_hi(0), Y_lo(0), X_lo(1), Y_hi(1)
    // (LEAN, whose stream runs Y_lo, X_lo, Y_hi, X_hi per K-tile: all eight half-tiles of K-tiles 0 and 1)
    int idn0, idn1;
    if constexpr (LEAN) {
      const int i00 = ids[srow0], i01 = ids[srow1];
      PW_ISSUE_Y(0, false, 1)
      PW_ISSUE_X(i00, i01, false, 0)
      PW_ISSUE_Y(0, true, 2)
      PW_ISSUE_X(i00, i01, true, 3)
      idn0 = ids[BK + srow0]; idn1 = ids[BK + srow1];
      PW_ISSUE_Y(1, false, 5)
      PW_ISSUE_X(idn0, idn1, false, 4)
      PW_ISSUE_Y(1, true, 6)
      PW_ISSUE_X(idn0, idn1, true, 7)
      PH_WAIT(8);                                             // K-tile 0 has landed (the fourth-phase wait of a K-tile "-1")
    } else {
      const int i00 = ids[srow0], i01 = ids[srow1];
      PW_ISSUE_X(i00, i01, false, 0)
      PW_ISSUE_Y(0, true, 2)
      PW_ISSUE_X(i00, i01, true, 3)
      PW_ISSUE_Y(0, false, 1)
      idn0 = ids[BK + srow0]; idn1 = ids[BK + srow1];        // ids of the next X issue: X_hi(1) in phase (0, 0)
      PW_ISSUE_X(idn0, idn1, false, 4)
      PW_ISSUE_Y(1, true, 6)
      PH_WAIT(4);
    }
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                // waves 4-7 run one segment behind
    if (wm == 1) __builtin_amdgcn_s_setprio(1);
    if (WTSON) asm volatile("s_memrealtime s[78:79]\n\ts_memtime s[84:85]\n\ts_waitcnt lgkmcnt(0)\n\t"
                            "s_mov_b32 s83, s84\n\ts_mov_b32 s88, s84\n\ts_mov_b32 s90, s84\n\ts_mov_b32 s92, s84\n\ts_mov_b32 s94, s84" ::: WTS_CLOB);

    i16x8 xf[4][2], yf[2][2];
    constexpr bool abl_st = ABL & 1, abl_mm = ABL & 2, abl_rd = ABL & 4, SF = (ABL & 1024) != 0;
    if (abl_rd) {
#pragma unroll
      for (int x = 0; x < 4; ++x) { xf[x][0] = xf[x][1] = i16x8{1, 2, 3, 4, 5, 6, 7, (short)x}; }
#pragma unroll
      for (int x = 0; x < 2; ++x) { yf[x][0] = yf[x][1] = i16x8{1, 2, 3, 4, 5, 6, 7, (short)x}; }
    }
#define PW_LOAD(dst, adr, cnt, slot)                                                                   \
    if (!abl_rd) _Pragma("unroll") for (int x = 0; x < cnt; ++x) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) { \
      const unsigned char* p_ = smem + (slot) * PH_SLOT + kk * 8192 + adr[x];                          \
      const i16x4 lo = ph_tr(p_), hi = ph_tr(p_ + 1024);                                               \
      dst[x][kk] = i16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                      \
    }
#define PW_MFMA(nh, mh) PW_MFMAY(nh, mh, yf)
#define PW_MFMAY(nh, mh, YB)                                                                           \
    WTS("s[88:89]")            /* b: this wave's load segment is done (reads issued, stream issued, wait passed) */ \
    __builtin_amdgcn_s_barrier();                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    WTS("s[90:91]")                                                                                    \
    if (M32T) {                /* (lab, ABL bit 12: TIMING ONLY -- the phase's 256 clocks of matrix pipe as 8 x 32x32x16 instead of 16 x 16x16x32, */ \
      _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)  /* the same fragment registers read; results wrong) */ \
        acc32[nh][ni >> 1][mh] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xf[ni][kk]), __builtin_bit_cast(f16x8, YB[ni & 1][kk]), acc32[nh][ni >> 1][mh], 0, 0, 0); \
    } else                                                                                             \
    if (!abl_mm) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni)  \
      _Pragma("unroll") for (int mi = 0; mi < 2; ++mi)                                                 \
        acc[nh][ni][mh][mi] = T::mfma(xf[ni][kk], YB[mi][kk], acc[nh][ni][mh][mi]);                    \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (WTSON) {                                                                                       \
      asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                          \
                   "s_sub_u32 s100, s92, s94\n\ts_add_u32 s98, s98, s100\n\t"   /* mfma += d - previous c */ \
                   "s_sub_u32 s100, s84, s92\n\ts_add_u32 s99, s99, s100\n\t"   /* bar2 += e - d */          \
                   "s_sub_u32 s100, s88, s84\n\ts_add_u32 s95, s95, s100\n\t"   /* load (incl. its wait) += b - e */ \
                   "s_sub_u32 s100, s90, s88\n\ts_add_u32 s97, s97, s100\n\t"   /* bar1 += c - b */          \
                   "s_mov_b32 s94, s90\n\ts_memtime s[92:93]" ::: WTS_CLOB);                                  \
      __builtin_amdgcn_sched_barrier(0);                                                               \
    }                                                                                                  \
    __builtin_amdgcn_s_barrier();                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    WTS("s[84:85]")
    // one K-tile of parity par (slots 4*par ..), the stream working on K-tiles t+1 and t+2
    /* (lab, ABL bit 10: the phase's two LDS-DMA instructions IN FRONT of its fragment reads instead of behind them) */ \
#define PW_TILE(par, t, CHK)                                                                           \
    if (SF && (!(CHK) || (t) + 1 < nk) && !abl_st) PW_ISSUE_X(idn0, idn1, true, 4 * (1 - (par)) + 3)   \
    PW_LOAD(xf, xa, 4, 4 * (par) + 0) PW_LOAD(yf, ya, 2, 4 * (par) + 1)                                \
    if (!SF && (!(CHK) || (t) + 1 < nk) && !abl_st) PW_ISSUE_X(idn0, idn1, true, 4 * (1 - (par)) + 3)  \
    PW_MFMA(0, 0)                                                                                      \
    if (SF && (!(CHK) || (t) + 1 < nk) && !abl_st) PW_ISSUE_Y((t) + 1, false, 4 * (1 - (par)) + 1)     \
    PW_LOAD(yf, ya, 2, 4 * (par) + 2)                                                                  \
    if (!SF && (!(CHK) || (t) + 1 < nk) && !abl_st) PW_ISSUE_Y((t) + 1, false, 4 * (1 - (par)) + 1)    \
    if (!(CHK) || (t) + 2 < nk) { idn0 = ids[((t) + 2) * BK + srow0]; idn1 = ids[((t) + 2) * BK + srow1]; } \
    PW_MFMA(0, 1)                                                                                      \
    if (SF && (!(CHK) || (t) + 2 < nk) && !abl_st) PW_ISSUE_X(idn0, idn1, false, 4 * (par) + 0)        \
    PW_LOAD(xf, xa, 4, 4 * (par) + 3)                                                                  \
    if (!SF && (!(CHK) || (t) + 2 < nk) && !abl_st) PW_ISSUE_X(idn0, idn1, false, 4 * (par) + 0)       \
    PW_MFMA(1, 1)                                                                                      \
    if (SF && (!(CHK) || (t) + 2 < nk) && !abl_st) PW_ISSUE_Y((t) + 2, true, 4 * (par) + 2)            \
    PW_LOAD(yf, ya, 2, 4 * (par) + 1)                                                                  \
    if (!SF && (!(CHK) || (t) + 2 < nk) && !abl_st) PW_ISSUE_Y((t) + 2, true, 4 * (par) + 2)           \
    if ((!(CHK) || (t) + 2 < nk) && !abl_st) { PH_WAIT(4); } else PH_WAIT(0);                          \
    PW_MFMA(1, 0)
    // The LEAN schedule.  A segment costs what its instruction count costs (~11 clocks each beside the partner's MFMAs), the MFMA segment
    // beside it 256 clocks; the four LOAD segments of a K-tile were not equal: the first read X_lo AND Y_lo (24 fragment reads), the fourth
    // read Y_lo a second time.  Now: (1) TWO Y fragment buffers whose roles alternate per K-tile -- Y_lo(t) stays in its registers from the
    // first to the fourth phase (no second read), Y_hi(t) goes to the other buffer, and the fourth LOAD segment reads Y_lo(t + 1) into that
    // one once Y_hi is spent: 16 / 8 / 16 / 8 reads, 48 per K-tile instead of 56; (2) the stream runs Y_lo, X_lo, Y_hi, X_hi per K-tile,
    // every half-tile issued at least five phases before its first read (Y_lo(t + 2) in the second phase into the slot Y_lo(t) left in
    // the phase before this K-tile began; X_hi with the Y_hi in front of it in the fourth); two counted waits per K-tile: behind the third
    // LOAD segment all but the five youngest half-tiles (Y_lo(t + 1) has landed), behind the fourth all but the four youngest (X_lo, Y_hi,
    // X_hi of K-tile t + 1); (3) the fragment addresses of the upper half of the ring (LDS offsets past the 16-bit immediate) are values
    // of their own instead of copies made per iteration.
#define PW_LOADL(dst, alo, ahi, cnt, slot)                                                             \
    _Pragma("unroll") for (int x = 0; x < cnt; ++x) _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) { \
      const unsigned p_ = ((slot) < 4 ? alo[x] : ahi[x]) + ((slot) & 3) * PH_SLOT + kk * 8192;         \
      const i16x4 lo = ph_tr_at(p_), hi = ph_tr_at(p_ + 1024);                                         \
      dst[x][kk] = i16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                      \
    }
    /* YL: the buffer that holds Y_lo(t) (read by the K-tile before, or in front of the loop); YH: the other one */ \
#define PW_TILEL(par, t, CHK, YL, YH)                                                                  \
    PW_LOADL(xf, xa0, xa4, 4, 4 * (par) + 0)                                                           \
    PW_MFMAY(0, 0, YL)                                                                                 \
    PW_LOADL(YH, ya0, ya4, 2, 4 * (par) + 2)                                                           \
    if (!(CHK) || (t) + 2 < nk) { PW_ISSUE_Y((t) + 2, false, 4 * (par) + 1)                            \
      idn0 = ids[((t) + 2) * BK + srow0]; idn1 = ids[((t) + 2) * BK + srow1]; }                        \
    PW_MFMAY(0, 1, YH)                                                                                 \
    PW_LOADL(xf, xa0, xa4, 4, 4 * (par) + 3)                                                           \
    if (!(CHK) || (t) + 2 < nk) { PW_ISSUE_X(idn0, idn1, false, 4 * (par) + 0) PH_WAIT(10); } else PH_WAIT(0); \
    PW_MFMAY(1, 1, YH)                                                                                 \
    if (!(CHK) || (t) + 1 < nk) { PW_LOADL(YH, ya0, ya4, 2, 4 * (1 - (par)) + 1) }                     \
    if (!(CHK) || (t) + 2 < nk) { PW_ISSUE_Y((t) + 2, true, 4 * (par) + 2) PW_ISSUE_X(idn0, idn1, true, 4 * (par) + 3) PH_WAIT(8); } \
    else PH_WAIT(0);                                                                                   \
    PW_MFMAY(1, 0, YL)
    int t = 0;
    if constexpr (LEAN) {
      const unsigned lds0 = (unsigned)(uintptr_t)LDS_PTR(smem);
      unsigned xa0[4], ya0[2], xa4[4], ya4[2];                // LDS addresses of the fragments in slots 0-3 / 4-7 (as numbers)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) { xa0[ni] = lds0 + xa[ni]; xa4[ni] = xa0[ni] + 4 * PH_SLOT; asm volatile("" : "+v"(xa0[ni]), "+v"(xa4[ni])); }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) { ya0[mi] = lds0 + ya[mi]; ya4[mi] = ya0[mi] + 4 * PH_SLOT; asm volatile("" : "+v"(ya0[mi]), "+v"(ya4[mi])); }
      // the body of the loop carries no end-of-stream tests: both of its K-tiles have two more behind them
      i16x8 yg[2][2];                                         // the second Y buffer
      PW_LOADL(yf, ya0, ya4, 2, 1)                            // Y_lo(0) (every wave: in front of the loop)
      for (; t + 3 < nk; t += 2) {
        PW_TILEL(0, t, 0, yf, yg)
        PW_TILEL(1, t + 1, 0, yg, yf)
      }
      for (; t + 1 < nk; t += 2) {
        PW_TILEL(0, t, 1, yf, yg)
        PW_TILEL(1, t + 1, 1, yg, yf)
      }
      if (t < nk) { PW_TILEL(0, t, 1, yf, yg) }
    } else {
      for (; t + 1 < nk; t += 2) {
        PW_TILE(0, t, 1)
        PW_TILE(1, t + 1, 1)
      }
      if (t < nk) { PW_TILE(0, t, 1) }                        // (an odd count's last K-tile: t is even, its parity is 0)
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                // waves 0-3 catch the extra barrier of waves 4-7
    if (wm == 1) __builtin_amdgcn_s_setprio(0);
    if (WTSON) asm volatile("s_memtime s[80:81]\n\ts_waitcnt lgkmcnt(0)" ::: WTS_CLOB);      // s80: end of the K loop
#undef PW_LOAD
#undef PW_MFMA
#undef PW_MFMAY
#undef PW_TILE
#undef PW_TILEL
#undef PW_LOADL
#undef PW_ISSUE_X
#undef PW_ISSUE_Y
  }

  // The epilogues' store addresses are functions of the lane (li, g) and of kernel constants: left to itself hipcc computes the sixteen
  // of them IN FRONT of the K loop and carries them through it (32 registers; with the lean loop's second Y buffer: 19-26 spilled dwords
  // per lane, 13 MB of scratch written before the loop and read after it -- the kernel ran 5 us LONGER).  li_e / g_e are the same two
  // numbers behind an opaque barrier: what depends on them is computed here, behind the loop.
  int li_e = li, g_e = g;
  asm volatile("" : "+v"(li_e), "+v"(g_e));
  if constexpr (UPD) {
    // ---- the update on the tile (k_reduce_sgd's parameter workgroups, element for element: the same products, the same rule, the same
    // roundings -- parameters, history and half copy bit for bit those of the two-launch form, tests/test_gpu_fused_update.py)
    const WgradUpd& u = a.upd;
    __syncthreads();                                      // (every wave is out of the K loop: the ring is free)
    float* red8 = (float*)smem;                           // the kernel's dynamic LDS is all 160 KiB of the CU: no static word beside it
    float sw = u.scales->sw_next;
    if (u.recompute_scale) {                              // the scale of the new half copy from the previous update's per-block maxima
      float mm = 0.f;
      for (int k = tid; k < u.wmax_prev_n; k += GEMM_THREADS) mm = fmaxf(mm, u.wmax_prev[k]);
#pragma unroll
      for (int o2 = 32; o2 > 0; o2 >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o2, 64));
      if (lane == 0) red8[wave] = mm;
      __syncthreads();
      mm = red8[0];
#pragma unroll
      for (int w8 = 1; w8 < 8; ++w8) mm = fmaxf(mm, red8[w8]);
      __syncthreads();
      const float mx = fmaxf(mm, __uint_as_float(u.scales->wmax_bits));
      sw = 1.f;
      if (u.prec == 0 && mx > 0.f && isfinite(mx)) { int e; frexpf(mx, &e); sw = ldexpf(1.f, 12 - e); }
    }
    const float sgf = u.gg ? u.sg * u.gg->mul : u.sg;
    const float inv = u.ip_scale / (sgf * u.scales->sx);
    const float lr_w = u.rate * u.lr_mult_w, dc_w = u.weight_decay * u.decay_mult_w;
    auto rule = [&](float w, float gr, float& h) {        // (k_sgd's)
      if (dc_w != 0.f) gr += dc_w * (u.reg == 2 ? w : (float)((w > 0.f) - (w < 0.f)));
      float up;
      if (u.solver_type == 1) { const float h0 = h; h = lr_w * gr + u.momentum * h0; up = (1.f + u.momentum) * h - u.momentum * h0; }
      else if (u.solver_type == 2) { h += gr * gr; up = lr_w * (gr / (sqrtf(h) + u.delta)); }
      else { h = lr_w * gr + u.momentum * h; up = h; }
      return w - up;
    };
    float wmax = 0.f;
    // The tile leaves the accumulators through LDS, 128 rows at a time (a 133 KB image, rows 1040 bytes apart: the sixteen rows a
    // ds_write_b128 touches fall on disjoint banks), and is applied to W in ROW order: a wave's instruction covers 1 KiB of one row of W --
    // whole lines, sixteen independent 16-byte loads of W and of the history in flight per lane.  (First build: the rule applied straight
    // from the accumulators' layout -- 64-byte pieces of sixteen rows per instruction, eight dependent load/store groups per lane: the
    // 300 MB of the update took 86 us inside the GEMM against 63 us for k_reduce_sgd's 370 MB; profiles/r05_shipped_update.txt.)
    constexpr int TSTR = 260;                             // floats between two rows of the LDS image
    float* tile = (float*)smem + 64;                      // (behind red8)
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
      __syncthreads();                                    // the image (and red8's readers) are done with
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            *(f32x4*)(tile + (wn * 32 + mi * 16 + li_e) * TSTR + nh * 128 + wm * 64 + ni * 16 + g_e * 4) = acc[nh][ni][mh][mi];
      __syncthreads();
      // (the sixteen row groups in an order that depends on the tile: with every workgroup of the chip walking rows 0-7, 8-15, ... of its
      // tile at the same moment the requests of a moment differ in few address bits above the row pitch, and where they fall in the memory
      // system's interleave is decided by the buffers' PHYSICAL placement: two processes in sixteen ran this epilogue at half speed,
      // profiles/r05_shipped_update.txt)
      const int rot = (tn * 5 + tm * 3 + mh * 8) & 15;
      f32x4 wv[16], hv[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int e = tid + GEMM_THREADS * ((j + rot) & 15), r = e >> 6, c = (e & 63) * 4;
        const int m = m0 + mh * 128 + r, n = n0 + c;
        const bool in = m < u.D && n < u.F;
        const int64_t o = (int64_t)m * u.F + n;
        wv[j] = in ? __builtin_nontemporal_load((const f32x4*)(u.W + o)) : f32x4{0.f, 0.f, 0.f, 0.f};
        hv[j] = in ? __builtin_nontemporal_load((const f32x4*)(u.hW + o)) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int e = tid + GEMM_THREADS * ((j + rot) & 15), r = e >> 6, c = (e & 63) * 4;
        const int m = m0 + mh * 128 + r, n = n0 + c;
        if (m >= u.D || n >= u.F) continue;
        const int64_t o = (int64_t)m * u.F + n;
        const f32x4 v = *(const f32x4*)(tile + r * TSTR + c);
        f32x4 wq = wv[j], hq = hv[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float gr = __fmul_rn(v[q], inv);          // (a rounded product, as the slab path stores it: no contraction into the rule)
          float hj = hq[q];
          wq[q] = rule(wq[q], gr, hj);
          hq[q] = hj;
          wmax = fmaxf(wmax, fabsf(wq[q]));
        }
        __builtin_nontemporal_store(wq, (f32x4*)(u.W + o));
        __builtin_nontemporal_store(hq, (f32x4*)(u.hW + o));
        const uint32_t lo = T::from_float(wq[0] * sw) | ((uint32_t)T::from_float(wq[1] * sw) << 16);
        const uint32_t hi = T::from_float(wq[2] * sw) | ((uint32_t)T::from_float(wq[3] * sw) << 16);
        *(uint2*)(u.Wh + (int64_t)m * a.Fp + n) = make_uint2(lo, hi);
      }
    }
    __syncthreads();
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o2, 64));
    if (lane == 0) red8[wave] = wmax;
    __syncthreads();
    if (tid == 0) {
      float wb = red8[0];
#pragma unroll
      for (int w8 = 1; w8 < 8; ++w8) wb = fmaxf(wb, red8[w8]);
      u.wmax_blocks[blockIdx.x] = wb;
      if (blockIdx.x == 0) { u.scales->sw_cur = sw; if (u.recompute_scale) u.scales->sw_next = sw; }
    }
    return;
  }
  if constexpr (M32T) {
    const float z = (float)(a.abl & 1);            // (0 at run time: the timing study stores zeros -- the trajectory stays that of a decaying W)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[nh][ni][mh][mi][j] = acc32[nh][ni >> 1][mh][((ni & 1) * 2 + mi) * 4 + j] * z;
  }
  if constexpr (S16) {
    // ---- f16 slabs: the tile's largest magnitude -> a power of two that puts it in [2^14, 2^15) -> halves, eight consecutive columns per lane
    float mx = 0.f;
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(acc[nh][ni][mh][mi][j]));
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
    __syncthreads();                                      // (every wave is out of the K loop: the ring is free)
    float* red8 = (float*)smem;
    if (lane == 0) red8[wave] = mx;
    __syncthreads();
    mx = red8[0];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) mx = fmaxf(mx, red8[w8]);
    float sc = 1.f;
    if (mx > 0.f && mx < 3.0e38f) { int e; (void)frexpf(mx, &e); sc = ldexpf(1.f, 15 - e); }      // mx sc in [2^14, 2^15); inf / nan: passed through at scale 1
    if (tid == 0) a.slab_sc[(int64_t)sp * (tilesM * tilesN) + tm * tilesN + tn] = 1.f / sc;        // (a power of two: exact)
    uint16_t* slab = (uint16_t*)a.slabs + (int64_t)sp * slab_pitch(a.Dp, a.Fp);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + mh * 128 + wn * 32 + mi * 16 + li_e;
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            uint32_t p[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const f32x4 v = acc[nh][2 * pr + q][mh][mi];
              const _Float16 h0 = (_Float16)(v[0] * sc), h1 = (_Float16)(v[1] * sc), h2 = (_Float16)(v[2] * sc), h3 = (_Float16)(v[3] * sc);
              p[q][0] = (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
              p[q][1] = (uint32_t)__builtin_bit_cast(uint16_t, h2) | ((uint32_t)__builtin_bit_cast(uint16_t, h3) << 16);
            }
            // (rows of 16 lanes, as k_fwd_gemm_ph's O16 epilogue: the lane of row g then holds columns (g & 1) 16 + (g >> 1) 8 .. + 7 of the 32-column pair)
            const auto s0 = __builtin_amdgcn_permlane16_swap(p[0][0], p[1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(p[0][1], p[1][1], false, false);
            const int n = n0 + nh * 128 + wm * 64 + pr * 32 + (g_e & 1) * 16 + (g_e >> 1) * 8;
            *(uint4*)(slab + (int64_t)m * a.Fp + n) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
      }
    return;
  }
  // D' = X_frag^T-major: the lane's column is m (dY column = output row d), its 4 registers 4 consecutive n
  float* slab = a.slabs + (int64_t)sp * slab_pitch(a.Dp, a.Fp);
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int m = m0 + mh * 128 + wn * 32 + mi * 16 + li_e;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int n = n0 + nh * 128 + wm * 64 + ni * 16 + g_e * 4;
          const f32x4 v = acc[nh][ni][mh][mi];
          // (plain stores: as non-temporal ones the 67 MB go straight to HBM and the kernel takes 8 us longer, profiles/r03_step_ablations.txt 5d)
          if (!(ABL & 64) || v[0] == 12345.f) *(float4*)(slab + (int64_t)m * a.Fp + n) = make_float4(v[0], v[1], v[2], v[3]);   // (lab, ABL 64: no stores)
        }
    }
  if (WTSON) {
    uint32_t o_[10];
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime s[84:85]\n\ts_memrealtime s[86:87]\n\ts_waitcnt lgkmcnt(0)\n\t"
                 "s_mov_b32 %0, s95\n\ts_mov_b32 %1, s96\n\ts_mov_b32 %2, s97\n\ts_mov_b32 %3, s98\n\ts_mov_b32 %4, s99\n\t"
                 "s_sub_u32 %5, s83, s101\n\ts_sub_u32 %6, s80, s101\n\ts_sub_u32 %7, s84, s101\n\ts_sub_u32 %8, s86, s78\n\ts_sub_u32 %9, s84, s83"
                 : "=s"(o_[0]), "=s"(o_[1]), "=s"(o_[2]), "=s"(o_[3]), "=s"(o_[4]), "=s"(o_[5]), "=s"(o_[6]), "=s"(o_[7]), "=s"(o_[8]), "=s"(o_[9])
                 :: WTS_CLOB, "memory");
    if (lane == 0) {
      uint32_t* o = (uint32_t*)(a.slabs + (int64_t)a.S * slab_pitch(a.Dp, a.Fp)) + ((size_t)blockIdx.x * 8 + wave) * 12;      // (lab: behind the last slab -- the harness allocates the room)
#pragma unroll
      for (int j = 0; j < 10; ++j) o[j] = o_[j];
      o[10] = (uint32_t)nk_all; o[11] = 0;
    }
  }
#undef WTS
#undef WTS_CLOB
}

#ifdef VV_LAB
// ------------------------------------------------------------------------------- weight gradient, four waves ----
// k_wgrad_gemm_w4: the same product and the same 256 x 256 tile as k_wgrad_gemm_ph with FOUR waves, one per SIMD, each
// owning a 128 x 128 quadrant (64 accumulator tiles = 256 registers; the fragments live in the other half of the
// 512-register file a lone wave may use).  Why: the eight-wave kernel reads 28 operand fragments per wave and K-tile
// (224 KiB of LDS reads per CU and K-tile) for 64 MFMAs each; its ablation (profiles/r02_ph_gemm_ablation.txt) shows
// the LDS-DMA writes and the fragment reads serialising on the LDS (staging alone 0.145 ms + reads alone 0.111 ms =
// 0.237 ms with the MFMAs removed, 0.26 ms with them).  A 128 x 128 wave tile needs 16 fragments per 32-deep step for
// 64 MFMAs: 128 KiB of reads per K-tile.
//   * stage = one 32-deep K step: four sub-slots of 8 KiB (X_lo, X_hi, Y_lo, Y_hi: 32 k-rows x 128 columns, the image
//     of k_wgrad_gemm_ph's half-tiles) in a ring of 4 stages (128 KiB); wave (wx, wy) reads X half wx and Y half wy;
//   * every wave stages 2 of the 8 1-KiB pieces of each sub-slot; stage s+4 is issued right after the barrier that
//     opens stage s (its sub-slots are the ones stage s's fragments were read from, now in registers everywhere), so three
//     stages (96 KiB per CU) are in flight across every barrier; one counted wait (vmcnt(16)) and ONE barrier per stage;
//   * fragments are double-buffered in registers: the reads of stage s+1 are issued between the MFMAs of stage s.
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_wgrad_gemm_w4(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SUB = 8192;                              // one sub-slot: 32 k-rows x 128 halves
  int32_t* ids = (int32_t*)(smem + PH_LDS_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave & 1, wy = wave >> 1;
  const int tilesM = a.Dp / BM, tilesN = a.Fp / BN;
  const int L = ph_xcd_remap(blockIdx.x, gridDim.x);
  const int tmc = a.tm_count > 0 ? a.tm_count : tilesM;
  const int tm = a.tm_begin + L % tmc, tn = (L / tmc) % tilesN, sp = L / (tmc * tilesN);
  const int m0 = tm * BM, n0 = tn * BN;
  int total_steps = a.Rp / BK, kps = a.ksteps_per_split;
  if (a.n_dev) {
    total_steps = (*a.n_dev + BK - 1) / BK;
    kps = (total_steps + a.S - 1) / a.S;
  }
  const int k_begin = sp * kps;
  int k_end = k_begin + kps;
  if (k_end > total_steps) k_end = total_steps;
  const int nk_all = k_end > k_begin ? k_end - k_begin : 0;

  f32x4 acc[8][8];                     // [X tile][Y tile]
#pragma unroll
  for (int ni = 0; ni < 8; ++ni)
#pragma unroll
    for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: piece p = i*4 + wave (i = 0, 1) of a sub-slot = k-rows 4p .. 4p+3; lane -> (row, 16-byte chunk)
  const int srow0 = wave * 4 + (lane >> 4), srow1 = srow0 + 16;
  const int scol0 = ((lane & 15) ^ (ph_h(srow0) << 1)) * 8;
  const int scol1 = ((lane & 15) ^ (ph_h(srow1) << 1)) * 8;
  const uint16_t* tb0 = a.table + n0 + scol0;
  const uint16_t* tb1 = a.table + n0 + scol1;
  const int g = lane >> 4, li = lane & 15, q4 = li >> 2, pp = li & 3;
  const int hx = ph_h(8 * g + q4) << 1;
  const int rd = (8 * g + q4) * 256 + (pp & 1) * 8;
  int fa[8];                           // fragment tile ti of a sub-slot (both operands share the image)
#pragma unroll
  for (int ti = 0; ti < 8; ++ti) fa[ti] = rd + (((ti * 2 + (pp >> 1)) ^ hx) << 4);
  const unsigned char* xbase = smem + wx * SUB;
  const unsigned char* ybase = smem + (2 + wy) * SUB;

  for (int c0 = 0; c0 < nk_all; c0 += PH_WG_IDS / BK) {
    const int nk_c = nk_all - c0 < PH_WG_IDS / BK ? nk_all - c0 : PH_WG_IDS / BK;
    const int nk = (nk_c + 1) & ~1;                           // even number of K-tiles: the stage count is a multiple of 4
    const int ns = nk * 2;
    const int64_t kg0 = (int64_t)(k_begin + c0) * BK;
    const int live = nk_c * BK;
    __syncthreads();
    for (int i = tid; i < nk * BK; i += 256) ids[i] = i < live ? a.rows[kg0 + i] : a.zero_row;
    __syncthreads();
    const uint16_t* pa0 = a.dYh + (kg0 + srow0) * a.Dp + m0 + scol0;
    const uint16_t* pa1 = a.dYh + (kg0 + srow1) * a.Dp + m0 + scol1;
    const int64_t a_step = (int64_t)32 * a.Dp;
    // stage st into ring position q: X_lo, X_hi, Y_lo, Y_hi (2 pieces each)
    auto issue = [&](int st, int q, int id0, int id1) {
      unsigned char* dst = smem + q * 4 * SUB + wave * 1024;
      const int64_t r0 = (int64_t)id0 * a.Fp, r1 = (int64_t)id1 * a.Fp;
      ph_glds16(tb0 + r0, dst);                   ph_glds16(tb1 + r1, dst + 4096);
      ph_glds16_hi(tb0 + r0, dst + SUB);          ph_glds16_hi(tb1 + r1, dst + SUB + 4096);
      const uint16_t* y0 = pa0 + st * a_step; const uint16_t* y1 = pa1 + st * a_step;
      ph_glds16(y0, dst + 2 * SUB);               ph_glds16(y1, dst + 2 * SUB + 4096);
      ph_glds16_hi(y0, dst + 3 * SUB);            ph_glds16_hi(y1, dst + 3 * SUB + 4096);
    };
    // X fragments are single-buffered (tile ni is dead after its eight MFMAs and is reloaded for the next stage right
    // behind them), Y fragments double-buffered: 32 + 64 registers instead of 128
    i16x8 xf[8], yf[2][8];
#define W4_FRAG(dst, base, q, ti)                                                                      \
    { const unsigned char* p_ = (base) + (q) * 4 * SUB + fa[ti];                                       \
      const i16x4 lo_ = ph_tr(p_), hi_ = ph_tr(p_ + 1024);                                             \
      dst = i16x8{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]}; }
#define W4_RELOAD(q, ti) W4_FRAG(xf[ti], xbase, ((q) + 1) & 3, ti) W4_FRAG(yf[((q) + 1) & 1][ti], ybase, ((q) + 1) & 3, ti)
#define W4_GROUP(q, ni, nxt)                                                                           \
    if ((ni) > 0 && (nxt)) { W4_RELOAD(q, (ni) > 0 ? (ni) - 1 : 0) }                                   \
    _Pragma("unroll") for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = T::mfma(xf[ni], yf[(q) & 1][mi], acc[ni][mi]); \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                 \
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                               \
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                               \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);
    // stage s (ring position q, Y fragments in buffer q & 1): wait for stage s+1, open with the barrier, restage the
    // ring position with stage s+4, reload the fragments for stage s+1 between the MFMAs of stage s
#define W4_STAGE(q, s, WAITN, more)                                                                    \
    PH_WAIT(WAITN);                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                 \
    __builtin_amdgcn_s_barrier();                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    if (more) {                                                                                        \
      issue((s) + 4, q, idn0, idn1);                                                                   \
      if ((s) + 5 < ns) { idn0 = ids[((s) + 5) * 32 + srow0]; idn1 = ids[((s) + 5) * 32 + srow1]; }    \
    }                                                                                                  \
    { const bool nxt_ = (s) + 1 < ns;                                                                  \
      W4_GROUP(q, 0, nxt_) W4_GROUP(q, 1, nxt_) W4_GROUP(q, 2, nxt_) W4_GROUP(q, 3, nxt_)              \
      W4_GROUP(q, 4, nxt_) W4_GROUP(q, 5, nxt_) W4_GROUP(q, 6, nxt_) W4_GROUP(q, 7, nxt_)              \
      if (nxt_) { W4_RELOAD(q, 7) } }                                                                  \
    __builtin_amdgcn_sched_barrier(0);
    // prologue: stages 0..3 in flight, stage 0's fragments in buffer 0
#pragma unroll
    for (int st = 0; st < 4; ++st) issue(st, st, ids[st * 32 + srow0], ids[st * 32 + srow1]);
    int idn0 = ns > 4 ? ids[4 * 32 + srow0] : a.zero_row, idn1 = ns > 4 ? ids[4 * 32 + srow1] : a.zero_row;
    PH_WAIT(24);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int ti = 0; ti < 8; ++ti) { W4_FRAG(xf[ti], xbase, 0, ti) W4_FRAG(yf[0][ti], ybase, 0, ti) }
    int s = 0;
    for (; s + 4 < ns; s += 4) {
      W4_STAGE(0, s, 16, true)
      W4_STAGE(1, s + 1, 16, true)
      W4_STAGE(2, s + 2, 16, true)
      W4_STAGE(3, s + 3, 16, true)
    }
    W4_STAGE(0, s, 16, false)
    W4_STAGE(1, s + 1, 8, false)
    W4_STAGE(2, s + 2, 0, false)
    W4_STAGE(3, s + 3, 0, false)
#undef W4_FRAG
#undef W4_RELOAD
#undef W4_GROUP
#undef W4_STAGE
  }

  float* slab = a.slabs + (int64_t)sp * slab_pitch(a.Dp, a.Fp);
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wy * 128 + mi * 16 + li;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      const int n = n0 + wx * 128 + ni * 16 + g * 4;
      const f32x4 v = acc[ni][mi];
      *(float4*)(slab + (int64_t)m * a.Fp + n) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <typename T>
static void launch_wgrad_w4_t(const WgradArgs& a, hipStream_t s) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_wgrad_gemm_w4<T>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, PH_WG_LDS_BYTES), true);
  (void)once;
  const dim3 grid((a.tm_count > 0 ? a.tm_count : a.Dp / BM) * (a.Fp / BN) * a.S), block(256);
  VV_LAUNCH((k_wgrad_gemm_w4<T>), grid, block, PH_WG_LDS_BYTES, s, a);
}
void launch_wgrad_gemm_w4(int prec, const WgradArgs& a, hipStream_t s) {
  if (prec == 0) launch_wgrad_w4_t<F16>(a, s); else launch_wgrad_w4_t<BF16>(a, s);
}
#endif  // VV_LAB

// ------------------------------------------------------------------------------- launchers ----
// (KernelOpts::fwd_lead = 0: sibling workgroups ask for their gathered rows at the same moment again)

template <typename T, int DROP, bool VEC, int MQ, int DEAD = 0>
static void launch_fwd_ph_q(const FwdArgs& a, hipStream_t s, long tiles_est) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES), true);
  (void)once;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  constexpr int BMT = 64 * MQ - 16 * DEAD;
  const dim3 grid(((a.R + BMT - 1) / BMT) * (Dp / BN)), block(GEMM_THREADS);
  (void)tiles_est;
  if constexpr (!DROP && VEC && DEAD == 0) {
    if (a.h16) {                          // ip2 as f16 (FwdArgs::h16): the same three forms -- sibling lead, gated, plain -- with the narrow epilogue
      constexpr int LDS10 = 10 * PH_SLOT;
      if constexpr (MQ <= 3) {
        if (ko().fwd_lead && Dp / BN > 1 && !a.gate) {
          static bool once_l16 = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 1, 0, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS10), true);
          (void)once_l16;
          VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 1, 0, true>), grid, block, LDS10, s, a);
          return;
        }
      }
      if (a.gate) {
        static bool once_g16 = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, true, DEAD, 0, 0, true>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES), true);
        (void)once_g16;
        VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, true, DEAD, 0, 0, true>), grid, block, PH_LDS_BYTES, s, a);
        return;
      }
      static bool once_p16 = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 0, 0, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES), true);
      (void)once_p16;
      VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 0, 0, true>), grid, block, PH_LDS_BYTES, s, a);
      return;
    }
  }
  if constexpr (!DROP && VEC && DEAD == 0 && MQ <= 3) {
    // the sibling lead: 78.6-79.7 against 81.7-82.9 us at the benchmark's de-duplicated size (192-row tiles, one round), 215
    // against 218 us for 192-row tiles in three rounds.  Not for 256-row tiles: that instantiation has no registers left for
    // it (256 + 48 bytes of scratch: dense 229 against 183 us, cfg 5 560 against 470 us)
    if (ko().fwd_lead && Dp / BN > 1 && !a.gate) {        // (the gated kernel keeps its static LDS word: no room beside ten slots)
      constexpr int LDS10 = 10 * PH_SLOT;
      if (ko().fwd_merge) {
        static bool once_m = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 1, 1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS10), true);
        (void)once_m;
        VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 1, 1>), grid, block, LDS10, s, a);
        return;
      }
      static bool once_l = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 1>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS10), true);
      (void)once_l;
      VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 1>), grid, block, LDS10, s, a);
      return;
    }
  }
  if constexpr (!DROP && VEC) {
    if (a.gate) {
      static bool once_g = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, true, DEAD>,
                            hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES), true);
      (void)once_g;
      VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, true, DEAD>), grid, block, PH_LDS_BYTES, s, a);
      return;
    }
  }
  if constexpr (!DROP && VEC && DEAD == 0) {      // (also measured with dropout on the shipped 128-row shape: 68-69 against 64-65 us, and on cfg 5: 465 against 457 us)
    if (ko().fwd_merge) {
      static bool once_m0 = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 0, 1>,
                             hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES), true);
      (void)once_m0;
      VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD, 0, 1>), grid, block, PH_LDS_BYTES, s, a);
      return;
    }
  }
  VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, MQ, 0, false, DEAD>), grid, block, PH_LDS_BYTES, s, a);
}
// the gated instantiation exists for the plain forward (no dropout, D % 4 == 0): the caller gates only then
bool fwd_gemm_can_gate(const FwdArgs& a) { return a.drop_ratio == 0.f && a.D % 4 == 0; }

// (lab: KernelOpts::fwd_ring10 selects the ten-slot forward kernel, KernelOpts::ph_mq forces the tile)

// Tile of the forward GEMM for R rows (R_hint > 0: the distinct-row count of the previous step, the rows the workgroups
// will really find) and the number of workgroups that get a tile: the least (rounds of 256 workgroups) x (cost of one
// K-tile of that height); the cost of a K-tile is not proportional to the tile height: the B half-tiles, the barriers and
// the phase structure do not shrink with it.  Tiles: 256, 192 and 128 rows; 176 (192 without its last 16-row MFMA tile:
// 236 instead of 216 workgroups on the 256 CUs at the benchmark's ~20 650 distinct rows) only on request (VV_PH_MQ=31):
// measured on one box, 400 steps each, the 176-row launch took 85.9 us against 84.4 us -- every workgroup of the single
// round still stages and waits for the same half-tiles, so fewer rows per workgroup shorten nothing, and the 20 extra
// workgroups take the CUs the grouping kernels of the next step were running on (profiles/r03_step_ablations.txt).
static const int kTileRows[4] = {256, 192, 176, 128};
static const int kTileCost[4] = {100, 85, 1000000, 70};
static int fwd_pick_tile(int R, int R_hint, int D, long* tiles_out) {
  const int Dp = (int)round_up(D, D_ALIGN);
  const int Rh = R_hint > 0 ? (int)std::min<long>(R, R_hint + R_hint / 32 + 64) : R;
  int best = 0; long best_cost = -1;
  for (int t = 0; t < 4; ++t) {
    const long tiles = ((Rh + kTileRows[t] - 1) / kTileRows[t]) * (long)(Dp / BN);
    const long cost = ((tiles + 255) / 256) * kTileCost[t];
    if (best_cost < 0 || cost < best_cost) { best = t; best_cost = cost; }
  }
  { const int g_ph_mq = ko().ph_mq; if (g_ph_mq == 4) best = 0; else if (g_ph_mq == 3) best = 1; else if (g_ph_mq == 31) best = 2; else if (g_ph_mq == 2) best = 3; }
  if (tiles_out) *tiles_out = ((Rh + kTileRows[best] - 1) / kTileRows[best]) * (long)(Dp / BN);
  return best;
}
long fwd_gemm_plan(int R, int R_hint, int D, int* mq_out) {
  long tiles = 0;
  const int t = fwd_pick_tile(R, R_hint, D, &tiles);
  if (mq_out) *mq_out = t == 0 ? 4 : t == 3 ? 2 : 3;
  return tiles;
}

template <typename T, int DROP, bool VEC>
static void launch_fwd_ph_t(const FwdArgs& a, hipStream_t s) {
  const int Dp = (int)round_up(a.D, D_ALIGN);
  (void)Dp;
  const int best = fwd_pick_tile(a.R, a.n_dev ? a.R_hint : 0, a.D, nullptr);
#ifdef VV_LAB
  if constexpr (T::id == 0 && !DROP && VEC) {
    if (a.abl) {
      const dim3 grid(((a.R + 255) / 256) * (Dp / BN)), block(GEMM_THREADS);
#define VV_ABL_FWP(N)                                                                                  \
      if (a.abl == N) {                                                                                \
        (void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, 4, N>, hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES); \
        VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, 4, N>), grid, block, PH_LDS_BYTES, s, a);                \
        return;                                                                                        \
      }
      VV_ABL_FWP(1) VV_ABL_FWP(2) VV_ABL_FWP(3) VV_ABL_FWP(4) VV_ABL_FWP(6) VV_ABL_FWP(7) VV_ABL_FWP(8) VV_ABL_FWP(9) VV_ABL_FWP(14)
#undef VV_ABL_FWP
    }
  }
  if constexpr (T::id == 0 && !DROP && VEC) {
    // lab: ablations of the 192-row kernel at the de-duplicated size (VV_LAB_FWD_ABL; the step's results are wrong)
    const int lab_abl = ko().lab_fwd_abl;
    if (lab_abl && best == 1 && !a.gate) {
      const dim3 grid(((a.R + 191) / 192) * (Dp / BN)), block(GEMM_THREADS);
#define VV_LAB_FWP(N)                                                                                  \
      if (lab_abl == N) {                                                                              \
        if (ko().fwd_lead) {                                                                           \
          (void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, 3, N, false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PH_SLOT); \
          VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, 3, N, false, 0, 1>), grid, block, 10 * PH_SLOT, s, a);  \
        } else {                                                                                       \
          (void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, 3, N>, hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS_BYTES); \
          VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, 3, N>), grid, block, PH_LDS_BYTES, s, a);              \
        }                                                                                              \
        return;                                                                                        \
      }
      VV_LAB_FWP(1) VV_LAB_FWP(2) VV_LAB_FWP(3) VV_LAB_FWP(6) VV_LAB_FWP(8) VV_LAB_FWP(14) VV_LAB_FWP(64) VV_LAB_FWP(67) VV_LAB_FWP(128) VV_LAB_FWP(256)
#undef VV_LAB_FWP
      if (lab_abl == 8192 && ko().fwd_lead) {   // the K-cut ceiling (k_fwd_gemm_ph, ABL bit 13), in the output form the step runs
        long tiles_k = 0;
        (void)fwd_pick_tile(a.R, a.n_dev ? a.R_hint : 0, a.D, &tiles_k);
        const dim3 grid(tiles_k + 40);          // the product launch's grid + the 40 helpers
        if (a.h16) {
          (void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, 3, 8192, false, 0, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PH_SLOT);
          VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, 3, 8192, false, 0, 1, 0, true>), grid, block, 10 * PH_SLOT, s, a);
        } else {
          (void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<T, DROP, VEC, 3, 8192, false, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PH_SLOT);
          VV_LAUNCH((k_fwd_gemm_ph<T, DROP, VEC, 3, 8192, false, 0, 1>), grid, block, 10 * PH_SLOT, s, a);
        }
        return;
      }
    }
  }
  if constexpr (!DROP && VEC) {
    if (ko().fwd_ring10 && !a.gate && a.D % 4 == 0 && a.bias && (best == 0 || best == 1)) {
      const dim3 block(GEMM_THREADS);
      if (best == 0) {
        static bool o4 = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph10<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PH_SLOT), true); (void)o4;
        VV_LAUNCH((k_fwd_gemm_ph10<T, 4>), dim3(((a.R + 255) / 256) * (Dp / BN)), block, 10 * PH_SLOT, s, a);
      } else {
        static bool o3 = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph10<T, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 10 * PH_SLOT), true); (void)o3;
        VV_LAUNCH((k_fwd_gemm_ph10<T, 3>), dim3(((a.R + 191) / 192) * (Dp / BN)), block, 10 * PH_SLOT, s, a);
      }
      return;
    }
  }
#endif
  long tiles_est = 0;
  (void)fwd_pick_tile(a.R, a.n_dev ? a.R_hint : 0, a.D, &tiles_est);
  if (best == 0) launch_fwd_ph_q<T, DROP, VEC, 4>(a, s, tiles_est);
  else if (best == 1) launch_fwd_ph_q<T, DROP, VEC, 3>(a, s, tiles_est);
#ifdef VV_LAB
  else if (best == 2) launch_fwd_ph_q<T, DROP, VEC, 3, 1>(a, s, tiles_est);      // (176-row tile: only on request, KernelOpts::ph_mq = 31)
#endif
  else launch_fwd_ph_q<T, DROP, VEC, 2>(a, s, tiles_est);
}

template <typename T>
static void launch_fwd_ph_p(const FwdArgs& a, hipStream_t s) {
  const bool drop = a.drop_ratio > 0.f, vec = a.D % 4 == 0;
  if (drop && a.mask) { if (vec) launch_fwd_ph_t<T, 2, true>(a, s); else launch_fwd_ph_t<T, 2, false>(a, s); }
  else if (drop) { if (vec) launch_fwd_ph_t<T, 1, true>(a, s); else launch_fwd_ph_t<T, 1, false>(a, s); }
  else { if (vec) launch_fwd_ph_t<T, 0, true>(a, s); else launch_fwd_ph_t<T, 0, false>(a, s); }
}

template <typename T>
static void launch_wgrad_ph_t(const WgradArgs& a, hipStream_t s) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_wgrad_gemm_ph<T>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, PH_WG_LDS_BYTES), true);
  (void)once;
  const dim3 grid((a.tm_count > 0 ? a.tm_count : a.Dp / BM) * (a.Fp / BN) * a.S), block(GEMM_THREADS);
#ifdef VV_LAB
  if constexpr (T::id == 0) {
    if (a.abl) {
#define VV_ABL_WGP(N)                                                                                  \
      if (a.abl == N) {                                                                                \
        (void)hipFuncSetAttribute((const void*)k_wgrad_gemm_ph<T, N>, hipFuncAttributeMaxDynamicSharedMemorySize, PH_WG_LDS_BYTES); \
        VV_LAUNCH((k_wgrad_gemm_ph<T, N>), grid, block, PH_WG_LDS_BYTES, s, a);                         \
        return;                                                                                        \
      }
      VV_ABL_WGP(1) VV_ABL_WGP(2) VV_ABL_WGP(3) VV_ABL_WGP(4) VV_ABL_WGP(6) VV_ABL_WGP(7) VV_ABL_WGP(8) VV_ABL_WGP(9) VV_ABL_WGP(64) VV_ABL_WGP(71) VV_ABL_WGP(4096)
#undef VV_ABL_WGP
    }
  }
#endif
#define VV_WG_LAUNCH(...)                                                                              \
  {                                                                                                    \
    static bool once_ = ((void)hipFuncSetAttribute((const void*)k_wgrad_gemm_ph<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, PH_WG_LDS_BYTES), true); \
    (void)once_;                                                                                       \
    VV_LAUNCH((k_wgrad_gemm_ph<__VA_ARGS__>), grid, block, PH_WG_LDS_BYTES, s, a);                     \
    return;                                                                                            \
  }
  // (WgradArgs::lean: the table lies below 4 GiB and its row ids below 2^24 -- api.hip decides: the lean instantiations)
  if (a.slab16 && !a.fuse_upd) { if (a.lean) VV_WG_LAUNCH(T, 0, false, true, true) else VV_WG_LAUNCH(T, 0, false, true) }
  if (a.fuse_upd) {                       // (one split of K: the update where the gradient is born -- api.hip decides, WgradUpd)
    if (a.lean) VV_WG_LAUNCH(T, 0, true, false, true) else VV_WG_LAUNCH(T, 0, true)
  }
  if (a.lean) VV_WG_LAUNCH(T, 0, false, false, true)
  VV_LAUNCH((k_wgrad_gemm_ph<T>), grid, block, PH_WG_LDS_BYTES, s, a);
#undef VV_WG_LAUNCH
}

void launch_wgrad_gemm_ph(int prec, const WgradArgs& a, hipStream_t s) {
  if (prec == 0) launch_wgrad_ph_t<F16>(a, s); else launch_wgrad_ph_t<BF16>(a, s);
}

// Box calibration (probe.hip: vv_box_probe).  The benchmark's own forward instantiation -- f16 operands, 192-row tiles, sibling lead, no
// dropout -- on the rows the caller hands it (contiguous rows of a random table: FwdArgs::n_dev null, the grid covers R).  marks: the same
// kernel with its four time marks (ABL bit 11: kernel start, loop start, loop end, last store + the 100 MHz real-time counter; every wave
// leaves uint32[12] at FwdArgs::mask): shader clocks over the K loop / real time over the K loop = the clock the chip held under the GEMM.
void launch_fwd_probe(const FwdArgs& a, hipStream_t s, bool marks) {
  constexpr int LDS10 = 10 * PH_SLOT;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const dim3 grid(((a.R + 191) / 192) * (Dp / BN)), block(GEMM_THREADS);
  if (marks) {
    static bool once_m = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<F16, 0, true, 3, 2048, false, 0, 1>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS10), true);
    (void)once_m;
    hipLaunchKernelGGL((k_fwd_gemm_ph<F16, 0, true, 3, 2048, false, 0, 1>), grid, block, LDS10, s, a);
  } else {
    static bool once_p = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<F16, 0, true, 3, 0, false, 0, 1>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS10), true);
    (void)once_p;
    hipLaunchKernelGGL((k_fwd_gemm_ph<F16, 0, true, 3, 0, false, 0, 1>), grid, block, LDS10, s, a);
  }
}

void launch_fwd_gemm_ph(int prec, const FwdArgs& a, hipStream_t s) {
  if (prec == 0) launch_fwd_ph_p<F16>(a, s); else launch_fwd_ph_p<BF16>(a, s);
}

}  // namespace vv
