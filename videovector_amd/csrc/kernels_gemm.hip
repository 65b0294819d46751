// kernels_gemm.hip -- the round-1 forms of the two MFMA kernels of the videovec training step (gfx950 only) and the
// launch dispatch.  The default kernels are the phase-staggered ones of kernels_gemm_ph.hip (VV_GEMM_VARIANT=5); these
// are kept as the measured baseline (variant 0: forward 0.111 ms, weight gradient 0.109 ms at the de-duplicated size
// against 0.081 / 0.071) and for the staging-transpose form of the weight gradient (VV_WGRAD_TR=0).
//
//   k_fwd_gemm   : ip2 = ReLU(X W^T + b) with X gathered row-by-row from the HBM-resident feature
//                  table through the triplet index.  Replaces the data layer's batch copy +
//                  SLICE/CONCAT transpose + InnerProductLayer::Forward + ReLU (+Dropout)
//                  (reference: video_sampled_shots_data_layer.cpp:439-452,856-875,
//                  slice_layer.cu:22-33, concat_layer.cu:21-32, inner_product_layer.cu:12-27,
//                  relu_layer.cu:10-27, dropout_layer.cu:15-41).
//   k_wgrad_gemm : dW = dY^T X (split-K partial slabs), X gathered again through the same index.
//                  Replaces InnerProductLayer::Backward's weight gradient
//                  (inner_product_layer.cu:36-42).
//
// Both: 256x256 output tile per 512-thread workgroup, K advanced 64 at a time, operand tiles
// brought HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B per lane; the gather is simply the
// per-lane source address), double buffered; 8 waves x (8x4) MFMA 16x16x32 accumulators.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "vv_internal.h"

namespace vv {

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  // LDS destination = wave-uniform base + lane * 16 (hardware rule); source is per lane.
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  // Blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous run of
  // logical tiles so tiles that share operand rows also share an L2.  Speed only.
  const int x = bid & 7, q = nblk >> 3, rem = nblk & 7;
  return x * q + (x < rem ? x : rem) + (bid >> 3);
}

// Ablation switches for timing studies only (VV_ABLATE; results are wrong when set):
//   1 = no LDS-DMA staging inside the K loop, 2 = no MFMA (fragments kept alive), 4 = no LDS reads
template <typename T, int ABL>
__device__ __forceinline__ f32x4 mfma_abl(i16x8 x, i16x8 y, f32x4 c) {
  if constexpr (ABL & 2) { asm volatile("" ::"v"(x), "v"(y)); return c; }
  else return T::mfma(x, y, c);
}

// ------------------------------------------------------------------------------- forward ------
// LDS operand image: [256 rows][64 halves] = 128-B rows of 8 16-B chunks, chunk' = chunk ^ (row&7)
// (conflict-free for the ds_read_b128 fragment reads: tools/lds_banks.py).
// MI = 16-row MFMA sub-tiles per wave along M: the tile is (32*MI) x 256.  MI = 8 is the square
// 256x256 tile; the launcher picks a smaller MI when that fills the 256 CUs in fewer, fuller rounds
// (56 320 rows: 220 tiles of 256 rows x 2 = 440 WGs = 2 rounds at 86 %, but 252 tiles of 224 rows
// x 2 = 504 WGs = 2 rounds of tiles that are 12.5 % shorter).
template <typename T, bool DROP, bool VEC, int ABL = 0, int MI = 8, int SCHED = 0>
__global__ __launch_bounds__(GEMM_THREADS) void k_fwd_gemm(FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BMK = 32 * MI;                   // rows of this tile
  constexpr int NA = (BMK * 8 + 511) / 512;      // A staging instructions per thread and K-step
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const int tilesN = Dp / BN;
  // dedup mode: the grid covers the worst case, the live row count sits in device memory
  const int R = a.n_dev ? *a.n_dev : a.R;
  const int nact = a.n_dev ? ((R + BMK - 1) / BMK) * tilesN : (int)gridDim.x;
  // the kernels that read this step's index batch ran before this one (same stream): tell the host its staging slot is free
  if (a.seq_host && blockIdx.x == 0 && tid == 0) __hip_atomic_store(a.seq_host, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if ((int)blockIdx.x >= nact) return;
  const int L = xcd_remap(blockIdx.x, nact);
  const int m0 = (L / tilesN) * BMK, n0 = (L % tilesN) * BN;
  const int Fp = a.Fp;

  const uint16_t* a_src[NA];
  const uint16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = (i * 8 + wave) * 64 + lane;
    const int row = c >> 3, lc = (c & 7) ^ (row & 7);
    const int grow = m0 + row;
    const int trow = (row < BMK && grow < R) ? a.rows[grow] : a.zero_row;
    a_src[i] = a.table + (int64_t)trow * Fp + lc * 8;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 8 + wave) * 64 + lane;
    const int row = c >> 3, lc = (c & 7) ^ (row & 7);
    b_src[i] = a.Wh + (int64_t)(n0 + row) * Fp + lc * 8;
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto stage = [&](int p, int kt) {
    unsigned char* As = smem + p * 2 * LDS_TILE_BYTES;
    unsigned char* Bs = As + LDS_TILE_BYTES;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if ((i * 8 + wave) * 8 < BMK)       // wave-uniform: the last instruction may cover only 4 waves
        glds16(a_src[i] + kt * BK, As + (i * 8 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(b_src[i] + kt * BK, Bs + (i * 8 + wave) * 1024);
  };

  const int nk = Fp / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // SCHED 1: the 8 LDS-DMA instructions of the next K-step are not issued as one burst but two
  // at a time in front of each quarter of the MFMA work, with the quarters pinned by sched_barrier
  auto stage_part = [&](int p, int kt, int q) {
    unsigned char* As = smem + p * 2 * LDS_TILE_BYTES;
    unsigned char* Bs = As + LDS_TILE_BYTES;
    if (q < NA && (q * 8 + wave) * 8 < BMK) glds16(a_src[q] + kt * BK, As + (q * 8 + wave) * 1024);
    glds16(b_src[q] + kt * BK, Bs + (q * 8 + wave) * 1024);
  };
  const int frow = lane & 15, fq = lane >> 4;
  for (int t = 0; t < nk; ++t) {
    const int p = t & 1;
    if constexpr (!(ABL & 1) && SCHED == 0) { if (t + 1 < nk) stage(p ^ 1, t + 1); }
    const unsigned char* As = smem + p * 2 * LDS_TILE_BYTES;
    const unsigned char* Bs = As + LDS_TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + fq) ^ (frow & 7)) << 4;
      i16x8 af[MI], bf[4];
      if constexpr (SCHED == 1 && !(ABL & 1)) { if (t + 1 < nk) stage_part(p ^ 1, t + 1, 2 * kk); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        af[mi] = (ABL & 4) ? i16x8{1, 2, 3, 4, 5, 6, 7, (short)mi}
                           : *(const i16x8*)(As + (wm * (MI * 16) + mi * 16 + frow) * 128 + coff);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
        bf[ni] = (ABL & 4) ? i16x8{1, 2, 3, 4, 5, 6, 7, (short)ni}
                           : *(const i16x8*)(Bs + (wn * 64 + ni * 16 + frow) * 128 + coff);
      if constexpr (SCHED == 1) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < MI / 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_abl<T, ABL>(bf[ni], af[mi], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(ABL & 1)) { if (t + 1 < nk) stage_part(p ^ 1, t + 1, 2 * kk + 1); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = MI / 2; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_abl<T, ABL>(bf[ni], af[mi], acc[mi][ni]);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_abl<T, ABL>(bf[ni], af[mi], acc[mi][ni]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // Epilogue: descale, bias, ReLU, dropout.  The MFMA was issued with the operands swapped
  // (D' = W_tile X_tile^T), so in the 16x16 C/D map (col = lane&15, row = 4*(lane>>4)+j) the lane's
  // column is the batch row m and its 4 registers are 4 CONSECUTIVE outputs n: one 16-B store.
  const float descale = 1.0f / (a.scales->sx * a.scales->sw_cur);
  const float dscale = DROP ? 1.0f / (1.0f - a.drop_ratio) : 1.0f;
  const float lo = a.relu ? 0.f : -INFINITY;
  // the lane's 16 bias values, loaded once and together (inside the store loop every quad waited for its own loads AND -- vmcnt counts
  // stores -- for the previous quad's store: kernels_gemm_ph.hip, epilogue)
  float bq[4][4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = n0 + wn * 64 + ni * 16 + fq * 4;
    if (VEC) {
      const float4 b4 = n < a.D ? *(const float4*)(a.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
      bq[ni][0] = b4.x; bq[ni][1] = b4.y; bq[ni][2] = b4.z; bq[ni][3] = b4.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) bq[ni][j] = n + j < a.D ? a.bias[n + j] : 0.f;
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * (MI * 16) + mi * 16 + frow;
    if (m >= R) continue;
    int64_t ref_row = 0;
    if (DROP) {
      const int bb = m / a.CN, ch = m - bb * a.CN;
      ref_row = (int64_t)ch * a.B + bb;
    }
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + fq * 4;
      if (n >= a.D) continue;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = fmaxf(acc[mi][ni][j] * descale + bq[ni][j], lo);
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

// ------------------------------------------------------------------------------- wgrad --------
// LDS operand image: [64 k-rows][256 halves] = 512-B rows of 32 16-B chunks, both operands
// k-major exactly as they sit in HBM (dY rows / gathered feature rows).  MFMA fragments need 8
// consecutive k for one m (or n): read with ds_read_b64_tr_b16 (4 k-rows x 16 columns per 16-lane
// group, delivered column-major).  chunk' = chunk ^ (h(row) << 1), h = (row&3) | ((row>>3)&1)<<2,
// puts the 8 row segments of one 32-lane half on disjoint banks (tools/lds_banks.py).
template <int ABL>
__device__ __forceinline__ i16x4 tr_read(const unsigned char* p) {
  if constexpr (ABL & 4) return i16x4{1, 2, 3, 4};
  else return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4*)(p));
}
__device__ __forceinline__ int wg_h(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <typename T, bool TR, int ABL = 0, int SCHED = 0>
__global__ __launch_bounds__(GEMM_THREADS) void k_wgrad_gemm(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int tilesM = a.Dp / BM, tilesN = a.Fp / BN;
  // logical order: tm fastest (the two M tiles of one (split, tn) read the same feature bytes)
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = L % tilesM, tn = (L / tilesM) % tilesN, sp = L / (tilesM * tilesN);
  const int m0 = tm * BM, n0 = tn * BN;
  int total_steps = a.Rp / BK, kps = a.ksteps_per_split;
  if (a.n_dev) {                      // dedup mode: live K extent in device memory
    total_steps = (*a.n_dev + BK - 1) / BK;
    kps = (total_steps + a.S - 1) / a.S;
  }
  const int k_begin = sp * kps;
  int k_end = k_begin + kps;
  if (k_end > total_steps) k_end = total_steps;
  const int nk = k_end > k_begin ? k_end - k_begin : 0;

  int srow[4], slc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (i * 8 + wave) * 64 + lane;
    srow[i] = c >> 5;
    slc[i] = (c & 31) ^ (wg_h(srow[i]) << 1);
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int mi = 0; mi < 8; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  int32_t rid[4];   // table rows of the NEXT k-step to stage
  auto load_ids = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rid[i] = a.rows[(int64_t)(k_begin + kt) * BK + srow[i]];
  };
  auto stage = [&](int p, int kt) {
    unsigned char* As = smem + p * 2 * LDS_TILE_BYTES;
    unsigned char* Bs = As + LDS_TILE_BYTES;
    const int64_t kg = (int64_t)(k_begin + kt) * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a.dYh + (kg + srow[i]) * a.Dp + m0 + slc[i] * 8, As + (i * 8 + wave) * 1024);
      glds16(a.table + (int64_t)rid[i] * a.Fp + n0 + slc[i] * 8, Bs + (i * 8 + wave) * 1024);
    }
  };

  if (nk > 0) {
    load_ids(0);
    stage(0, 0);
    if (nk > 1) load_ids(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  for (int t = 0; t < nk; ++t) {
    const int p = t & 1;
    if constexpr (!(ABL & 1) && !(SCHED & 1)) {
      if (t + 1 < nk) {
        stage(p ^ 1, t + 1);
        if (t + 2 < nk) load_ids(t + 2);
      }
    }
    const unsigned char* As = smem + p * 2 * LDS_TILE_BYTES;
    const unsigned char* Bs = As + LDS_TILE_BYTES;
    auto stage_part = [&](int q) {       // SCHED 1: two LDS-DMA instructions in front of each MFMA quarter
      if (t + 1 >= nk) return;
      unsigned char* An = smem + (p ^ 1) * 2 * LDS_TILE_BYTES;
      unsigned char* Bn = An + LDS_TILE_BYTES;
      const int64_t kg = (int64_t)(k_begin + t + 1) * BK;
      glds16(a.dYh + (kg + srow[q]) * a.Dp + m0 + slc[q] * 8, An + (q * 8 + wave) * 1024);
      glds16(a.table + (int64_t)rid[q] * a.Fp + n0 + slc[q] * 8, Bn + (q * 8 + wave) * 1024);
    };
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      i16x8 af[8], bf[4];
      if constexpr ((SCHED & 1) && !(ABL & 1)) { stage_part(2 * kk); __builtin_amdgcn_sched_barrier(0); }
      if constexpr (TR) {
        const int row1 = kk * 32 + 8 * g + q;
        const int hx = wg_h(row1) << 1;       // identical for row1 + 4
        const int rbase = row1 * 512 + (pp & 1) * 8;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int off = rbase + (((wm * 16 + mi * 2 + (pp >> 1)) ^ hx) << 4);
          const i16x4 lo = tr_read<ABL>(As + off);
          const i16x4 hi = tr_read<ABL>(As + off + 4 * 512);
          af[mi] = i16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int off = rbase + (((wn * 8 + ni * 2 + (pp >> 1)) ^ hx) << 4);
          const i16x4 lo = tr_read<ABL>(Bs + off);
          const i16x4 hi = tr_read<ABL>(Bs + off + 4 * 512);
          bf[ni] = i16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
      } else {
        // reference fragment loader: eight 16-bit LDS reads per fragment (slow, layout-obvious)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
          const int col = wm * 128 + mi * 16 + li;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int row = kk * 32 + 8 * g + j;
            af[mi][j] = *(const short*)(As + row * 512 + (((col >> 3) ^ (wg_h(row) << 1)) << 4) +
                                        (col & 7) * 2);
          }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int col = wn * 64 + ni * 16 + li;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int row = kk * 32 + 8 * g + j;
            bf[ni][j] = *(const short*)(Bs + row * 512 + (((col >> 3) ^ (wg_h(row) << 1)) << 4) +
                                        (col & 7) * 2);
          }
        }
      }
      if constexpr (SCHED != 0) {
        if constexpr (SCHED & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_abl<T, ABL>(bf[ni], af[mi], acc[mi][ni]);
        if constexpr (SCHED & 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (SCHED & 1) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!(ABL & 1)) { stage_part(2 * kk + 1); __builtin_amdgcn_sched_barrier(0); }
        }
        if constexpr (SCHED & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mi = 4; mi < 8; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_abl<T, ABL>(bf[ni], af[mi], acc[mi][ni]);
        if constexpr (SCHED & 2) __builtin_amdgcn_s_setprio(0);
        if constexpr (SCHED & 1) __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int mi = 0; mi < 8; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mfma_abl<T, ABL>(bf[ni], af[mi], acc[mi][ni]);
      }
    }
    if constexpr ((SCHED & 1) && !(ABL & 1)) { if (t + 2 < nk) load_ids(t + 2); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // swapped operands (D' = X_tile^T dY_tile): lane column = m (output row d), registers = 4
  // consecutive n (feature columns) -> 16-B stores into the split's fp32 slab.
  float* slab = a.slabs + (int64_t)sp * slab_pitch(a.Dp, a.Fp);
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const int m = m0 + wm * 128 + mi * 16 + li;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + wn * 64 + ni * 16 + g * 4;
      *(float4*)(slab + (int64_t)m * a.Fp + n) =
          make_float4(acc[mi][ni][0], acc[mi][ni][1], acc[mi][ni][2], acc[mi][ni][3]);
    }
  }
}

// ------------------------------------------------------------------------------- launchers ----
// VV_GEMM_VARIANT: 5 (default) = the phase-staggered kernels of kernels_gemm_ph.hip; 6 / 7 = only the weight-gradient /
// only the forward one of them; 8 = phase-staggered forward + four-wave weight gradient; 0 = the round-1 kernels of this
// file (two-buffer K = 64 steps, full drain per step), kept as the measured baseline and for VV_WGRAD_TR=0 (operands
// transposed while staging instead of transposed LDS reads).  VV_ABLATE applies to the phase-staggered kernels.
// (per context: KernelOpts, vv_internal.h; the variants and ablations are settable in a -DVV_LAB build only)
int gemm_variant() { return ko().gemm_variant; }
bool ablate_on() { return ko().ablate != 0; }

template <typename T, bool DROP, bool VEC>
static void launch_fwd_t(const FwdArgs& a, hipStream_t s) {
  const int Rp = (int)round_up(a.R, R_ALIGN), Dp = (int)round_up(a.D, D_ALIGN);
  const dim3 block(GEMM_THREADS);
  if constexpr (!DROP && VEC) {
    // balanced M tiling: the tile height 32*MI that needs the least (rounds of 256 WGs) x (tile height).  Dedup mode sizes
    // the tiles for the expected row count (R_hint, a few steps old: 3 % + 64 rows of slack) while the grid covers the
    // worst case R; surplus workgroups exit at once.
    const int Rh = a.n_dev && a.R_hint > 0 ? (int)std::min<long>(a.R, a.R_hint + a.R_hint / 32 + 64) : a.R;
    int best = 8; long best_cost = -1;
    for (int mi = 8; mi >= (a.n_dev ? 4 : 7); --mi) {
      const long tiles = ((Rh + 32 * mi - 1) / (32 * mi)) * (long)(Dp / BN);
      const long cost = ((tiles + 255) / 256) * mi;
      if (best_cost < 0 || cost < best_cost) { best = mi; best_cost = cost; }
    }
#define VV_FWD_MI(M)                                                                                          \
    if (best == M) {                                                                                         \
      static bool o = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm<T, DROP, VEC, 0, M, 1>,             \
                       hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES), true);                   \
      (void)o;                                                                                               \
      const dim3 g2(((a.R + 32 * M - 1) / (32 * M)) * (Dp / BN));                                            \
      VV_LAUNCH((k_fwd_gemm<T, DROP, VEC, 0, M, 1>), g2, block, GEMM_LDS_BYTES, s, a);                       \
      return;                                                                                                \
    }
    VV_FWD_MI(8) VV_FWD_MI(7) VV_FWD_MI(6) VV_FWD_MI(5) VV_FWD_MI(4)
#undef VV_FWD_MI
  }
  static bool once = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm<T, DROP, VEC>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES), true);
  (void)once;
  const dim3 grid((Rp / BM) * (Dp / BN));
  VV_LAUNCH((k_fwd_gemm<T, DROP, VEC>), grid, block, GEMM_LDS_BYTES, s, a);
}

template <typename T>
static void launch_fwd_p(const FwdArgs& a, hipStream_t s) {
  const bool drop = a.drop_ratio > 0.f, vec = a.D % 4 == 0;
  if (drop) { if (vec) launch_fwd_t<T, true, true>(a, s); else launch_fwd_t<T, true, false>(a, s); }
  else { if (vec) launch_fwd_t<T, false, true>(a, s); else launch_fwd_t<T, false, false>(a, s); }
}

void launch_fwd_gemm_ph(int prec, const FwdArgs& a, hipStream_t s);
void launch_fwd_gemm(int prec, const FwdArgs& a, hipStream_t s) {
  const int g_gemm_variant = ko().gemm_variant;
  if (g_gemm_variant == 5 || g_gemm_variant == 7 || g_gemm_variant == 8) { FwdArgs b = a; b.abl = ko().ablate; launch_fwd_gemm_ph(prec, b, s); return; }
  if (prec == 0) launch_fwd_p<F16>(a, s); else launch_fwd_p<BF16>(a, s);
}

template <typename T, bool TR>
static void launch_wgrad_t(const WgradArgs& a, hipStream_t s) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_wgrad_gemm<T, TR>,
                      hipFuncAttributeMaxDynamicSharedMemorySize, GEMM_LDS_BYTES), true);
  (void)once;
  const dim3 grid((a.Dp / BM) * (a.Fp / BN) * a.S), block(GEMM_THREADS);
  VV_LAUNCH((k_wgrad_gemm<T, TR>), grid, block, GEMM_LDS_BYTES, s, a);
}

void launch_wgrad_gemm_ph(int prec, const WgradArgs& a, hipStream_t s);
void launch_wgrad_gemm_w4(int prec, const WgradArgs& a, hipStream_t s);
// the update in the epilogue (WgradArgs::fuse_upd) exists in the phase-staggered kernel only: api.hip asks before it fills WgradUpd
bool wgrad_can_fuse_update() {
  const int v = ko().gemm_variant;
  return (v == 5 || v == 6) && ko().wgrad_tr != 0 && !ko().ablate && !ko().lab_wg_abl;
}
void launch_wgrad_gemm(int prec, const WgradArgs& a, hipStream_t s) {
  const int g_gemm_variant = ko().gemm_variant; const bool g_wgrad_tr = ko().wgrad_tr != 0;
#ifdef VV_LAB
  if (g_gemm_variant == 8 && g_wgrad_tr) { launch_wgrad_gemm_w4(prec, a, s); return; }
#endif
  if ((g_gemm_variant == 5 || g_gemm_variant == 6 || g_gemm_variant == 8) && g_wgrad_tr) {
    // (lab: VV_LAB_WG_ABL ablates this kernel alone, at whatever size the step runs -- VV_ABLATE switches the de-duplication off)
    WgradArgs b = a; b.abl = ko().ablate ? ko().ablate : ko().lab_wg_abl; launch_wgrad_gemm_ph(prec, b, s); return;
  }
  if (prec == 0) { if (g_wgrad_tr) launch_wgrad_t<F16, true>(a, s); else launch_wgrad_t<F16, false>(a, s); }
  else { if (g_wgrad_tr) launch_wgrad_t<BF16, true>(a, s); else launch_wgrad_t<BF16, false>(a, s); }
}

}  // namespace vv
