// kernels_elem.hip -- the HBM-bound kernels of the videovec training step (gfx950 only).
//
//   k_score_loss : everything between ip2 and the loss, forward AND backward, one workgroup per
//                  batch item: context mean, the two L2 normalisations, the (1+Nn) dot-product
//                  scores, max-margin loss + violations, and the gradient w.r.t. ip1_nonorm.
//                  Replaces SLICE dim 0, ELTWISE SUM, NORMALIZATION x2, CONCAT, ELTWISE PROD x(1+Nn),
//                  SUM x(1+Nn), CONCAT dim 1, MAX_MARGIN_LOSS (CPU-only in the reference), SPLIT,
//                  and all their backward passes + ReLU/Dropout backward
//                  (reference: eltwise_layer.cu:34-119, normalization_layer.cu:10-97,
//                  sum_layer.cu:10-55, max_margin_loss_layer.cpp:53-214, split_layer.cu:18-33,
//                  relu_layer.cu:36-59, dropout_layer.cu:44-73).
//   k_reduce     : split-K slabs -> dW, per-item partials -> db (flat gradient buffer).
//   k_sgd        : SGDSolver::ComputeUpdateValue + Blob::Update in one pass
//                  (solver.cpp:502-531, blob.cpp:112-136), also refreshes the half copy of W.
//   table / conversion helpers.
#include <algorithm>
#include "vv_internal.h"

namespace vv {

// 16-byte accesses with the non-temporal hint (streams that nobody reads again soon)
__device__ __forceinline__ float4 nt_load4(const float* p) {
  const f32x4 v = __builtin_nontemporal_load((const f32x4*)p);
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store4(float* p, const float4& v) {
  __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, (f32x4*)p);
}
// Four consecutive split-K partial products at element offset `off` of the slab buffer: fp32, or (S16, WgradArgs::slab16) f16 times the inverse
// factor of their (split, tile).  slab_tile: the tile of element (d, f) in the factor array's order (k_wgrad_gemm_ph: tm * tilesN + tn).
template <bool S16> __device__ __forceinline__ float4 slab_ld4(const float* slabs, int64_t off, float inv) {
  if constexpr (S16) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const u32x2 q = __builtin_nontemporal_load((const u32x2*)((const uint16_t*)slabs + off));
    const h4 h = __builtin_bit_cast(h4, q);
    return make_float4((float)h[0] * inv, (float)h[1] * inv, (float)h[2] * inv, (float)h[3] * inv);
  } else {
    return nt_load4(slabs + off);
  }
}
__device__ __forceinline__ int slab_tile(int d, int f, int Fp) { return (d >> 8) * (Fp >> 8) + (f >> 8); }

constexpr int SL_THREADS = 256;

// Item of a workgroup in the item-major kernels (one workgroup per batch item).  Workgroups b, b + 8, ... share an XCD (and its L2): instead
// of dealing the items round-robin, XCD x takes the contiguous items [x B/8, (x+1) B/8) -- neighbouring items' own rows (target, context)
// are neighbouring rows of the de-duplicated H, so an XCD's L2 sees runs instead of every eighth row: k_score_fwd 26.5 -> 25.2 us at the benchmark's batch
// (bit-identical results: only which workgroup takes which item changes).  (rr: the lab's switch back to round-robin.)
__device__ __forceinline__ int item_of_block(int rr) {
  const int n = (int)gridDim.x, bid = (int)blockIdx.x;
  if (rr || (n & 7)) return bid;
  return (bid & 7) * (n >> 3) + (bid >> 3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over the 64 lanes with DPP row shifts / row broadcasts (no LDS crossbar traffic, six fused add+DPP
// instructions against twelve bpermute + add for the xor butterfly); the total is valid in LANE 63 only.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum63(float v) {
  v = dpp_add<0x111, 0xF>(v);    // row_shr:1
  v = dpp_add<0x112, 0xF>(v);    // row_shr:2
  v = dpp_add<0x114, 0xF>(v);    // row_shr:4
  v = dpp_add<0x118, 0xF>(v);    // row_shr:8   -> lane 15 of every row holds the row's sum
  v = dpp_add<0x142, 0xA>(v);    // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);    // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
  return v;
}

// block-wide sum for 256 threads; red must hold >= 4 floats; result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum63(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ------------------------------------------------------------------------------- score/loss ---
// Rows of item b: r = b*CN + ch, ch 0 = target, 1..C-1 = context, C.. = negatives.
// Math:
//   A = sum_j c_j H[j],  Ah = A/(|A|+eps)
//   per q in {0, C..}: n_q = |H[q]|, t_q = Ah.H[q], score_q = t_q/(n_q+eps)
//   d_k = s+ - s-_k, h = max(0, margin - d), loss += w_b h^2 (L2) or w_b |h| (L1), viol += d < 0
//   g_k = grad_scale * w_b * (2h | [h>0]);  c_0 = -sum_k g_k, c_{C+k} = g_k
//   dAh = sum_q c_q H[q]/(n_q+eps);   dH[q] = c_q (n_q^2 Ah - H[q] t_q)/(n_q^3 + eps)
//   dA  = (sA dAh - A (A.dAh))/(sA^1.5 + eps), dH[j] = c_j dA
//   dY  = dH * drop_scale * [H > 0]
// block-wide sums over NW waves; red must hold >= 3 * NW floats; results broadcast to all threads
template <int NW>
__device__ __forceinline__ float block_sum_w(float v, float* red) {
  v = wave_sum63(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += red[w];
  return s;
}
// NT threads per item: 256 (four waves) where a batch brings a thousand items; 1024 for wide rows in small batches (the reference's
// shipped configuration: 128 items of 15 rows x 4096 columns -- with four waves per item half of the chip's SIMDs had no wave at all
// and every row was one wave's serial chain of loads: 55 us for 47 MB)
template <typename T, bool VEC, int NT = SL_THREADS>
__global__ __launch_bounds__(NT) void k_score_loss(ScoreArgs a) {
  constexpr int NWV = NT / 64;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int D = a.D, C = a.C, Nn = a.Nn, CN = C + Nn;
  float* A = sm;               // [D] context mean
  float* Ah = A + D;           // [D] normalised context mean
  const int PD = D > 4 * NT ? D : 4 * NT;   // (phase 4: G row groups of D columns, G D <= 4 NT when D / 4 < NT -- score_loss_lds() sizes the same)
  float* acc0 = Ah + D;        // [G][D] row-group partial dAh (G*D <= PD)
  float* acc1 = acc0 + PD;     // [G][D] row-group partial db
  float* n2 = acc1 + PD;       // [CN] squared norms
  float* tq = n2 + CN;         // [CN] dots with Ah
  float* cq = tq + CN;         // [CN] upstream coefficients
  float* red = cq + CN;        // [16]
  int* hoff = (int*)(red + 16); // [CN] row of H holding channel ch (dedup: the shared per-slot row)
  int* ooff = hoff + CN;       // [CN] row of dYh receiving channel ch's gradient
  float* k1 = (float*)(ooff + CN);   // [CN] per-row constants of the backward pass (see phase 4)
  float* k2 = k1 + CN;
  float* k3 = k2 + CN;
  const int b = item_of_block(a.items_rr), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float eps = 1e-10f;
  __shared__ float ggs[16];
  float sgm;
  if (!gg_begin(a.guard, ggs, sgm)) return;
  const float sg = a.sg * sgm;
  for (int ch = tid; ch < CN; ch += NT) {
    const int r = b * CN + ch;
    hoff[ch] = a.map ? a.map[r] : r;
    ooff[ch] = a.map ? a.seg_start[a.map[r]] + a.ord[r] : r;
  }
  __syncthreads();
#define HROW(ch) (a.H + (int64_t)hoff[ch] * D)

  // ---- phase 1: context mean (column-parallel) and its norm
  float ssq = 0.f;
  for (int d = tid; d < D; d += NT) {
    float s = 0.f;
    for (int j = 1; j < C; ++j) s += a.coeff[j - 1] * HROW(j)[d];
    A[d] = s;
    ssq += s * s;
  }
  const float sA = block_sum_w<NWV>(ssq, red);
  const float nA = sqrtf(sA) + eps;
  for (int d = tid; d < D; d += NT) {
    Ah[d] = A[d] / nA;
  }
  __syncthreads();

  // ---- phase 2: norms and dots of target / negative rows (one row per wave at a time)
  for (int qi = wave; qi < 1 + Nn; qi += NWV) {
    const int ch = qi == 0 ? 0 : C + qi - 1;
    const float* h = HROW(ch);
    float s = 0.f, t = 0.f;
    if (VEC) {
      for (int d = lane * 4; d < D; d += 256) {
        const float4 x = *(const float4*)(h + d);
        const float4 y = *(const float4*)(Ah + d);
        s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
        t += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
    } else {
      for (int d = lane; d < D; d += 64) { const float x = h[d]; s += x * x; t += x * Ah[d]; }
    }
    s = wave_sum63(s); t = wave_sum63(t);
    if (lane == 63) { n2[ch] = s; tq[ch] = t; }
  }
  __syncthreads();

  // ---- phase 3: scores, hinge, loss, coefficients
  const float sp = tq[0] / (sqrtf(n2[0]) + eps);
  const float wb = a.item_w ? a.item_w[b] : 1.f;
  float lsum = 0.f, vsum = 0.f, gsum = 0.f;
  for (int k = tid; k < Nn; k += NT) {
    const int ch = C + k;
    const float sn = tq[ch] / (sqrtf(n2[ch]) + eps);
    const float d = sp - sn;
    const float h = fmaxf(0.f, a.margin - d);
    float g;     // wb: max_margin_loss_layer.cpp:82-97 (sqrt(w)*h squared, or w*h) and :152-186
    if (a.norm == 2) { lsum += wb * h * h; g = 2.f * wb * h * a.grad_scale; }
    else { lsum += wb * fabsf(h); g = h > 0.f ? wb * a.grad_scale : 0.f; }
    vsum += d < 0.f ? 1.f : 0.f;
    gsum += g;
    cq[ch] = g;
    if (a.s_bogus) a.s_bogus[(int64_t)b * Nn + k] = sn;
  }
  lsum = block_sum_w<NWV>(lsum, red);
  vsum = block_sum_w<NWV>(vsum, red);
  gsum = block_sum_w<NWV>(gsum, red);
  if (tid == 0) {
    cq[0] = -gsum;
    a.loss_part[b] = lsum;
    a.viol_part[b] = vsum;
    if (a.s_true) a.s_true[b] = sp;
    // workgroup 0 is in the kernel's first round: "this step's forward GEMM has finished" (placed here, where few values
    // are live: at the kernel's head it cost the RPW 7 form six registers and, at 133, an occupancy step)
    if (a.gate_host && b == 0) __hip_atomic_store(a.gate_host, a.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  // per-row constants, once per row instead of once per (row, column group): g*sg = k1*Ah - k2*x, dAh += k3*x
  for (int qi = tid; qi < 1 + Nn; qi += NT) {
    const int ch = qi == 0 ? 0 : C + qi - 1;
    const float c = cq[ch], s = n2[ch], rs = sqrtf(s);
    const float cd = c * __builtin_amdgcn_rcpf(s * rs + eps) * a.drop_scale * sg;
    k1[ch] = cd * s; k2[ch] = cd * tq[ch]; k3[ch] = c * __builtin_amdgcn_rcpf(rs + eps);
  }
  __syncthreads();

  // ---- phase 4: backward of the normalised target / negative rows, column-parallel: a thread
  // owns one group of W consecutive columns (16-B loads) and walks the rows of its row group,
  // keeping its dAh / db partial sums in registers (deterministic order, no atomics).
  float gmx = 0.f;                            // max |g| in scaled units before rounding (f16 gradient-scale guard)
  constexpr int W = VEC ? 4 : 1;
  const int Dv = D / W;                       // column groups
  const int Dvp = Dv < NT ? Dv : NT;
  const int G = NT / Dvp;             // row groups running side by side
  if (tid < G * Dvp) {
    const int rg = tid / Dvp;
    for (int cg = tid % Dvp; cg < Dv; cg += Dvp) {
      const int d = cg * W;
      float ya[W], pa[W], pb[W];
#pragma unroll
      for (int e = 0; e < W; ++e) { ya[e] = Ah[d + e]; pa[e] = 0.f; pb[e] = 0.f; }
      for (int qi = rg; qi < 1 + Nn; qi += G) {
        const int ch = qi == 0 ? 0 : C + qi - 1;
        const float r1 = k1[ch], r2 = k2[ch], r3 = k3[ch];
        const float* h = HROW(ch) + d;
        float xv[W];
        if (VEC) { const float4 x = *(const float4*)h; xv[0] = x.x; xv[1 % W] = x.y; xv[2 % W] = x.z; xv[3 % W] = x.w; }
        else xv[0] = h[0];
        uint16_t* dy = a.dYh + (int64_t)ooff[ch] * a.Dp + d;
        uint16_t o[W];
#pragma unroll
        for (int e = 0; e < W; ++e) {
          pa[e] += r3 * xv[e];
          float g = r1 * ya[e] - r2 * xv[e];            // already times sg (a power of two: exact)
          g = xv[e] > 0.f ? g : 0.f;
          pb[e] += g;
          gmx = fmaxf(gmx, fabsf(g));
          o[e] = T::from_float(g);
        }
        if (VEC) *(uint2*)dy = make_uint2(o[0] | ((uint32_t)o[1 % W] << 16), o[2 % W] | ((uint32_t)o[3 % W] << 16));
        else dy[0] = o[0];
      }
#pragma unroll
      for (int e = 0; e < W; ++e) { acc0[rg * D + d + e] = pa[e]; acc1[rg * D + d + e] = pb[e] * (1.f / sg); }
    }
  }
  __syncthreads();

  // ---- phase 5: backward of the context normalisation and mean (column-parallel)
  float dot = 0.f;
  for (int d = tid; d < D; d += NT) {
    float u = 0.f;
    for (int gI = 0; gI < G; ++gI) u += acc0[gI * D + d];
    acc0[d] = u;                 // each column is owned by one thread from here on
    dot += A[d] * u;
  }
  dot = block_sum_w<NWV>(dot, red);
  const float inv_denA = 1.f / (sA * sqrtf(sA) + eps);
  for (int d = tid; d < D; d += NT) {
    const float dA = (sA * acc0[d] - A[d] * dot) * inv_denA;
    float dbv = 0.f;
    for (int gI = 0; gI < G; ++gI) dbv += acc1[gI * D + d];
    for (int j = 1; j < C; ++j) {
      const float x = HROW(j)[d];
      float g = a.coeff[j - 1] * dA * a.drop_scale;
      g = x > 0.f ? g : 0.f;
      dbv += g;
      gmx = fmaxf(gmx, fabsf(g * sg));
      a.dYh[(int64_t)ooff[j] * a.Dp + d] = T::from_float(g * sg);
    }
    a.dbp[(int64_t)b * D + d] = dbv;
  }
  if (a.guard.gg) gg_end(a.guard, gg_block_max(gmx, ggs));
}

// Register-resident variant for small (1+Nn) x D: each wave keeps its target / negative rows in
// VGPRs between the forward reductions and the backward pass, so every row of ip2 is read from HBM
// exactly once (the streaming kernel above re-reads them, and at ~8 workgroups per CU the 110 KB per
// item do not survive in L2).  RPW = rows per wave, DV = float4 chunks per lane (D = 256*DV).
template <int NW>
__device__ __forceinline__ void block_sum3_w(float& x, float& y, float& z, float* red) {
  x = wave_sum63(x); y = wave_sum63(y); z = wave_sum63(z);
  __syncthreads();
  if ((threadIdx.x & 63) == 63) { const int w = threadIdx.x >> 6; red[w] = x; red[NW + w] = y; red[2 * NW + w] = z; }
  __syncthreads();
  x = y = z = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { x += red[w]; y += red[NW + w]; z += red[2 * NW + w]; }
}

// NW waves per item, RPW target/negative rows per wave (row qi lives in wave qi % NW), DV float4 chunks per lane
// (D = 256 * DV); the context rows and the per-column tail are owned column-wise, CV = D / (64 NW) columns per thread.
// Eight waves with two rows each keep the register footprint small enough for every workgroup of a 1024-item batch to
// be resident at once (the four-wave form held 13 row slots per wave: ~200 VGPRs, two rounds of workgroups, and the
// block-wide sums of each phase were exposed latency).
template <typename T, int NW, int RPW, int DV>
__global__ __launch_bounds__(64 * NW) void k_score_loss_reg(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int THREADS = 64 * NW;
  constexpr int CV = 256 * DV / THREADS;
  static_assert(CV >= 1 && CV * THREADS == 256 * DV, "D must be a multiple of the thread count");
  constexpr int CXM = 6;       // context rows kept in registers (C - 1 <= CXM on this path)
  const int D = a.D, C = a.C, Nn = a.Nn, CN = C + Nn;
  float* A = sm;               // [D]
  float* Ah = A + D;           // [D]
  float* acc0 = Ah + D;        // [NW][D] per-wave partial dAh
  float* acc1 = acc0 + NW * D; // [NW][D] per-wave partial db
  float* n2 = acc1 + NW * D;   // [CN]
  float* tq = n2 + CN;         // [CN]
  float* cq = tq + CN;         // [CN]
  float* red = cq + CN;        // [3 NW]
  int* ooff = (int*)(red + 3 * NW);   // [CN] row of dYh receiving channel ch's gradient
  const int b = item_of_block(a.items_rr), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float eps = 1e-10f;
  __shared__ float ggs[16];
  float sgm;
  if (!gg_begin(a.guard, ggs, sgm)) return;
  const float sg = a.sg * sgm;

  // Every global read of ip2 is issued up front: the target / negative rows of this wave
  // (qi = wave, wave+NW, ...) and the context rows (thread tid owns columns tid + THREADS v).
  float4 x[RPW][DV];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int qi = wave + NW * k;
    const int ch = qi == 0 ? 0 : C + qi - 1;
    const int r = b * CN + (qi <= Nn ? ch : 0);
    const int hr = a.map ? a.map[r] : r;
#pragma unroll
    for (int v = 0; v < DV; ++v)
      x[k][v] = qi <= Nn ? *(const float4*)(a.H + (int64_t)hr * D + lane * 4 + v * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float cx[CXM][CV];
#pragma unroll
  for (int j = 0; j < CXM; ++j) {
    const int r = b * CN + (j + 1 < C ? j + 1 : 0);
    const int hr = a.map ? a.map[r] : r;
#pragma unroll
    for (int v = 0; v < CV; ++v) cx[j][v] = j + 1 < C ? a.H[(int64_t)hr * D + tid + v * THREADS] : 0.f;
  }
  float cf[CXM];
#pragma unroll
  for (int j = 0; j < CXM; ++j) cf[j] = j + 1 < C ? a.coeff[j] : 0.f;
  for (int ch = tid; ch < CN; ch += THREADS) {
    const int r = b * CN + ch;
    ooff[ch] = a.map ? a.seg_start[a.map[r]] + a.ord[r] : r;
  }

  // ---- phase 1: context mean and its norm
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CXM; ++j) s += cf[j] * cx[j][v];
    A[tid + v * THREADS] = s;
    ssq += s * s;
  }
  const float sA = block_sum_w<NW>(ssq, red);
  const float nA = sqrtf(sA) + eps;
#pragma unroll
  for (int v = 0; v < CV; ++v) Ah[tid + v * THREADS] = A[tid + v * THREADS] / nA;
  __syncthreads();

  // ---- phase 2: norms and dots from registers
  float4 y[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) y[v] = *(const float4*)(Ah + lane * 4 + v * 256);
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int qi = wave + NW * k;
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const float4 xx = x[k][v];
      s += xx.x * xx.x + xx.y * xx.y + xx.z * xx.z + xx.w * xx.w;
      t += xx.x * y[v].x + xx.y * y[v].y + xx.z * y[v].z + xx.w * y[v].w;
    }
    s = wave_sum63(s); t = wave_sum63(t);
    if (lane == 63 && qi <= Nn) { const int ch = qi == 0 ? 0 : C + qi - 1; n2[ch] = s; tq[ch] = t; }
  }
  __syncthreads();

  // ---- phase 3: scores, hinge, loss, coefficients
  const float sp = tq[0] / (sqrtf(n2[0]) + eps);
  const float wb = a.item_w ? a.item_w[b] : 1.f;
  float lsum = 0.f, vsum = 0.f, gsum = 0.f;
  for (int k = tid; k < Nn; k += THREADS) {
    const int ch = C + k;
    const float sn = tq[ch] / (sqrtf(n2[ch]) + eps);
    const float d = sp - sn;
    const float h = fmaxf(0.f, a.margin - d);
    float g;     // wb: max_margin_loss_layer.cpp:82-97 (sqrt(w)*h squared, or w*h) and :152-186
    if (a.norm == 2) { lsum += wb * h * h; g = 2.f * wb * h * a.grad_scale; }
    else { lsum += wb * fabsf(h); g = h > 0.f ? wb * a.grad_scale : 0.f; }
    vsum += d < 0.f ? 1.f : 0.f;
    gsum += g;
    cq[ch] = g;
    if (a.s_bogus) a.s_bogus[(int64_t)b * Nn + k] = sn;
  }
  block_sum3_w<NW>(lsum, vsum, gsum, red);
  if (tid == 0) {
    cq[0] = -gsum;
    a.loss_part[b] = lsum;
    a.viol_part[b] = vsum;
    if (a.s_true) a.s_true[b] = sp;
    // workgroup 0 is in the kernel's first round: "this step's forward GEMM has finished" (placed here, where few values
    // are live: at the kernel's head it cost the RPW 7 form six registers and, at 133, an occupancy step)
    if (a.gate_host && b == 0) __hip_atomic_store(a.gate_host, a.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();

  // ---- phase 4: backward of this wave's rows from registers; one coalesced 512-B store per chunk.  Per row three
  // scalars k1 = c n^2/den, k2 = c t/den (both times drop_scale * sg) and k3 = c/(n+eps); per element
  // g*sg = k1*Ah - k2*x (masked), dAh += k3*x, db*sg += g*sg.  sg is a power of two, so carrying it through the sums
  // and dividing at the end changes no bit.
  float gmx = 0.f;
  float4 pa[DV], pb[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) { pa[v] = make_float4(0.f, 0.f, 0.f, 0.f); pb[v] = pa[v]; }
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int qi = wave + NW * k;
    if (qi > Nn) continue;                         // wave-uniform
    const int ch = qi == 0 ? 0 : C + qi - 1;
    const float c = cq[ch], s = n2[ch], t = tq[ch];
    const float rs = sqrtf(s);
    const float k3 = c * __builtin_amdgcn_rcpf(rs + eps);
    const float cd = c * __builtin_amdgcn_rcpf(s * rs + eps) * a.drop_scale * sg;
    const float k1 = cd * s, k2 = cd * t;
    uint16_t* dy = a.dYh + (int64_t)ooff[ch] * a.Dp;
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const float xv[4] = {x[k][v].x, x[k][v].y, x[k][v].z, x[k][v].w};
      const float yv[4] = {y[v].x, y[v].y, y[v].z, y[v].w};
      float g[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        g[e] = k1 * yv[e] - k2 * xv[e];
        g[e] = xv[e] > 0.f ? g[e] : 0.f;
        gmx = fmaxf(gmx, fabsf(g[e]));
      }
      pa[v].x += k3 * xv[0]; pa[v].y += k3 * xv[1]; pa[v].z += k3 * xv[2]; pa[v].w += k3 * xv[3];
      pb[v].x += g[0]; pb[v].y += g[1]; pb[v].z += g[2]; pb[v].w += g[3];
      const uint32_t lo = T::from_float(g[0]) | ((uint32_t)T::from_float(g[1]) << 16);
      const uint32_t hi = T::from_float(g[2]) | ((uint32_t)T::from_float(g[3]) << 16);
      *(uint2*)(dy + lane * 4 + v * 256) = make_uint2(lo, hi);
    }
  }
  const float inv_sg = 1.f / sg;
#pragma unroll
  for (int v = 0; v < DV; ++v) {
    *(float4*)(acc0 + wave * D + lane * 4 + v * 256) = pa[v];
    *(float4*)(acc1 + wave * D + lane * 4 + v * 256) = make_float4(pb[v].x * inv_sg, pb[v].y * inv_sg, pb[v].z * inv_sg, pb[v].w * inv_sg);
  }
  __syncthreads();

  // ---- phase 5: backward of the context normalisation and mean (context rows still in registers)
  float dot = 0.f;
  float u[CV];
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    float us = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) us += acc0[w * D + d];
    u[v] = us;
    dot += A[d] * us;
  }
  dot = block_sum_w<NW>(dot, red);
  const float inv_denA = 1.f / (sA * sqrtf(sA) + eps);
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    const float dA = (sA * u[v] - A[d] * dot) * inv_denA;
    float dbv = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) dbv += acc1[w * D + d];
#pragma unroll
    for (int j = 0; j < CXM; ++j) {
      if (j + 1 < C) {
        float g = cf[j] * dA * a.drop_scale;
        g = cx[j][v] > 0.f ? g : 0.f;
        dbv += g;
        gmx = fmaxf(gmx, fabsf(g * sg));
        a.dYh[(int64_t)ooff[j + 1] * a.Dp + d] = T::from_float(g * sg);
      }
    }
    a.dbp[(int64_t)b * D + d] = dbv;
  }
  if (a.guard.gg) gg_end(a.guard, gg_block_max(gmx, ggs));
}
#undef HROW


template <typename T>
static bool launch_score_loss_reg(const ScoreArgs& a, hipStream_t s) {
  // fast path: D == 512, at most 6 context rows, at most 56 target/negative rows (52 with four waves)
  if (a.D != 512 || a.C - 1 > 6) return false;
  const int rows = 1 + a.Nn;
  const int nw = ko().score_waves == 4 ? 4 : 8;
  const size_t lds = sizeof(float) * ((size_t)(2 + 2 * nw) * a.D + 4 * (a.C + a.Nn) + 3 * nw);
#define VV_SLR(NW, RPW)                                                                                   \
  do {                                                                                                    \
    (void)hipFuncSetAttribute((const void*)k_score_loss_reg<T, NW, RPW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    VV_LAUNCH((k_score_loss_reg<T, NW, RPW, 2>), dim3(a.B), dim3(64 * NW), lds, s, a);                    \
  } while (0)
  if (nw == 4) { if (rows > 52) return false; VV_SLR(4, 13); return true; }
  if (rows <= 16) VV_SLR(8, 2);
  else if (rows <= 32) VV_SLR(8, 4);
  else if (rows <= 56) VV_SLR(8, 7);
  else return false;
#undef VV_SLR
  return true;
}

// ---- rows of H as fp32 or as f16 (round 6, FwdArgs::h16 / ScoreArgs::h16).  The kernels below hold a row as float4 pieces per lane; which
// columns a piece covers depends on the storage: fp32 rows are read 16 bytes = 4 columns per lane and piece (piece v: columns lane 4 + 256 v),
// f16 rows 16 bytes = 8 columns per lane (pieces 2 w, 2 w + 1: columns 512 w + lane 8 .. + 3 and + 4 .. + 7) -- 8-byte row loads run at ~0.6 of the
// 16-byte rate per byte (profiles/r04_h16_intermediate.txt).  h_col: first column of piece v; h_row: the DV pieces of one row.
template <bool H16> __device__ __forceinline__ int h_col(int lane, int v) { return H16 ? (v >> 1) * 512 + lane * 8 + (v & 1) * 4 : lane * 4 + v * 256; }
template <bool H16> __device__ __forceinline__ float h_at(const float* H, int64_t i) { return H16 ? (float)((const _Float16*)H)[i] : H[i]; }
template <bool H16, int DV> __device__ __forceinline__ void h_row(const float* H, int64_t row, int D, int lane, float4* x) {
  if constexpr (H16) {
    static_assert(DV % 2 == 0, "f16 rows are read 8 columns at a time");
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const _Float16* hp = (const _Float16*)H + row * D + lane * 8;
#pragma unroll
    for (int w = 0; w < DV / 2; ++w) {
      const h8 q = *(const h8*)(hp + 512 * w);
      x[2 * w] = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
      x[2 * w + 1] = make_float4((float)q[4], (float)q[5], (float)q[6], (float)q[7]);
    }
  } else {
#pragma unroll
    for (int v = 0; v < DV; ++v) x[v] = *(const float4*)(H + row * D + lane * 4 + v * 256);
  }
}

// ---- segment-wise backward, pass 1: k_score_loss_reg without the per-instance gradient rows.  Forward as there; the
// backward stops at the factored form (vv_internal.h: SegRec): one record per instance, Ah_b and dA_b per item.
// (Sixteen waves per item, four rows per wave -- one item per CU at a time instead of two items of eight waves: 36 against 25 us, round 4.)
// DROP (ScoreArgs::drop): the rows of H are the shared PRE-dropout projections; every instance applies its own mask (and 1 / (1 - ratio))
// as its row arrives, and everything behind that -- norms, scores, records -- is the reference's graph on the masked rows.
// (Round 6, with f16 rows -- 86 VGPRs instead of 112-127 -- two more residencies were measured and not kept: capped at 80 VGPRs = three
// workgroups per CU instead of two, 7 registers spilled: 22.9 -> 23.7 us; FOUR waves per item with 14 rows per wave, three or four 256-thread
// workgroups per CU = all 1024 items resident at once: 23.0 -> 26.2-27.1 us.  tools/sessions/r06_log.md s5, s8.)
template <int NW, int RPW, int DV, bool DROP = false, bool H16 = false>
__global__ __launch_bounds__(64 * NW) void k_score_fwd(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int THREADS = 64 * NW;
  constexpr int CV = 256 * DV / THREADS;
  static_assert(CV >= 1 && CV * THREADS == 256 * DV, "D must be a multiple of the thread count");
  constexpr int CXM = 6;
  const int D = a.D, C = a.C, Nn = a.Nn, CN = C + Nn;
  float* A = sm;               // [D]
  float* Ah = A + D;           // [D]
  float* acc0 = Ah + D;        // [NW][D] per-wave partial dAh
  float* n2 = acc0 + NW * D;   // [CN]
  float* tq = n2 + CN;         // [CN]
  float* cq = tq + CN;         // [CN]
  float* red = cq + CN;        // [3 NW]
  int* ooff = (int*)(red + 3 * NW);   // [CN] grouped position of channel ch's instance
  const int b = item_of_block(a.items_rr), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float eps = 1e-10f;
#ifdef VV_LAB
  // (lab: phase stamps of wave 0, written by thread 0 at the kernel's end -- ScoreArgs::lab_ts)
  uint32_t ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long rt0_ = 0;
#define SF_TS(i) if (a.lab_ts && wave == 0) ts_[i] = (uint32_t)__builtin_readcyclecounter();
  if (a.lab_ts && wave == 0) rt0_ = __builtin_amdgcn_s_memrealtime();
#else
#define SF_TS(i)
#endif
  SF_TS(0)

  // (round 5: the context rows are requested FIRST -- vmcnt completes in order, and the context mean, its block sum and the normalised vector
  // are what the kernel computes first: behind the 14 target / negative rows they waited for all of them, profiles/r05_score_fwd_stamps.txt)
  float cx[CXM][CV];
#pragma unroll
  for (int j = 0; j < CXM; ++j) {
    const int r = b * CN + (j + 1 < C ? j + 1 : 0);
    const int hr = a.map[r];
#pragma unroll
    for (int v = 0; v < CV; ++v) cx[j][v] = j + 1 < C ? h_at<H16>(a.H, (int64_t)hr * D + tid + v * THREADS) : 0.f;
    if (DROP && j + 1 < C) {
      const int64_t rr = (int64_t)(j + 1) * a.B + b;
      const uint32_t rc = drop_row_ctr(rr, D, a.drop.s32);
#pragma unroll
      for (int v = 0; v < CV; ++v) {
        const int col = tid + v * THREADS;
        const uint32_t kp = drop_keep4(a.drop, rr, rc, col & ~3);
        cx[j][v] = ((kp >> (col & 3)) & 1u) ? cx[j][v] * a.drop.scale : 0.f;
      }
    }
  }
  // (round 6, ScoreArgs::prefetch, f16 rows: this workgroup is in the kernel's FIRST round -- blocks 0 .. n/2 - 1 are resident together, two per
  // CU --; block bid + n/2 of the second round runs on the same XCD.  While this item's compute phases run the memory system is idle (every
  // workgroup of a round loads, then computes, in step): one line of the second-round item's rows per thread is requested then, into the XCD's
  // L2, where the second round finds it.  The row index is requested here, with this item's own indices, so that nothing waits for it later.)
  int pf_row = -1;
  if (H16 && a.prefetch && !a.items_rr && !((int)gridDim.x & 7) && THREADS == 512) {
    const int bid2 = (int)blockIdx.x + ((int)gridDim.x >> 1), r2 = tid >> 3;
    if (bid2 < (int)gridDim.x && r2 < CN) pf_row = a.map[((bid2 & 7) * ((int)gridDim.x >> 3) + (bid2 >> 3)) * CN + r2];
  }
  float4 x[RPW][DV];
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int qi = wave + NW * k;
    const int ch = qi == 0 ? 0 : C + qi - 1;
    const int r = b * CN + (qi <= Nn ? ch : 0);
    const int hr = a.map[r];
#ifdef VV_LAB
    if (a.lab_hack) {                                // (lab: what the row loads' time depends on -- results wrong)
      const int64_t rs = a.lab_hack == 1 ? D / 2 : D;
#pragma unroll
      for (int v = 0; v < DV; ++v)
        x[k][v] = (qi <= Nn && !(a.lab_hack == 2 && v > 0)) ? *(const float4*)(a.H + (int64_t)hr * rs + lane * 4 + v * 256) : make_float4(1.f, 0.f, 0.f, 0.f);
    } else
#endif
    {
      h_row<H16, DV>(a.H, hr, D, lane, x[k]);          // (always a valid row: channel 0 stands in past the item's last negative)
#pragma unroll
      for (int v = 0; v < DV; ++v) x[k][v] = qi <= Nn ? x[k][v] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (DROP && qi <= Nn) {
      const int64_t rr = (int64_t)ch * a.B + b;                    // the instance's row in the reference's order
      const uint32_t rc = drop_row_ctr(rr, D, a.drop.s32);
#pragma unroll
      for (int v = 0; v < DV; ++v) {
        const uint32_t kp = drop_keep4(a.drop, rr, rc, h_col<H16>(lane, v));
        x[k][v].x = (kp & 1u) ? x[k][v].x * a.drop.scale : 0.f; x[k][v].y = (kp & 2u) ? x[k][v].y * a.drop.scale : 0.f;
        x[k][v].z = (kp & 4u) ? x[k][v].z * a.drop.scale : 0.f; x[k][v].w = (kp & 8u) ? x[k][v].w * a.drop.scale : 0.f;
      }
    }
  }
  float cf[CXM];
#pragma unroll
  for (int j = 0; j < CXM; ++j) cf[j] = j + 1 < C ? a.coeff[j] : 0.f;
  for (int ch = tid; ch < CN; ch += THREADS) {
    const int r = b * CN + ch;
    ooff[ch] = a.seg_start[a.map[r]] + a.ord[r];
  }

  // ---- context mean and its norm
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < CXM; ++j) s += cf[j] * cx[j][v];
    A[tid + v * THREADS] = s;
    ssq += s * s;
  }
  const float sA = block_sum_w<NW>(ssq, red);
  SF_TS(1)                                         // the context rows have arrived, first block-wide sum
  const float nA = sqrtf(sA) + eps;
  float* Vb = a.V + (int64_t)2 * b * D;
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const float ah = A[tid + v * THREADS] / nA;
    Ah[tid + v * THREADS] = ah;
    Vb[tid + v * THREADS] = ah;
  }
  __syncthreads();

  // ---- norms and dots from registers
  float4 y[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) y[v] = *(const float4*)(Ah + h_col<H16>(lane, v));
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int qi = wave + NW * k;
    float s = 0.f, t = 0.f;
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      const float4 xx = x[k][v];
      s += xx.x * xx.x + xx.y * xx.y + xx.z * xx.z + xx.w * xx.w;
      t += xx.x * y[v].x + xx.y * y[v].y + xx.z * y[v].z + xx.w * y[v].w;
    }
    s = wave_sum63(s); t = wave_sum63(t);
    if (lane == 63 && qi <= Nn) { const int ch = qi == 0 ? 0 : C + qi - 1; n2[ch] = s; tq[ch] = t; }
  }
  __syncthreads();
  SF_TS(2)                                         // every wave's target / negative rows have arrived and are reduced
  if (H16 && a.prefetch) {
    // the second round's rows: one 128-byte line per thread, requested as an LDS-DMA into 256 dead bytes behind the kernel's LDS arrays (a
    // load into a register would land, whenever it lands, in a register the compiler has long handed to something else); never waited for
    const char* pf = (const char*)a.H + (int64_t)(pf_row >= 0 ? pf_row : 0) * D * 2 + (tid & 7) * 128;
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)(ooff + CN + 4)));
    if (pf_row >= 0) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" :: "s"(m0v & ~3u), "v"(pf) : "m0", "memory");
  }

  // ---- scores, hinge, loss, coefficients
  const float sp = tq[0] / (sqrtf(n2[0]) + eps);
  const float wb = a.item_w ? a.item_w[b] : 1.f;
  float lsum = 0.f, vsum = 0.f, gsum = 0.f;
  for (int k = tid; k < Nn; k += THREADS) {
    const int ch = C + k;
    const float sn = tq[ch] / (sqrtf(n2[ch]) + eps);
    const float d = sp - sn;
    const float h = fmaxf(0.f, a.margin - d);
    float g;
    if (a.norm == 2) { lsum += wb * h * h; g = 2.f * wb * h * a.grad_scale; }
    else { lsum += wb * fabsf(h); g = h > 0.f ? wb * a.grad_scale : 0.f; }
    vsum += d < 0.f ? 1.f : 0.f;
    gsum += g;
    cq[ch] = g;
    if (a.s_bogus) a.s_bogus[(int64_t)b * Nn + k] = sn;
  }
  block_sum3_w<NW>(lsum, vsum, gsum, red);
  if (tid == 0) {
    cq[0] = -gsum;
    a.loss_part[b] = lsum;
    a.viol_part[b] = vsum;
    if (a.s_true) a.s_true[b] = sp;
    // workgroup 0 is in the kernel's first round: "this step's forward GEMM has finished" (placed here, where few values
    // are live: at the kernel's head it cost the RPW 7 form six registers and, at 133, an occupancy step)
    if (a.gate_host && b == 0) __hip_atomic_store(a.gate_host, a.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  SF_TS(3)

  // ---- dAh = sum_q k3_q x_q from registers; the record of every target / negative instance
  float bnd = 0.f;
  float4 pa[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) pa[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < RPW; ++k) {
    const int qi = wave + NW * k;
    if (qi > Nn) continue;                         // wave-uniform
    const int ch = qi == 0 ? 0 : C + qi - 1;
    const float c = cq[ch], s = n2[ch], t = tq[ch];
    const float rs = sqrtf(s);
    const float k3 = c * __builtin_amdgcn_rcpf(rs + eps);
    const float cd = c * __builtin_amdgcn_rcpf(s * rs + eps) * a.drop_scale * a.sg;
    bnd = fmaxf(bnd, fabsf(cd * s) + fabsf(cd * t) * rs);           // |alpha| + |beta| |x|: GuardArgs::bound
    if (lane == 0) { SegRec rc; rc.alpha = cd * s; rc.beta = cd * t; rc.vec = 2 * b; rc.pad = b * CN + ch; a.rec[ooff[ch]] = rc; }
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      pa[v].x += k3 * x[k][v].x; pa[v].y += k3 * x[k][v].y; pa[v].z += k3 * x[k][v].z; pa[v].w += k3 * x[k][v].w;
    }
  }
#pragma unroll
  for (int v = 0; v < DV; ++v) *(float4*)(acc0 + wave * D + h_col<H16>(lane, v)) = pa[v];
  if (tid < C - 1) {
    SegRec rc; rc.alpha = a.coeff[tid] * a.drop_scale * a.sg; rc.beta = 0.f; rc.vec = 2 * b + 1; rc.pad = b * CN + tid + 1;
    a.rec[ooff[tid + 1]] = rc;
  }
  if (lane == 0) red[wave] = bnd;                   // (red is free here: block_sum3_w's readers are past the barrier above)
  __syncthreads();
  SF_TS(4)

  // ---- backward of the context normalisation: dA_b
  float bnd_item = 0.f;
  if (tid == 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w) bnd_item = fmaxf(bnd_item, red[w]);
  }
  float dot = 0.f;
  float u[CV];
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    float us = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) us += acc0[w * D + d];
    u[v] = us;
    dot += A[d] * us;
  }
  dot = block_sum_w<NW>(dot, red);
  const float inv_denA = 1.f / (sA * sqrtf(sA) + eps);
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    Vb[D + d] = (sA * u[v] - A[d] * dot) * inv_denA;
  }
  if (tid == 0 && a.bound_out) {
    // context instances: |coeff_j ds sg| max_d |dA_d| <= |alpha_j| 4 sA gsum / (sA^1.5 + eps)   (|u_d| <= sum_q |c_q| = 2 gsum)
    float am = 0.f;
#pragma unroll
    for (int j = 0; j < CXM; ++j) am = fmaxf(am, fabsf(cf[j]));      // (the coefficients are in registers since the kernel's start)
    bnd_item = fmaxf(bnd_item, am * a.drop_scale * a.sg * 4.f * sA * gsum * inv_denA);
    atomicMax(a.bound_out + (b & (GG_BOUND_SLOTS - 1)) * GG_BOUND_STRIDE, ((unsigned long long)(unsigned)a.bound_seq << 32) | __float_as_uint(bnd_item));
  }
#ifdef VV_LAB
  SF_TS(5)
  if (a.lab_ts && tid == 0) {
    const unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
    uint32_t* o = a.lab_ts + (size_t)blockIdx.x * 16;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = ts_[i];
    o[6] = (uint32_t)rt0_; o[7] = (uint32_t)rt1; o[8] = (uint32_t)b;
    o[9] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 4);           // HW_ID (wave, simd, cu, sh, se ...)
    o[10] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20);          // XCC_ID
  }
#endif
#undef SF_TS
}

#ifdef VV_LAB
// (lab build only: profiles/attic/score_fwd_pipelined.hip.txt -- not adopted)
// ---- k_score_fwd, PERSISTENT AND PIPELINED (round 5).  k_score_fwd's own phase stamps (profiles/r05_score_fwd_stamps.txt): a workgroup lives
// 11.6 us of which 8.9 us are the wait for its rows (index load -> 110 KB of rows) and 2.8 us its five compute phases; the 1024 workgroups
// run as two rounds of 512 that start together, so the chip alternates between everybody loading and everybody computing.  Here one
// workgroup per CU (the second register set keeps a second one out) walks its items -- the same item -> XCD ranges as item_of_block --
// with the NEXT item's rows (and the indices of the one after) requested before the current item's phases run.  What that takes:
//   * every request of the prefetch is UNCONDITIONAL (indices clamped to a valid row instead of `if (...) load`): hipcc's counted waits
//     (vmcnt is in order) can only leave requests in flight that are certain to have been issued -- behind a conditional one it waits for all;
//   * no global load inside an item's phases (the coefficients and the item's weight come with the prefetch);
//   * the phases' barriers are bare s_barrier behind s_waitcnt lgkmcnt(0): __syncthreads() is a workgroup-scope release, and behind a
//     global store hipcc makes that s_waitcnt vmcnt(0) -- the prefetch would be drained at the first barrier (nothing a phase stores to
//     global memory is read by another wave of the workgroup).
// Per item the arithmetic, its order and every output are k_score_fwd's: bit-identical results (tests/test_gpu_segbwd.py,
// test_gpu_fullsize.py, test_gpu_dedup.py run both).  No dropout form (k_score_fwd<.., true> stays one workgroup per item).
#define SFP_BAR() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int NW>
__device__ __forceinline__ float sfp_block_sum(float v, float* red) {
  v = wave_sum63(v);
  SFP_BAR();
  if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = v;
  SFP_BAR();
  float s = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += red[w];
  return s;
}
template <int NW>
__device__ __forceinline__ void sfp_block_sum3(float& x, float& y, float& z, float* red) {
  x = wave_sum63(x); y = wave_sum63(y); z = wave_sum63(z);
  SFP_BAR();
  if ((threadIdx.x & 63) == 63) { const int w = threadIdx.x >> 6; red[w] = x; red[NW + w] = y; red[2 * NW + w] = z; }
  SFP_BAR();
  x = y = z = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { x += red[w]; y += red[NW + w]; z += red[2 * NW + w]; }
}
template <int NW, int RPW, int DV>
__global__ __launch_bounds__(64 * NW) void k_score_fwd_p(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int THREADS = 64 * NW;
  constexpr int CV = 256 * DV / THREADS;
  static_assert(CV >= 1 && CV * THREADS == 256 * DV, "D must be a multiple of the thread count");
  constexpr int CXM = 6;
  const int D = a.D, C = a.C, Nn = a.Nn, CN = C + Nn;
  float* A = sm;               // [D]
  float* Ah = A + D;           // [D]
  float* acc0 = Ah + D;        // [NW][D] per-wave partial dAh
  float* n2 = acc0 + NW * D;   // [CN]
  float* tq = n2 + CN;         // [CN]
  float* cq = tq + CN;         // [CN]
  float* red = cq + CN;        // [3 NW]
  int* ooff = (int*)(red + 3 * NW);   // [CN] grouped position of channel ch's instance
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = (int)gridDim.x;
  const float eps = 1e-10f;
  // virtual block vb = blockIdx + j G (G a multiple of 8: the XCD of vb is this workgroup's) -> item, as item_of_block over B blocks
  auto item_of = [&](int vb) { return (a.items_rr || (a.B & 7)) ? vb : (vb & 7) * (a.B >> 3) + (vb >> 3); };
  struct Idx { int hr[RPW]; int hc[CXM]; int m, o; float wb; };
  struct Rows { float4 x[RPW][DV]; float cx[CXM][CV]; int seg, o; float wb; };   // (o, wb: carried over from the item's Idx)
  const int tch = tid < CN ? tid : CN - 1;                     // this thread's channel (the threads past CN repeat the last one: no branch)
  auto load_idx = [&](int b, Idx& I) {
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int qi = wave + NW * k;
      const int q = qi <= Nn ? qi : Nn;                        // (a slot past the last negative repeats it: its row is loaded and ignored)
      I.hr[k] = a.map[b * CN + (q == 0 ? 0 : C + q - 1)];
    }
#pragma unroll
    for (int j = 0; j < CXM; ++j) I.hc[j] = a.map[b * CN + (j + 1 < C ? j + 1 : C - 1)];
    I.m = a.map[b * CN + tch]; I.o = a.ord[b * CN + tch];
    I.wb = *(a.item_w ? a.item_w + b : a.loss_part);           // (no weights: any readable float; the value is not used)
  };
  auto load_rows = [&](const Idx& I, Rows& R) {
#pragma unroll
    for (int k = 0; k < RPW; ++k)
#pragma unroll
      for (int v = 0; v < DV; ++v) R.x[k][v] = *(const float4*)(a.H + (int64_t)I.hr[k] * D + lane * 4 + v * 256);
#pragma unroll
    for (int j = 0; j < CXM; ++j)
#pragma unroll
      for (int v = 0; v < CV; ++v) R.cx[j][v] = a.H[(int64_t)I.hc[j] * D + tid + v * THREADS];
    R.seg = a.seg_start[I.m];
    R.o = I.o; R.wb = I.wb;
  };
  float cf[CXM];
#pragma unroll
  for (int j = 0; j < CXM; ++j) cf[j] = j + 1 < C ? a.coeff[j] : 0.f;
  const float coeff_t = a.coeff[tid < C - 1 ? tid : 0];          // the context instance's coefficient of thread tid < C - 1

  // one item's phases: k_score_fwd's, statement for statement
  auto compute = [&](int b, const Rows& R) {
    if (tid < CN) ooff[tid] = R.seg + R.o;
    float ssq = 0.f;
#pragma unroll
    for (int v = 0; v < CV; ++v) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < CXM; ++j) s += cf[j] * R.cx[j][v];
      A[tid + v * THREADS] = s;
      ssq += s * s;
    }
    const float sA = sfp_block_sum<NW>(ssq, red);
    const float nA = sqrtf(sA) + eps;
    float* Vb = a.V + (int64_t)2 * b * D;
#pragma unroll
    for (int v = 0; v < CV; ++v) {
      const float ah = A[tid + v * THREADS] / nA;
      Ah[tid + v * THREADS] = ah;
      Vb[tid + v * THREADS] = ah;
    }
    SFP_BAR();
    float4 y[DV];
#pragma unroll
    for (int v = 0; v < DV; ++v) y[v] = *(const float4*)(Ah + lane * 4 + v * 256);
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int qi = wave + NW * k;
      float s = 0.f, t = 0.f;
#pragma unroll
      for (int v = 0; v < DV; ++v) {
        const float4 xx = R.x[k][v];
        s += xx.x * xx.x + xx.y * xx.y + xx.z * xx.z + xx.w * xx.w;
        t += xx.x * y[v].x + xx.y * y[v].y + xx.z * y[v].z + xx.w * y[v].w;
      }
      s = wave_sum63(s); t = wave_sum63(t);
      if (lane == 63 && qi <= Nn) { const int ch = qi == 0 ? 0 : C + qi - 1; n2[ch] = s; tq[ch] = t; }
    }
    SFP_BAR();
    const float sp = tq[0] / (sqrtf(n2[0]) + eps);
    const float wb = a.item_w ? R.wb : 1.f;
    float lsum = 0.f, vsum = 0.f, gsum = 0.f;
    for (int k = tid; k < Nn; k += THREADS) {
      const int ch = C + k;
      const float sn = tq[ch] / (sqrtf(n2[ch]) + eps);
      const float d = sp - sn;
      const float h = fmaxf(0.f, a.margin - d);
      float g;
      if (a.norm == 2) { lsum += wb * h * h; g = 2.f * wb * h * a.grad_scale; }
      else { lsum += wb * fabsf(h); g = h > 0.f ? wb * a.grad_scale : 0.f; }
      vsum += d < 0.f ? 1.f : 0.f;
      gsum += g;
      cq[ch] = g;
      if (a.s_bogus) a.s_bogus[(int64_t)b * Nn + k] = sn;
    }
    sfp_block_sum3<NW>(lsum, vsum, gsum, red);
    if (tid == 0) {
      cq[0] = -gsum;
      a.loss_part[b] = lsum;
      a.viol_part[b] = vsum;
      if (a.s_true) a.s_true[b] = sp;
      if (a.gate_host && b == 0) __hip_atomic_store(a.gate_host, a.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    SFP_BAR();
    float bnd = 0.f;
    float4 pa[DV];
#pragma unroll
    for (int v = 0; v < DV; ++v) pa[v] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RPW; ++k) {
      const int qi = wave + NW * k;
      if (qi > Nn) continue;                         // wave-uniform
      const int ch = qi == 0 ? 0 : C + qi - 1;
      const float c = cq[ch], s = n2[ch], t = tq[ch];
      const float rs = sqrtf(s);
      const float k3 = c * __builtin_amdgcn_rcpf(rs + eps);
      const float cd = c * __builtin_amdgcn_rcpf(s * rs + eps) * a.drop_scale * a.sg;
      bnd = fmaxf(bnd, fabsf(cd * s) + fabsf(cd * t) * rs);
      if (lane == 0) { SegRec rc; rc.alpha = cd * s; rc.beta = cd * t; rc.vec = 2 * b; rc.pad = b * CN + ch; a.rec[ooff[ch]] = rc; }
#pragma unroll
      for (int v = 0; v < DV; ++v) {
        pa[v].x += k3 * R.x[k][v].x; pa[v].y += k3 * R.x[k][v].y; pa[v].z += k3 * R.x[k][v].z; pa[v].w += k3 * R.x[k][v].w;
      }
    }
#pragma unroll
    for (int v = 0; v < DV; ++v) *(float4*)(acc0 + wave * D + lane * 4 + v * 256) = pa[v];
    if (tid < C - 1) {
      SegRec rc; rc.alpha = coeff_t * a.drop_scale * a.sg; rc.beta = 0.f; rc.vec = 2 * b + 1; rc.pad = b * CN + tid + 1;
      a.rec[ooff[tid + 1]] = rc;
    }
    if (lane == 0) red[wave] = bnd;
    SFP_BAR();
    float bnd_item = 0.f;
    if (tid == 0) {
#pragma unroll
      for (int w = 0; w < NW; ++w) bnd_item = fmaxf(bnd_item, red[w]);
    }
    float dot = 0.f;
    float u[CV];
#pragma unroll
    for (int v = 0; v < CV; ++v) {
      const int d = tid + v * THREADS;
      float us = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) us += acc0[w * D + d];
      u[v] = us;
      dot += A[d] * us;
    }
    dot = sfp_block_sum<NW>(dot, red);
    const float inv_denA = 1.f / (sA * sqrtf(sA) + eps);
#pragma unroll
    for (int v = 0; v < CV; ++v) {
      const int d = tid + v * THREADS;
      Vb[D + d] = (sA * u[v] - A[d] * dot) * inv_denA;
    }
    if (tid == 0 && a.bound_out) {
      float am = 0.f;
#pragma unroll
      for (int j = 0; j < CXM; ++j) am = fmaxf(am, fabsf(cf[j]));
      bnd_item = fmaxf(bnd_item, am * a.drop_scale * a.sg * 4.f * sA * gsum * inv_denA);
      atomicMax(a.bound_out + (b & (GG_BOUND_SLOTS - 1)) * GG_BOUND_STRIDE, ((unsigned long long)(unsigned)a.bound_seq << 32) | __float_as_uint(bnd_item));
    }
    SFP_BAR();                                       // the LDS areas are the next item's
  };

  const int me = (int)blockIdx.x;
  const int n_it = ((int)a.B - me + G - 1) / G;      // items of this workgroup: virtual blocks me, me + G, ...
  if (n_it <= 0) return;
  auto item = [&](int j) { return item_of(me + (j < n_it ? j : n_it - 1) * G); };   // (past the end: the last item again -- requests stay unconditional)
  Idx I;
  Rows R0, R1;
  load_idx(item(0), I);
  load_rows(I, R0);
  load_idx(item(1), I);
  int j = 0;
  for (; j + 2 < n_it; j += 2) {                     // two items per trip: the register sets swap roles by name, not by copy
    load_rows(I, R1);                                // item j + 1's rows fly beside item j's phases
    load_idx(item(j + 2), I);
    compute(item(j), R0);
    load_rows(I, R0);
    load_idx(item(j + 3), I);
    compute(item(j + 1), R1);
  }
  if (j + 1 < n_it) {                                // two items left
    load_rows(I, R1);
    compute(item(j), R0);
    compute(item(j + 1), R1);
  } else {
    compute(item(j), R0);
  }
}
#undef SFP_BAR


#endif  // VV_LAB
// ---- the same pass for items that do not fit in registers (D = 1024, hundreds of negatives: the per-GPU shape of BASELINE
// configs[4]): ONE sweep over the item's rows.  A wave takes negatives wave, wave + NW, ...; a row is in registers (DV float4
// per lane) while its norm and its dot with Ah are reduced inside the wave, and because every wave has computed the TARGET's
// score for itself first, the row's hinge coefficient g_k is known right there: its record is written and k3 x_k is added to
// the wave's partial dAh before the row is dropped.  Only the target's own coefficient (-sum of all g_k) has to wait for the
// block-wide sum; wave 0 adds that row at the end.  (A segment-wise form of the streaming k_score_loss read every row twice: 0.92 ms of the
// 2.26 ms cfg-5 step; this kernel 0.47 ms.)  Sums in a fixed order: a wave's rows in sequence, then the waves in sequence.
// (DV = 4 sits at the edge of 128 VGPRs: two workgroups per CU need 4 waves per SIMD -- the gradient bound's two registers
// pushed it to 131 and one workgroup per CU, 0.47 -> 0.70 ms at cfg 5; the second launch bound holds it at 128)
// V16 (round 6, ScoreArgs::v16): the two per-item vectors k_seg_bwd gathers once per instance -- Ah_b and dA_b -- leave as f16 (840 k gathers
// of a 4 KB row at the per-GPU shape of configs[4]: half the bytes).  Ah_b as it is (|values| <= 1); dA_b times a power of two that puts its
// largest magnitude in [2^13, 2^14), whose inverse rides in the context instances' alpha (exact) -- so their records are written behind dA_b.
template <int NW, int DV, bool H16 = false, bool V16 = false>
__global__ __launch_bounds__(64 * NW, DV == 4 ? 4 : 1) void k_score_stream(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int THREADS = 64 * NW;
  constexpr int CV = 256 * DV / THREADS;
  static_assert(CV >= 1 && CV * THREADS == 256 * DV, "D must be a multiple of the thread count");
  const int D = 256 * DV, C = a.C, Nn = a.Nn, CN = C + Nn;
  float* A = sm;               // [D]
  float* Ah = A + D;           // [D]
  float* acc0 = Ah + D;        // [NW][D] per-wave partial dAh
  float* red = acc0 + NW * D;  // [4 NW]: three groups for the loss sums, one for the waves' gradient bounds
  const int b = item_of_block(a.items_rr), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float eps = 1e-10f;
  const int32_t* map = a.map + (int64_t)b * CN;
  const int32_t* ord = a.ord + (int64_t)b * CN;

  // ---- context mean and its norm (column-parallel: a thread owns CV columns)
  float ssq = 0.f;
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    float sx = 0.f;
    for (int j = 1; j < C; ++j) sx += a.coeff[j - 1] * h_at<H16>(a.H, (int64_t)map[j] * D + d);
    A[d] = sx;
    ssq += sx * sx;
  }
  const float sA = block_sum_w<NW>(ssq, red);
  const float nA = sqrtf(sA) + eps;
  float* Vb = a.V + (int64_t)2 * b * D;
  _Float16* Vh = (_Float16*)a.V + (int64_t)2 * b * D;
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    const float ah = A[d] / nA;
    Ah[d] = ah;
    if (V16) Vh[d] = (_Float16)ah; else Vb[d] = ah;
  }
  __syncthreads();
  float4 y[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) y[v] = *(const float4*)(Ah + h_col<H16>(lane, v));

  auto load_row = [&](int ch, float4* x) { h_row<H16, DV>(a.H, map[ch], D, lane, x); };
  // norm^2 and dot with Ah of a row held in registers; both totals in every lane
  auto norm_dot = [&](const float4* x, float& s, float& t) {
    float ps = 0.f, pt = 0.f;
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      ps += x[v].x * x[v].x + x[v].y * x[v].y + x[v].z * x[v].z + x[v].w * x[v].w;
      pt += x[v].x * y[v].x + x[v].y * y[v].y + x[v].z * y[v].z + x[v].w * y[v].w;
    }
    ps = wave_sum63(ps); pt = wave_sum63(pt);
    s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ps), 63));
    t = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pt), 63));
  };

  // ---- the target's score, by every wave for itself
  float4 x0[DV];
  load_row(0, x0);
  float s0, t0;
  norm_dot(x0, s0, t0);
  const float sp = t0 / (sqrtf(s0) + eps);
  const float wb = a.item_w ? a.item_w[b] : 1.f;

  // ---- the wave's negatives: one sweep, the next row's loads issued before this row's reductions
  float4 pa[DV];
#pragma unroll
  for (int v = 0; v < DV; ++v) pa[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  float lsum = 0.f, vsum = 0.f, gsum = 0.f, bnd = 0.f;
  float4 xc[DV], xn[DV];
  int k = wave;
  if (k < Nn) load_row(C + k, xc);
  for (; k < Nn; k += NW) {
    const int ch = C + k;
    const bool more = k + NW < Nn;               // wave-uniform
    if (more) load_row(ch + NW, xn);
    float sq, tq;
    norm_dot(xc, sq, tq);
    const float rs = sqrtf(sq);
    const float sn = tq / (rs + eps);
    const float d = sp - sn;
    const float h = fmaxf(0.f, a.margin - d);
    float g;     // max_margin_loss_layer.cpp:82-97, 152-186 (as k_score_fwd)
    if (a.norm == 2) { lsum += wb * h * h; g = 2.f * wb * h * a.grad_scale; }
    else { lsum += wb * fabsf(h); g = h > 0.f ? wb * a.grad_scale : 0.f; }
    vsum += d < 0.f ? 1.f : 0.f;
    gsum += g;
    const float k3 = g * __builtin_amdgcn_rcpf(rs + eps);
    const float cd = g * __builtin_amdgcn_rcpf(sq * rs + eps) * a.drop_scale * a.sg;
    bnd = fmaxf(bnd, fabsf(cd * sq) + fabsf(cd * tq) * rs);          // |alpha| + |beta| |x|: GuardArgs::bound
    if (lane == 0) {
      SegRec rc; rc.alpha = cd * sq; rc.beta = cd * tq; rc.vec = 2 * b; rc.pad = b * CN + ch;
      a.rec[a.seg_start[map[ch]] + ord[ch]] = rc;
      if (a.s_bogus) a.s_bogus[(int64_t)b * Nn + k] = sn;
    }
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      pa[v].x += k3 * xc[v].x; pa[v].y += k3 * xc[v].y; pa[v].z += k3 * xc[v].z; pa[v].w += k3 * xc[v].w;
    }
    if (more) {
#pragma unroll
      for (int v = 0; v < DV; ++v) xc[v] = xn[v];
    }
  }
  // ---- block-wide loss / violations / sum of the coefficients (every lane of a wave holds the wave's values)
  __syncthreads();
  if (lane == 0) { red[wave] = lsum; red[NW + wave] = vsum; red[2 * NW + wave] = gsum; }
  __syncthreads();
  lsum = vsum = gsum = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) { lsum += red[w]; vsum += red[NW + w]; gsum += red[2 * NW + w]; }
  if (tid == 0) {
    a.loss_part[b] = lsum;
    a.viol_part[b] = vsum;
    if (a.s_true) a.s_true[b] = sp;
    if (a.gate_host && b == 0) __hip_atomic_store(a.gate_host, a.gate_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // ---- the target row: coefficient -sum_k g_k (wave 0 adds it to its partial)
  if (wave == 0) {
    const float c = -gsum, rs = sqrtf(s0);
    const float k3 = c * __builtin_amdgcn_rcpf(rs + eps);
    const float cd = c * __builtin_amdgcn_rcpf(s0 * rs + eps) * a.drop_scale * a.sg;
    bnd = fmaxf(bnd, fabsf(cd * s0) + fabsf(cd * t0) * rs);
    if (lane == 0) { SegRec rc; rc.alpha = cd * s0; rc.beta = cd * t0; rc.vec = 2 * b; rc.pad = b * CN; a.rec[a.seg_start[map[0]] + ord[0]] = rc; }
#pragma unroll
    for (int v = 0; v < DV; ++v) {
      pa[v].x += k3 * x0[v].x; pa[v].y += k3 * x0[v].y; pa[v].z += k3 * x0[v].z; pa[v].w += k3 * x0[v].w;
    }
  }
#pragma unroll
  for (int v = 0; v < DV; ++v) *(float4*)(acc0 + wave * D + h_col<H16>(lane, v)) = pa[v];
  if (!V16)
    for (int j = 1 + tid; j < C; j += THREADS) {
      SegRec rc; rc.alpha = a.coeff[j - 1] * a.drop_scale * a.sg; rc.beta = 0.f; rc.vec = 2 * b + 1; rc.pad = b * CN + j;
      a.rec[a.seg_start[map[j]] + ord[j]] = rc;
    }
  if (lane == 0) red[3 * NW + wave] = bnd;          // (a fourth group of NW words behind the three of the loss sums)
  __syncthreads();

  // ---- backward of the context normalisation: dA_b
  float bnd_item = 0.f;
  if (tid == 0) {
#pragma unroll
    for (int w = 0; w < NW; ++w) bnd_item = fmaxf(bnd_item, red[3 * NW + w]);
  }
  float dot = 0.f;
  float u[CV];
#pragma unroll
  for (int v = 0; v < CV; ++v) {
    const int d = tid + v * THREADS;
    float us = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) us += acc0[w * D + d];
    u[v] = us;
    dot += A[d] * us;
  }
  dot = block_sum_w<NW>(dot, red);
  const float inv_denA = 1.f / (sA * sqrtf(sA) + eps);
  if (V16) {
    float da[CV], m = 0.f;
#pragma unroll
    for (int v = 0; v < CV; ++v) { da[v] = (sA * u[v] - A[tid + v * THREADS] * dot) * inv_denA; m = fmaxf(m, fabsf(da[v])); }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2, 64));
    __syncthreads();                                // (block_sum_w's readers of red[0 .. NW) are done)
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
    float sc = 1.f;
    if (m > 0.f && m < 3.0e38f) { int e; (void)frexpf(m, &e); sc = ldexpf(1.f, 14 - e); }      // m sc in [2^13, 2^14)
#pragma unroll
    for (int v = 0; v < CV; ++v) Vh[D + tid + v * THREADS] = (_Float16)(da[v] * sc);
    const float isc = 1.f / sc;                     // (a power of two: alpha_j / sc times the stored row is alpha_j dA_b)
    for (int j = 1 + tid; j < C; j += THREADS) {
      SegRec rc; rc.alpha = a.coeff[j - 1] * a.drop_scale * a.sg * isc; rc.beta = 0.f; rc.vec = 2 * b + 1; rc.pad = b * CN + j;
      a.rec[a.seg_start[map[j]] + ord[j]] = rc;
    }
  } else {
#pragma unroll
    for (int v = 0; v < CV; ++v) {
      const int d = tid + v * THREADS;
      Vb[D + d] = (sA * u[v] - A[d] * dot) * inv_denA;
    }
  }
  if (tid == 0 && a.bound_out) {                    // as k_score_fwd
    float coeff_max = 0.f;       // max |coeff_j|: the context instances' gradient bound
    for (int j = 0; j + 1 < C; ++j) coeff_max = fmaxf(coeff_max, fabsf(a.coeff[j]));
    bnd_item = fmaxf(bnd_item, coeff_max * a.drop_scale * a.sg * 4.f * sA * gsum * inv_denA);
    atomicMax(a.bound_out + (b & (GG_BOUND_SLOTS - 1)) * GG_BOUND_STRIDE, ((unsigned long long)(unsigned)a.bound_seq << 32) | __float_as_uint(bnd_item));
  }
}

// the segment-wise pair: k_seg_bwd holds a row of D = 512 or 1024 columns; the forward is the register-resident
// k_score_fwd where an item fits (D = 512, up to 56 target / negative rows, 6 context rows), else the streaming kernel
bool score_fwd_supported(const ScoreArgs& a) { return a.D == 512 || a.D == 1024; }
// ... with dropout (ScoreArgs::drop): the register-resident kernel and k_seg_bwd's D = 512 form carry the per-instance masks
bool score_fwd_dropout_supported(int D, int C, int Nn) { return D == 512 && C - 1 <= 6 && 1 + Nn <= 56 && ko().score_stream != 1; }

// (KernelOpts::score_stream = 1: the one-sweep streaming kernel for every shape, A/B against k_score_fwd)
void launch_score_fwd(const ScoreArgs& a_in, hipStream_t s) {
  ScoreArgs a = a_in;
  a.items_rr = ko().score_rr;
  a.prefetch = ko().score_pf;
  const int rows = 1 + a.Nn;
  if (!(a.D == 512 && a.C - 1 <= 6 && rows <= 56) || ko().score_stream == 1) {
    const size_t lds = sizeof(float) * ((size_t)(2 + 8) * a.D + 4 * 8);
#define VV_SS(DV, H16)                                                                                     \
    do {                                                                                                  \
      (void)hipFuncSetAttribute((const void*)k_score_stream<8, DV, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      VV_LAUNCH((k_score_stream<8, DV, H16>), dim3(a.B), dim3(512), lds, s, a);                            \
    } while (0)
    if (a.D == 512) { if (a.h16) VV_SS(2, true); else VV_SS(2, false); }
    else if (a.h16 && a.v16) {
      (void)hipFuncSetAttribute((const void*)k_score_stream<8, 4, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      VV_LAUNCH((k_score_stream<8, 4, true, true>), dim3(a.B), dim3(512), lds, s, a);
    }
    else { if (a.h16) VV_SS(4, true); else VV_SS(4, false); }
#undef VV_SS
    return;
  }
  const size_t lds = sizeof(float) * ((size_t)(2 + 8) * a.D + 4 * (a.C + a.Nn) + 3 * 8) + 288;      // (+ 288: the prefetch's dead LDS-DMA target, ScoreArgs::prefetch)
#ifdef VV_LAB
  // the persistent, pipelined form: batches of at least two items per CU, no dropout (KernelOpts::score_pipe, VV_SCORE_PIPE=0: one workgroup per item)
  static int n_cu = 0;
  if (!n_cu) { int dev = 0; hipDeviceProp_t pr; (void)hipGetDevice(&dev); n_cu = hipGetDeviceProperties(&pr, dev) == hipSuccess ? pr.multiProcessorCount : 256; }
  const int G = n_cu & ~7;
  if (ko().lab_score_pipe && !a.drop.mode && G >= 8 && a.B >= 2 * G) {
#define VV_SFP(RPW)                                                                                       \
    do {                                                                                                  \
      (void)hipFuncSetAttribute((const void*)k_score_fwd_p<8, RPW, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      VV_LAUNCH((k_score_fwd_p<8, RPW, 2>), dim3(G), dim3(512), lds, s, a);                              \
    } while (0)
    if (rows <= 16) VV_SFP(2); else if (rows <= 32) VV_SFP(4); else VV_SFP(7);
#undef VV_SFP
    return;
  }
#endif
#define VV_SF1(RPW, DROP, H16)                                                                            \
  do {                                                                                                    \
    (void)hipFuncSetAttribute((const void*)k_score_fwd<8, RPW, 2, DROP, H16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    VV_LAUNCH((k_score_fwd<8, RPW, 2, DROP, H16>), dim3(a.B), dim3(512), lds, s, a);                       \
  } while (0)
#define VV_SF(RPW)                                                                                        \
  do {                                                                                                    \
    if (a.drop.mode) { if (a.h16) VV_SF1(RPW, true, true); else VV_SF1(RPW, true, false); }               \
    else { if (a.h16) VV_SF1(RPW, false, true); else VV_SF1(RPW, false, false); }                         \
  } while (0)
  if (rows <= 16) VV_SF(2);
  else if (rows <= 32) VV_SF(4);
  else VV_SF(7);
#undef VV_SF
#undef VV_SF1
}

// eight consecutive columns of a row of H (fp32: two 16-byte loads; f16, SegBwdArgs::h16: one)
template <bool H16> __device__ __forceinline__ void seg_row8(const float* H, int64_t i, float4& lo, float4& hi) {
  if constexpr (H16) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const h8 q = *(const h8*)((const _Float16*)H + i);
    lo = make_float4((float)q[0], (float)q[1], (float)q[2], (float)q[3]);
    hi = make_float4((float)q[4], (float)q[5], (float)q[6], (float)q[7]);
  } else {
    lo = *(const float4*)(H + i); hi = *(const float4*)(H + i + 4);
  }
}
// ---- segment-wise backward, pass 2: one wave per distinct row, a lane owns 8 consecutive columns (D = 512).
// Persistent grid: wave w of block g takes rows 4 g + w, + 4 SEGB_BLOCKS, ...; the column sums of the rows it produced
// (the bias gradient) leave as one partial row per block.
// The records of a row sit in ARRIVAL order (k_dd_map's atomic counter), which differs from run to run; the sums must
// not.  A segment of up to 64 records is put in instance order first (each lane holds one record, counts the smaller
// instance indices, and files its record at that rank in a wave-private LDS strip); longer segments -- a row repeated
// more than 64 times in one batch -- are summed in an order-independent way instead: every product is rounded to a
// multiple of 2^-36 in f64 ((p + M) - M, M = 1.5 * 2^16; exact for |p| < 2^15 in the gradient's scaled units), and
// sums of such multiples are exact in f64 up to 2^17, whatever the order.
// DROP (SegBwdArgs::drop, CH == 1): every instance carries its own dropout mask m_i over the shared row:
//   dx_u = [x_u > 0] (sum_i m_i alpha_i V_i - x_u scale sum_i m_i beta_i)        (alpha, beta already carry one factor scale)
// -- the mask regenerated per instance from its reference row (b = vec / 2, ch = pad - b CN), the beta sum per column.
template <typename T, int CH, bool DROP = false, bool H16 = false, bool V16 = false>
__global__ __launch_bounds__(256) void k_seg_bwd(SegBwdArgs a) {
  static_assert(!DROP || CH == 1, "dropout rides the D = 512 form");
  __shared__ float cs[4][512 * CH];
  __shared__ SegRec strip[4][64];
  float sgm;                                          // a repeat (guard round 1) scales the sums by a further 2^-k
  if (!gg_begin(a.guard, &cs[0][0], sgm)) return;
  const bool first = a.guard.round == 0;              // the bias partials come out of the unrounded values: once
  int pro_shift = 0;
  if (a.guard.gg && a.guard.proactive) {              // GuardArgs::proactive: the scale is settled BEFORE anything is rounded
    const float bm = gg_bound_fold(a.guard.bound, a.guard.seq);
    const float G = bm * (float)(*a.guard.cnt_max) * 1.01f;          // no element of any row's sum can pass this
    // ... and the scale is settled here outright, in both directions: the bound is put in [2^13, 2^14) (the true maximum
    // then sits the bound's looseness -- tens to hundreds -- below: well inside f16's normal range), whatever the host's sg
    int e = 14;
    if (G > 0.f && G < 3.0e38f) (void)frexpf(G, &e);                 // G < 2^e  (zero gradient / inf / nan: no shift; gg_end's flag reports)
    pro_shift = e - 14 > 60 ? 60 : (e - 14 < -60 ? -60 : e - 14);
    sgm = ldexpf(1.f, -pro_shift);                                   // (first needed where the first row is rounded: the two loads above fly meanwhile)
  }
  const int U = a.info[0];
  const int Uk = min((U + BK - 1) / BK * BK, a.Rp);   // the wgrad K loop reads whole BK-row steps
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c0 = lane * 8, D = a.D;                   // D = 512 CH; chunk c covers columns 512 c + c0 .. + 7
  float col[CH][8];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) col[c][j] = 0.f;
  float gmx = 0.f;
  // A wave's rows are a chain of dependent reads each: segment bounds, records, the vectors the records name, and the row
  // itself.  The row's own values (needed last) and the NEXT row's segment bounds are requested right behind the records,
  // so that only records -> vectors is exposed: 18.7 -> 17.9 us.  (More waves do not help: 89 registers and five blocks per
  // CU instead of four measured the same 17.8-18.3 us; so did requesting the NEXT row's records one row early, bounds two rows
  // early -- 17.8 us: the kernel moves 42 MB of fp32 rows in and 21 MB out, ~3.5 TB/s, and is not latency-chained any more.)
  const int u_first = blockIdx.x * 4 + wave, u_step = 4 * (int)gridDim.x;     // (block numbers by XCD ranges, as item_of_block: measured, no change)
  int seg_b = 0, seg_e = 0;
  if (u_first < U) { seg_b = a.seg_start[u_first]; seg_e = a.seg_start[u_first + 1]; }
  for (int u = u_first; u < Uk; u += u_step) {
    if (u >= U) {
#pragma unroll
      for (int c = 0; c < CH; ++c) *(uint4*)(a.dYu + (int64_t)u * a.Dp + 512 * c + c0) = make_uint4(0u, 0u, 0u, 0u);
      continue;
    }
    const int b = seg_b, e = seg_e, n = e - b;
    SegRec mine; mine.alpha = 0.f; mine.beta = 0.f; mine.vec = 0; mine.pad = 0x7fffffff;
    if (n > 1 && n <= 64 && lane < n) mine = a.rec[b + lane];
    float4 xr0[CH], xr1[CH];
    if (CH == 1) {                                    // (D = 1024 with its long segments: the early row costs more in registers than it hides -- 0.437 -> 0.473 ms at cfg 5)
#pragma unroll
      for (int c = 0; c < CH; ++c) seg_row8<H16>(a.H, (int64_t)u * D + 512 * c + c0, xr0[c], xr1[c]);      // (as non-temporal loads: no gain, profiles/r03_step_ablations.txt 5d)
    }
    if (u + u_step < U) { seg_b = a.seg_start[u + u_step]; seg_e = a.seg_start[u + u_step + 1]; }
    float acc[CH][8];
    float bs = 0.f;
    float bsv[DROP ? 8 : 1];                          // DROP: sum_i m_i beta_i per column
    if (DROP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) bsv[j & (DROP ? 7 : 0)] = 0.f;
    }
    // the instance's keep bits for this lane's eight columns
    auto keep8 = [&](const SegRec& r) -> uint32_t {
      const int bb = r.vec >> 1, ch = r.pad - bb * a.drop.CN;
      const int64_t rr = (int64_t)ch * a.drop.B + bb;
      const uint32_t rc = drop_row_ctr(rr, D, a.drop.s32);
      return drop_keep4(a.drop, rr, rc, c0) | (drop_keep4(a.drop, rr, rc, c0 + 4) << 4);
    };
    if (n <= 64) {
      const SegRec* rs = a.rec + b;
      if (n > 1) {                                    // instance order
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += __builtin_amdgcn_readlane(mine.pad, j) < mine.pad;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the previous row's reads of the strip are done
        if (lane < n) strip[wave][rank] = mine;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rs = strip[wave];
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
      constexpr int UN = CH == 1 ? 4 : 2;             // instances in flight
      int i = 0;
      for (; i + UN - 1 < n; i += UN) {
        SegRec r[UN]; float4 v0[UN][CH], v1[UN][CH];
#pragma unroll
        for (int k = 0; k < UN; ++k) r[k] = rs[i + k];
#pragma unroll
        for (int k = 0; k < UN; ++k)
#pragma unroll
          for (int c = 0; c < CH; ++c) seg_row8<V16>(a.V, (int64_t)r[k].vec * D + 512 * c + c0, v0[k][c], v1[k][c]);
#pragma unroll
        for (int k = 0; k < UN; ++k) {
          const float al = r[k].alpha;
          if (DROP) {
            const uint32_t kp = keep8(r[k]);
            const float vv[8] = {v0[k][0].x, v0[k][0].y, v0[k][0].z, v0[k][0].w, v1[k][0].x, v1[k][0].y, v1[k][0].z, v1[k][0].w};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const bool kj = (kp >> j) & 1u;
              acc[0][j] += kj ? al * vv[j] : 0.f;
              bsv[j & (DROP ? 7 : 0)] += kj ? r[k].beta : 0.f;
            }
            continue;
          }
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            acc[c][0] += al * v0[k][c].x; acc[c][1] += al * v0[k][c].y; acc[c][2] += al * v0[k][c].z; acc[c][3] += al * v0[k][c].w;
            acc[c][4] += al * v1[k][c].x; acc[c][5] += al * v1[k][c].y; acc[c][6] += al * v1[k][c].z; acc[c][7] += al * v1[k][c].w;
          }
          bs += r[k].beta;
        }
      }
      for (; i < n; ++i) {
        const SegRec r = rs[i];
        if (DROP) {
          const float* vp = a.V + (int64_t)r.vec * D + c0;
          const float4 v0 = *(const float4*)vp, v1 = *(const float4*)(vp + 4);
          const uint32_t kp = keep8(r);
          const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const bool kj = (kp >> j) & 1u;
            acc[0][j] += kj ? r.alpha * vv[j] : 0.f;
            bsv[j & (DROP ? 7 : 0)] += kj ? r.beta : 0.f;
          }
          continue;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          float4 v0, v1;
          seg_row8<V16>(a.V, (int64_t)r.vec * D + 512 * c + c0, v0, v1);
          acc[c][0] += r.alpha * v0.x; acc[c][1] += r.alpha * v0.y; acc[c][2] += r.alpha * v0.z; acc[c][3] += r.alpha * v0.w;
          acc[c][4] += r.alpha * v1.x; acc[c][5] += r.alpha * v1.y; acc[c][6] += r.alpha * v1.z; acc[c][7] += r.alpha * v1.w;
        }
        bs += r.beta;
      }
    } else {                                          // order-independent sums
      const double M = 98304.0;
      double dacc[CH][8], dbs = 0.0;
      double dbsv[DROP ? 8 : 1];
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) dacc[c][j] = 0.0;
      if (DROP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) dbsv[j & (DROP ? 7 : 0)] = 0.0;
      }
      for (int i = b; i < e; ++i) {
        const SegRec r = a.rec[i];
        const double al = (double)r.alpha;
        const uint32_t kp = DROP ? keep8(r) : 0xffu;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          float4 v0, v1;
          seg_row8<V16>(a.V, (int64_t)r.vec * D + 512 * c + c0, v0, v1);
          const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) dacc[c][j] += (!DROP || ((kp >> j) & 1u)) ? (al * (double)vv[j] + M) - M : 0.0;
        }
        const double rb = ((double)r.beta + M) - M;
        dbs += rb;
        if (DROP) {
#pragma unroll
          for (int j = 0; j < 8; ++j) dbsv[j & (DROP ? 7 : 0)] += ((kp >> j) & 1u) ? rb : 0.0;
        }
      }
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = (float)dacc[c][j];
      bs = (float)dbs;
      if (DROP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bsv[j & (DROP ? 7 : 0)] = (float)dbsv[j & (DROP ? 7 : 0)];
      }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (CH != 1) seg_row8<H16>(a.H, (int64_t)u * D + 512 * c + c0, xr0[c], xr1[c]);
      const float4 x0 = xr0[c], x1 = xr1[c];
      const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
      float g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        g[j] = xv[j] > 0.f ? (acc[c][j] - (DROP ? bsv[j & (DROP ? 7 : 0)] * a.drop.scale : bs) * xv[j]) * sgm : 0.f;
        col[c][j] += g[j];
        gmx = fmaxf(gmx, fabsf(g[j]));
      }
      uint32_t o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = T::from_float(g[2 * j]) | ((uint32_t)T::from_float(g[2 * j + 1]) << 16);
      *(uint4*)(a.dYu + (int64_t)u * a.Dp + 512 * c + c0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
  }
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < 8; ++j) cs[wave][512 * c + c0 + j] = col[c][j];
  __syncthreads();
  if (first)
    for (int d = threadIdx.x; d < 512 * CH; d += 256)
        a.dbp[(int64_t)blockIdx.x * D + d] = (cs[0][d] + cs[1][d] + cs[2][d] + cs[3][d]) * (a.inv_sg / sgm);   // (sgm: a power of two, exact)
  if (a.guard.gg) {
    if (a.guard.proactive && blockIdx.x == 0 && threadIdx.x == 0) { a.guard.gg->shift[3] = pro_shift; a.guard.gg->mul = sgm; }   // for k_reduce
    __syncthreads();
    gg_end(a.guard, gg_block_max(gmx, &cs[0][0]));
  }
}

void launch_seg_bwd(int prec, const SegBwdArgs& a, hipStream_t s) {
  const int ch = a.D / 512;
  // (a conditional repeat of the gradient-scale guard normally returns at once: a quarter of the grid starts faster)
  const int nb = a.guard.round == 0 ? SEGB_BLOCKS : SEGB_BLOCKS / 4;
#define VV_SB(T, CH, DROP) do { if (a.h16) VV_LAUNCH((k_seg_bwd<T, CH, DROP, true>), dim3(nb), dim3(256), 0, s, a); \
                                 else VV_LAUNCH((k_seg_bwd<T, CH, DROP, false>), dim3(nb), dim3(256), 0, s, a); } while (0)
  if (a.drop.mode && ch == 1) {
    if (prec == 0) VV_SB(F16, 1, true); else VV_SB(BF16, 1, true);
    return;
  }
  if (ch == 2 && a.h16 && a.v16) {               // (the one-sweep score kernel's f16 vectors: SegBwdArgs::v16)
    if (prec == 0) VV_LAUNCH((k_seg_bwd<F16, 2, false, true, true>), dim3(nb), dim3(256), 0, s, a);
    else VV_LAUNCH((k_seg_bwd<BF16, 2, false, true, true>), dim3(nb), dim3(256), 0, s, a);
    return;
  }
  if (prec == 0) { if (ch == 1) VV_SB(F16, 1, false); else VV_SB(F16, 2, false); }
  else { if (ch == 1) VV_SB(BF16, 1, false); else VV_SB(BF16, 2, false); }
#undef VV_SB
}


void launch_score_loss(int prec, const ScoreArgs& a_in, hipStream_t s) {
  ScoreArgs a = a_in;
  a.items_rr = 1;          // per-instance rows (dense execution): an item's rows are contiguous already; XCD ranges measured + 1 us (42.8 against 41.8)
  if (ko().score_reg && (prec == 0 ? launch_score_loss_reg<F16>(a, s) : launch_score_loss_reg<BF16>(a, s))) return;
  const bool vec = a.D % 4 == 0;
  const bool wide = vec && a.D >= 2048 && a.B <= 512;       // few items of wide rows: sixteen waves per item
  // the two [G][D] partial-sum areas hold G D <= 4 NT floats (with the 1024-thread form and D = 2048 they ran over at a fixed 1024)
  const int nt = wide ? 1024 : SL_THREADS;
  const size_t lds = sizeof(float) * ((size_t)2 * a.D + 2 * (size_t)(a.D > 4 * nt ? a.D : 4 * nt) + 8 * (a.C + a.Nn) + 16);
  const dim3 grid(a.B), block(wide ? 1024 : SL_THREADS);
#define VV_SL(T, V, NT)                                                                          \
  do {                                                                                           \
    (void)hipFuncSetAttribute((const void*)k_score_loss<T, V, NT>,                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
    VV_LAUNCH((k_score_loss<T, V, NT>), grid, block, lds, s, a);                        \
  } while (0)
  if (prec == 0) { if (wide) VV_SL(F16, true, 1024); else if (vec) VV_SL(F16, true, SL_THREADS); else VV_SL(F16, false, SL_THREADS); }
  else { if (wide) VV_SL(BF16, true, 1024); else if (vec) VV_SL(BF16, true, SL_THREADS); else VV_SL(BF16, false, SL_THREADS); }
#undef VV_SL
}

// loss = scale * sum(loss_part), violations = sum(viol_part); fixed-order (deterministic)
__global__ void k_final_loss(const float* lp, const float* vp, int B, float scale, float* out2) {
  __shared__ float red[8];
  float l = 0.f, v = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) { l += lp[i]; v += vp[i]; }
  l = block_sum(l, red);
  v = block_sum(v, red + 4);
  if (threadIdx.x == 0) { out2[0] = l * scale; out2[1] = v; }
}
void launch_final_loss(const float* lp, const float* vp, int B, float scale, float* out2,
                       hipStream_t s) {
  hipLaunchKernelGGL(k_final_loss, dim3(1), dim3(256), 0, s, lp, vp, B, scale, out2);
}

// ------------------------------------------------------------------------------- reduce -------
// Blocks [0, nblk_dw) sum the split-K slabs into dW (grid-stride, 16-B accesses); the remaining
// blocks each own 64 bias columns and sum the per-item partials in a fixed order.
constexpr int RED_DW_BLOCKS = 2048;

// 16 bias columns per block, 16 row groups, 8 independent loads in flight per thread; the
// partial sums are combined in a fixed order (bit-reproducible, no atomics)
__device__ __forceinline__ void reduce_db(const ReduceArgs& a, int blk) {
  __shared__ float part[16][16];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int d = blk * 16 + tx;
  float s = 0.f;
  if (d < a.D) {
    const float* p = a.dbp + d;
    const int nb = a.db_rows > 0 ? a.db_rows : a.B;
    int b = ty;
    for (; b + 112 < nb; b += 128) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + 16 * u) * a.D];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; b < nb; b += 16) s += p[(int64_t)b * a.D];
  }
  part[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && d < a.D) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += part[u][tx];
    if (a.shard_rows > 0) a.grads[(int64_t)(d / a.shard_rows) * ((int64_t)a.shard_rows * a.F + a.shard_rows) + (int64_t)a.shard_rows * a.F + d % a.shard_rows] = t;
    else a.grads[(int64_t)a.D * a.F + d] = t;
  }
}

// loss = scale * sum(loss_part), violations = sum(viol_part); fixed-order (deterministic)
__device__ __forceinline__ void reduce_loss(const ReduceArgs& a) {
  __shared__ float red[8];
  float l = 0.f, v = 0.f;
  for (int i = threadIdx.x; i < a.B; i += 256) { l += a.loss_part[i]; v += a.viol_part[i]; }
  l = block_sum(l, red);
  v = block_sum(v, red + 4);
  if (threadIdx.x == 0) { a.loss_out[0] = l * a.loss_scale; a.loss_out[1] = v; }
  if (a.gmax_host) {
    // f16 gradient-scale guard: the step's largest |dY| (unscaled: the normal pass's per-block maxima over the host's sg)
    // and the shift the repeats applied go to a host-visible ring; the host reads entry seq - 4 when it issues step seq
    __shared__ float gsm[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < a.gmax_n0; i += 256) m = fmaxf(m, a.gmax_slots[i]);
    for (int i = threadIdx.x; i < a.gmax_n1; i += 256) m = fmaxf(m, a.gmax_slots[(size_t)a.gmax_stride + i]);
    m = gg_block_max(m, gsm);
    const float gbm = a.gbound ? gg_bound_fold(a.gbound, a.seq) : 0.f;
    if (threadIdx.x == 0) {
      // word 1: bits 0-15 the shift taken off on the device, bit 16 a conditional repeat ran, bit 30 a value passed the limit
      // in the step's FINAL round (cannot happen with the repeats / the proactive bound: the host treats it as an internal
      // error); high half: the proactive bound cnt_max x bound over the host's sg (0 on the other paths)
      unsigned fl = 0; float G = 0.f;
      if (a.gg) {
        fl = (unsigned)(a.gg->shift[3] & 0xFFFF) | (a.gg->repeat_seq == a.seq ? (1u << 16) : 0u) |
             (a.gg->flag[a.guard_last_round] == a.seq ? (1u << 30) : 0u);
        if (a.gbound) G = gbm * (float)(*a.gcnt) / a.sg;
      }
      unsigned long long* e = a.gmax_host + 2 * (a.seq & 15);
      __hip_atomic_store(e + 1, ((unsigned long long)__float_as_uint(G) << 32) | fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(e, ((unsigned long long)__float_as_uint(m / a.sg) << 32) | (unsigned)a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// next W->half scale from the running max: f16 keeps max|W|*sw in [2^11, 2^12); bf16 needs none.  One 256-thread workgroup.
__device__ __forceinline__ void scale_update_body(Scales* sc, const float* wmax_blocks, int nblocks, int prec) {
  __shared__ float red[8];
  float mm = 0.f;
  if (wmax_blocks)
    for (int i = threadIdx.x; i < nblocks; i += 256) mm = fmaxf(mm, wmax_blocks[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mm;
  __syncthreads();
  if (threadIdx.x != 0) return;
  mm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float m = fmaxf(mm, __uint_as_float(sc->wmax_bits));
  float sw = 1.f;
  if (prec == 0 && m > 0.f && isfinite(m)) {
    int e;
    frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
    sw = ldexpf(1.f, 12 - e);      // m * sw in [2^11, 2^12)
  }
  sc->sw_next = sw;
  sc->wmax_bits = 0u;
}

template <bool VEC, bool S16 = false>
__global__ __launch_bounds__(256) void k_reduce(ReduceArgs a) {
  // The few workgroups with serial work (bias columns: a loop over the per-block partials; loss; scale) are numbered last
  // but DISPATCHED first: the 2048 dW workgroups fill every CU's block slots, and a workgroup that only starts when the
  // first of them retires adds its own loop to the kernel's length (measured: 16.7 us with them last).
  const int n_special = (a.scale_sc ? 1 : 0) + ((a.parts & 2) ? (a.D + 15) / 16 + 1 : 0);
  int bid = (int)blockIdx.x < n_special ? (int)gridDim.x - n_special + (int)blockIdx.x : (int)blockIdx.x - n_special;
  // a pending W -> half scale update of the PREVIOUS step's SGD kernel rides as the last workgroup (its own one-workgroup
  // launch cost ~5 us of stream time a step: the kernel plus two dependent-launch gaps); this step's k_sgd reads the result
  const int G = (int)gridDim.x - (a.scale_sc ? 1 : 0);
  if (bid == G) { scale_update_body(a.scale_sc, a.scale_wmax, a.scale_n, a.scale_prec); return; }
  if (a.parts & 2) {
    if (bid == G - 1) { reduce_loss(a); return; }
    const int ndb = (a.D + 15) / 16;
    if (bid >= G - 1 - ndb) { reduce_db(a, bid - (G - 1 - ndb)); return; }
  }
  if (!(a.parts & 1)) return;
  const float sgf = a.sg_dev ? *a.sg_dev : (a.gg ? a.sg * a.gg->mul : a.sg);   // the scale the 16-bit gradients really carry
  const float inv = a.ip_scale / (sgf * a.scales->sx);
  const int64_t slab_sz = slab_pitch(a.Dp, a.Fp);
  const int d0 = a.d_begin, dn = a.d_count > 0 ? a.d_count : a.D;
  if (VEC) {
    const int fb = a.f_count > 0 ? a.f_begin : 0, fn = a.f_count > 0 ? a.f_count : a.F;
    const int f4 = fn / 4;
    const int nblk = (int)gridDim.x - (a.scale_sc ? 1 : 0) - ((a.parts & 2) ? (a.D + 15) / 16 + 1 : 0);      // this launch's dW blocks
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < (int64_t)dn * f4;
         i += (int64_t)nblk * 256) {
      const int d = d0 + (int)(i / f4), f = fb + (int)(i % f4) * 4;
      const float* p = a.slabs + (int64_t)d * a.Fp + f;
      if constexpr (S16) {             // f16 partial products: every split's four values times the inverse factor of its tile, summed in slab order
        const int64_t po = (int64_t)d * a.Fp + f;
        const int tiles = (a.Dp >> 8) * (a.Fp >> 8), tl = slab_tile(d, f, a.Fp);
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = u < a.S ? slab_ld4<true>(a.slabs, u * slab_sz + po, a.slab_sc[u * tiles + tl]) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s = t[0];
#pragma unroll
        for (int u = 1; u < 8; ++u)
          if (u < a.S) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
        int64_t o = (int64_t)d * a.F + f;
        if (a.n_chunks > 0) {
          int cc = 0;
          while (cc + 1 < a.n_chunks && f >= a.chunk_c0[cc + 1]) ++cc;
          const int c0 = a.chunk_c0[cc], c1 = a.chunk_c0[cc + 1];
          o = (int64_t)a.D * c0 + (int64_t)d * (c1 - c0) + (f - c0);
        }
        if (a.shard_rows > 0) o += (int64_t)(d / a.shard_rows) * a.shard_rows;
        *(float4*)(a.grads + o) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
        continue;
      }
      float4 s = nt_load4(p);
      int k = 1;
      if (a.S == 8) {                  // the usual split: all eight loads in flight at once, summed in slab order
        float4 t[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) t[u] = nt_load4(p + (u + 1) * slab_sz);
#pragma unroll
        for (int u = 0; u < 7; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
        k = 8;
      }
      for (; k + 3 < a.S; k += 4) {
        const float4 t0 = *(const float4*)(p + (k + 0) * slab_sz), t1 = *(const float4*)(p + (k + 1) * slab_sz);
        const float4 t2 = *(const float4*)(p + (k + 2) * slab_sz), t3 = *(const float4*)(p + (k + 3) * slab_sz);
        s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
        s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
        s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
        s.x += t3.x; s.y += t3.y; s.z += t3.z; s.w += t3.w;
      }
      for (; k < a.S; ++k) {
        const float4 t = *(const float4*)(p + k * slab_sz);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      int64_t o = (int64_t)d * a.F + f;
      if (a.n_chunks > 0) {            // chunk-major (vv_internal.h: ReduceArgs::n_chunks)
        int cc = 0;
        while (cc + 1 < a.n_chunks && f >= a.chunk_c0[cc + 1]) ++cc;
        const int c0 = a.chunk_c0[cc], c1 = a.chunk_c0[cc + 1];
        o = (int64_t)a.D * c0 + (int64_t)d * (c1 - c0) + (f - c0);
      }
      if (a.shard_rows > 0) o += (int64_t)(d / a.shard_rows) * a.shard_rows;       // shard-major: every shard carries its db entries behind its rows
      *(float4*)(a.grads + o) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    }
  } else {
    const int64_t nW = (int64_t)dn * a.F;
    for (int64_t i = (int64_t)bid * 256 + threadIdx.x; i < nW; i += (int64_t)RED_DW_BLOCKS * 256) {
      const int d = d0 + (int)(i / a.F), f = (int)(i % a.F);
      float s = 0.f;
      for (int k = 0; k < a.S; ++k) s += a.slabs[k * slab_sz + (int64_t)d * a.Fp + f];
      a.grads[(int64_t)d * a.F + f] = s * inv;
    }
  }
}
void launch_reduce(const ReduceArgs& a, hipStream_t s) {
  // dW blocks (fewer for a launch that reduces a column range only), db blocks, the loss block, the scale block
  const int dwb = !(a.parts & 1) ? 0 : (a.f_count > 0 && a.F % 4 == 0 ? std::max(64, (int)((int64_t)RED_DW_BLOCKS * a.f_count / a.F)) : RED_DW_BLOCKS);
  const dim3 grid(dwb + ((a.parts & 2) ? (a.D + 15) / 16 + 1 : 0) + (a.scale_sc ? 1 : 0));
  if (a.F % 4 == 0 && a.slab16) VV_LAUNCH((k_reduce<true, true>), grid, dim3(256), 0, s, a);     // (slab16 needs F % 8 == 0 and S <= 8: api.hip)
  else if (a.F % 4 == 0) VV_LAUNCH(k_reduce<true>, grid, dim3(256), 0, s, a);
  else VV_LAUNCH(k_reduce<false>, grid, dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------- SGD ----------
// solver.cpp:502-531: g += local_decay * w (L2) | sign(w) (L1); h = local_rate * g + momentum * h;
// blob.cpp:118-120: w -= h.  Also writes the scaled half copy used by the next forward and tracks
// max |w| for the f16 range guard.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void k_sgd(SgdArgs a) {
  if (a.skip_if && __hip_atomic_load(a.skip_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;   // (a failed exchange in front: SgdArgs::skip_if)
  const float sw = a.scales->sw_next;
  const int64_t nW = (int64_t)a.D * a.F;
  const float lr_w = a.rate * a.lr_mult_w, dc_w = a.weight_decay * a.decay_mult_w;
  float wmax = 0.f;
  // one parameter element: regulariser, then SGD (solver.cpp:502-531), Nesterov (:599-655) or AdaGrad (:714-781)
  auto rule = [&](float w, float g, float& h, float lr, float dc) {
    if (dc != 0.f) g += dc * (a.reg == 2 ? w : (float)((w > 0.f) - (w < 0.f)));
    float u;
    if (a.solver_type == 1) { const float h0 = h; h = lr * g + a.momentum * h0; u = (1.f + a.momentum) * h - a.momentum * h0; }
    else if (a.solver_type == 2) { h += g * g; u = lr * (g / (sqrtf(h) + a.delta)); }
    else { h = lr * g + a.momentum * h; u = h; }
    return w - u;
  };
  auto upd = [&](float w, float g, float& h) {
    w = rule(w, g, h, lr_w, dc_w);
    wmax = fmaxf(wmax, fabsf(w));
    return w;
  };
  if (VEC) {       // F % 4 == 0: 16-B accesses, one row segment per thread
    const int fc = a.chunked ? a.f_count : a.F;              // this launch's columns: one F-chunk, or all
    const int f4 = fc / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)a.D * f4;
         i += (int64_t)gridDim.x * 256) {
      const int d = (int)(i / f4), fl = (int)(i % f4) * 4, f = a.f_begin + fl;
      const int64_t o = (int64_t)d * a.F + f;
      const int64_t og = a.chunked ? (int64_t)a.D * a.f_begin + (int64_t)d * fc + fl : o;     // chunk-major gradient buffer
      float4 w = nt_load4(a.W + o), h = nt_load4(a.hW + o);
      const float4 g = nt_load4(a.grads + og);
      w.x = upd(w.x, g.x, h.x); w.y = upd(w.y, g.y, h.y); w.z = upd(w.z, g.z, h.z); w.w = upd(w.w, g.w, h.w);
      nt_store4(a.W + o, w);
      nt_store4(a.hW + o, h);
      const uint32_t lo = T::from_float(w.x * sw) | ((uint32_t)T::from_float(w.y * sw) << 16);
      const uint32_t hi = T::from_float(w.z * sw) | ((uint32_t)T::from_float(w.w * sw) << 16);
      if (a.pub_flag) __hip_atomic_store((unsigned long long*)(a.Wh + (int64_t)d * a.Fp + f), ((unsigned long long)hi << 32) | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *(uint2*)(a.Wh + (int64_t)d * a.Fp + f) = make_uint2(lo, hi);
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nW; i += (int64_t)gridDim.x * 256) {
      float h = a.hW[i];
      const float w = upd(a.W[i], a.grads[i], h);
      a.hW[i] = h;
      a.W[i] = w;
      const int d = (int)(i / a.F), f = (int)(i % a.F);
      a.Wh[(int64_t)d * a.Fp + f] = T::from_float(w * sw);
    }
  }
  const float lr_b = a.rate * a.lr_mult_b, dc_b = a.weight_decay * a.decay_mult_b;
  for (int d = blockIdx.x * 256 + threadIdx.x; a.do_bias && d < a.D; d += gridDim.x * 256) {
    float h = a.hb[d];
    const float bn = rule(a.b[d], a.grads[nW + d], h, lr_b, dc_b);
    if (a.pub_flag) __hip_atomic_store(a.b + d, bn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else a.b[d] = bn;
    a.hb[d] = h;
  }
  // per-block max |w| -> one slot per block (no atomics: thousands of adds on one address
  // serialise at ~12 ns each); k_scale_update folds the slots.  (Folding them in this kernel's last workgroup instead --
  // an arrival counter -- was measured: the 1024 counter adds made the kernel 10 us longer, the saved launch is ~3 us.)
  __shared__ float wm[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = wmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    // (published launches: at agent scope like everything else a kernel of the COMPUTE stream reads behind the gates -- the scale
    // workgroup of the next k_reduce folds these slots and is ordered behind this kernel by the gate flag only, not by its end)
    const float wmb = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (a.pub_flag) __hip_atomic_store(a.wmax_blocks + a.blk_off + blockIdx.x, wmb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else a.wmax_blocks[a.blk_off + blockIdx.x] = wmb;
    if (blockIdx.x == 0 && a.set_scale) {
      if (a.pub_flag) __hip_atomic_store(&a.scales->sw_cur, sw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else a.scales->sw_cur = sw;
    }
  }
  if (a.pub_flag) {                    // SgdArgs::pub_flag: every wave's stores have landed before the workgroup counts itself in
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const int old = __hip_atomic_fetch_add(a.pub_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == (int)gridDim.x - 1) {
        __hip_atomic_store(a.pub_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.pub_flag, a.pub_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__global__ void k_publish(int32_t* flag, int32_t seq) { __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// test hook (VV_COMM_TEST_DELAY_US): holds a stream for `us` microseconds, so that the gated forward GEMM really waits
__global__ void k_delay(int us) {
  const uint64_t t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(16);
}
void launch_delay(int us, hipStream_t s) { hipLaunchKernelGGL(k_delay, dim3(1), dim3(1), 0, s, us); }
void launch_publish(int32_t* flag, int32_t seq, hipStream_t s) { hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, s, flag, seq); }
void launch_sgd(int prec, const SgdArgs& a, hipStream_t s) {
  const bool vec = a.F % 4 == 0;
  const dim3 grid(a.chunked ? a.n_blk : SGD_BLOCKS), block(256);
  if (prec == 0) { if (vec) VV_LAUNCH((k_sgd<F16, true>), grid, block, 0, s, a); else VV_LAUNCH((k_sgd<F16, false>), grid, block, 0, s, a); }
  else { if (vec) VV_LAUNCH((k_sgd<BF16, true>), grid, block, 0, s, a); else VV_LAUNCH((k_sgd<BF16, false>), grid, block, 0, s, a); }
}

__global__ void k_scale_update(Scales* sc, const float* wmax_blocks, int nblocks, int prec) { scale_update_body(sc, wmax_blocks, nblocks, prec); }
void launch_scale_update(int prec, Scales* sc, const float* wmax_blocks, int n_blocks, hipStream_t s) {
  hipLaunchKernelGGL(k_scale_update, dim3(1), dim3(256), 0, s, sc, wmax_blocks, n_blocks, prec);
}

// ------------------------------------------------------------------------------- reduction + update in one launch ----
// k_reduce_sgd = k_reduce followed by k_sgd, element by element: a thread sums its 16 bytes of the eight slabs, scales them
// (the gradient, stored for whoever asks: non-temporal), and applies the solver's rule to the same 16 bytes of W right
// there -- the gradient is not written and read back between two launches, and one dependent launch boundary goes.  Same
// sums in the same order, the same rule(): bit for bit the parameters of the two-launch form (tests/test_gpu_fused_update.py).
// The W -> half scale that k_scale_update would have computed between the two launches is derived by every parameter
// workgroup for itself from the previous update's per-block maxima (4-8 KB out of L2, while its slab loads fly).
// (Round 4, residency: 76 registers = six workgroups per CU resident, the eight of a CU in two uneven rounds.  A one-element form at 57
// registers with all eight resident is SLOWER (23.7 against 21.8 us), so is every cap below six (LDS-limited 5 / 4 / 3 per CU: 23.8-25 /
// 24.7 / 29.5 us); half or a quarter of the workgroups with two / four elements per thread: the same 21.3-22.6 us.)
template <typename T, bool S16 = false>
__global__ __launch_bounds__(256) void k_reduce_sgd(FusedUpdArgs fa) {
  const ReduceArgs& a = fa.r;
  const SgdArgs& g = fa.g;
  const int ndb = (a.D + 15) / 16;
  const int n_special = ndb + 1;
  const int bid = (int)blockIdx.x < n_special ? (int)gridDim.x - n_special + (int)blockIdx.x : (int)blockIdx.x - n_special;
  const int G = (int)gridDim.x;
  auto rule = [&](float w, float gr, float& h, float lr, float dc) {       // (k_sgd's)
    if (dc != 0.f) gr += dc * (g.reg == 2 ? w : (float)((w > 0.f) - (w < 0.f)));
    float u;
    if (g.solver_type == 1) { const float h0 = h; h = lr * gr + g.momentum * h0; u = (1.f + g.momentum) * h - g.momentum * h0; }
    else if (g.solver_type == 2) { h += gr * gr; u = lr * (gr / (sqrtf(h) + g.delta)); }
    else { h = lr * gr + g.momentum * h; u = h; }
    return w - u;
  };
  if (bid == G - 1) { reduce_loss(a); return; }
  if (bid >= G - 1 - ndb) {
    // bias columns: the partials' sum is the gradient (stored), and the bias is updated from it on the spot
    __shared__ float part[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int d = (bid - (G - 1 - ndb)) * 16 + tx;
    float s = 0.f;
    if (d < a.D) {
      const float* p = a.dbp + d;
      const int nb = a.db_rows > 0 ? a.db_rows : a.B;
      int b = ty;
      for (; b + 112 < nb; b += 128) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(b + 16 * u) * a.D];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; b < nb; b += 16) s += p[(int64_t)b * a.D];
    }
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && d < a.D) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) t += part[u][tx];
      a.grads[(int64_t)a.D * a.F + d] = t;
      float h = g.hb[d];
      g.b[d] = rule(g.b[d], t, h, g.rate * g.lr_mult_b, g.weight_decay * g.decay_mult_b);
      g.hb[d] = h;
    }
    return;
  }
  // ---- parameter workgroups: 16-byte elements, one per thread at the benchmark's size (2048 workgroups), more for larger matrices
  const int nblk = G - n_special;
  const int f4 = a.F / 4;
  const int64_t n4 = (int64_t)a.D * f4;
  const int64_t slab_sz = slab_pitch(a.Dp, a.Fp);
  int64_t i = (int64_t)bid * 256 + threadIdx.x;
  // the first element's loads fly while the scale is worked out
  float4 t[8], w = make_float4(0.f, 0.f, 0.f, 0.f), h = w;
  auto load = [&](int64_t ii) {
    const int d = (int)(ii / f4), f = (int)(ii % f4) * 4;
    const int64_t po = (int64_t)d * a.Fp + f;
    if constexpr (S16) {
      const int tiles = (a.Dp >> 8) * (a.Fp >> 8), tl = slab_tile(d, f, a.Fp);
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = u < a.S ? slab_ld4<true>(a.slabs, u * slab_sz + po, a.slab_sc[u * tiles + tl]) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = u < a.S ? nt_load4(a.slabs + po + u * slab_sz) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int64_t o = (int64_t)d * a.F + f;
    w = nt_load4(g.W + o); h = nt_load4(g.hW + o);
  };
  // S16 (f16 partial products): EIGHT elements per thread -- 16-byte loads of the eight slabs (8-byte accesses run at ~0.6 of the 16-byte rate
  // per byte), two 16-byte accesses each for W and the history, one 16-byte store of the half copy; the same per-element arithmetic in the same
  // order as the four-element form and as k_reduce + k_sgd (bit-identical parameters, tests/test_gpu_fused_update.py)
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
  const int f8 = a.F / 8;
  const int64_t n8 = (int64_t)a.D * f8;
  h8v t8[8]; float sc8[8]; float4 w1 = w, h1 = w;
  auto load8 = [&](int64_t ii) {
    const int d = (int)(ii / f8), f = (int)(ii % f8) * 8;
    const int64_t po = (int64_t)d * a.Fp + f;
    const int tiles = (a.Dp >> 8) * (a.Fp >> 8), tl = slab_tile(d, f, a.Fp);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u < a.S) {
        t8[u] = __builtin_bit_cast(h8v, __builtin_nontemporal_load((const u32x4v*)((const uint16_t*)a.slabs + u * slab_sz + po)));
        sc8[u] = a.slab_sc[u * tiles + tl];
      } else { t8[u] = h8v{0, 0, 0, 0, 0, 0, 0, 0}; sc8[u] = 0.f; }
    }
    const int64_t o = (int64_t)d * a.F + f;
    w = nt_load4(g.W + o); w1 = nt_load4(g.W + o + 4); h = nt_load4(g.hW + o); h1 = nt_load4(g.hW + o + 4);
  };
  if constexpr (S16) { if (i < n8) load8(i); }
  else { if (i < n4) load(i); }
  // the scale of the new half copy
  float sw = g.scales->sw_next;
  if (fa.recompute_scale) {
    __shared__ float red[4];
    float mm = 0.f;
    for (int k = threadIdx.x; k < fa.wmax_prev_n; k += 256) mm = fmaxf(mm, fa.wmax_prev[k]);
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) mm = fmaxf(mm, __shfl_xor(mm, o2, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mm;
    __syncthreads();
    mm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float m = fmaxf(mm, __uint_as_float(g.scales->wmax_bits));     // (vv_params_set's seed; the host clears it behind this launch)
    sw = 1.f;
    if (fa.prec == 0 && m > 0.f && isfinite(m)) {
      int e;
      frexpf(m, &e);
      sw = ldexpf(1.f, 12 - e);
    }
  }
  const float sgf = a.sg_dev ? *a.sg_dev : (a.gg ? a.sg * a.gg->mul : a.sg);
  const float inv = a.ip_scale / (sgf * a.scales->sx);
  const float lr_w = g.rate * g.lr_mult_w, dc_w = g.weight_decay * g.decay_mult_w;
  float wmax = 0.f;
  if constexpr (S16) {
    for (; i < n8; ) {
      const int d = (int)(i / f8), f = (int)(i % f8) * 8;
      const int64_t o = (int64_t)d * a.F + f;
      float gr[8], wn[8], hn[8];
      const float w8[8] = {w.x, w.y, w.z, w.w, w1.x, w1.y, w1.z, w1.w}, hh8[8] = {h.x, h.y, h.z, h.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float sj = (float)t8[0][j] * sc8[0];
#pragma unroll
        for (int u = 1; u < 8; ++u)
          if (u < a.S) sj += (float)t8[u][j] * sc8[u];
        gr[j] = __fmul_rn(sj, inv);
        hn[j] = hh8[j];
        wn[j] = rule(w8[j], gr[j], hn[j], lr_w, dc_w);
        wmax = fmaxf(wmax, fabsf(wn[j]));
      }
      if (fa.store_grads) { nt_store4(a.grads + o, make_float4(gr[0], gr[1], gr[2], gr[3])); nt_store4(a.grads + o + 4, make_float4(gr[4], gr[5], gr[6], gr[7])); }
      i += (int64_t)nblk * 256;
      if (i < n8) load8(i);                      // the next element's loads before this one's stores
      nt_store4(g.W + o, make_float4(wn[0], wn[1], wn[2], wn[3])); nt_store4(g.W + o + 4, make_float4(wn[4], wn[5], wn[6], wn[7]));
      nt_store4(g.hW + o, make_float4(hn[0], hn[1], hn[2], hn[3])); nt_store4(g.hW + o + 4, make_float4(hn[4], hn[5], hn[6], hn[7]));
      uint32_t q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) q[j] = T::from_float(wn[2 * j] * sw) | ((uint32_t)T::from_float(wn[2 * j + 1] * sw) << 16);
      *(uint4*)(g.Wh + (int64_t)d * g.Fp + f) = make_uint4(q[0], q[1], q[2], q[3]);
    }
  } else
  for (; i < n4; ) {
    const int d = (int)(i / f4), f = (int)(i % f4) * 4;
    const int64_t o = (int64_t)d * a.F + f;
    float4 s = t[0];
    for (int u = 1; u < 8; ++u)
      if (u < a.S) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
    // (rounded products, as the two-launch form stores them: no contraction into the rule's first multiply-add)
    const float4 gr = make_float4(__fmul_rn(s.x, inv), __fmul_rn(s.y, inv), __fmul_rn(s.z, inv), __fmul_rn(s.w, inv));
    if (fa.store_grads) nt_store4(a.grads + o, gr);
    float4 wn = w, hn = h;
    wn.x = rule(wn.x, gr.x, hn.x, lr_w, dc_w); wn.y = rule(wn.y, gr.y, hn.y, lr_w, dc_w);
    wn.z = rule(wn.z, gr.z, hn.z, lr_w, dc_w); wn.w = rule(wn.w, gr.w, hn.w, lr_w, dc_w);
    wmax = fmaxf(wmax, fmaxf(fmaxf(fabsf(wn.x), fabsf(wn.y)), fmaxf(fabsf(wn.z), fabsf(wn.w))));
    i += (int64_t)nblk * 256;
    if (i < n4) load(i);                       // the next element's loads before this one's stores
    nt_store4(g.W + o, wn);
    nt_store4(g.hW + o, hn);
    const uint32_t lo = T::from_float(wn.x * sw) | ((uint32_t)T::from_float(wn.y * sw) << 16);
    const uint32_t hi = T::from_float(wn.z * sw) | ((uint32_t)T::from_float(wn.w * sw) << 16);
    *(uint2*)(g.Wh + (int64_t)d * g.Fp + f) = make_uint2(lo, hi);
  }
  __shared__ float wm[4];
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o2, 64));
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = wmax;
  __syncthreads();
  if (threadIdx.x == 0) {
    g.wmax_blocks[bid] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (bid == 0) { g.scales->sw_cur = sw; if (fa.recompute_scale) g.scales->sw_next = sw; }
  }
}
int launch_reduce_sgd(const FusedUpdArgs& a, hipStream_t s) {
  const int ndb = (a.r.D + 15) / 16;
  const int ept = a.r.slab16 ? 8 : 4;                                       // elements per thread (k_reduce_sgd's S16 form: eight)
  const int nblk = a.no_params ? 0 : (int)std::min<int64_t>(((int64_t)a.r.D * (a.r.F / ept) + 255) / 256, WMAX_SLOTS);
  const dim3 grid(nblk + ndb + 1);
  if (a.r.slab16) {
    if (a.prec == 0) VV_LAUNCH((k_reduce_sgd<F16, true>), grid, dim3(256), 0, s, a);
    else VV_LAUNCH((k_reduce_sgd<BF16, true>), grid, dim3(256), 0, s, a);
    return nblk;
  }
  if (a.prec == 0) VV_LAUNCH((k_reduce_sgd<F16>), grid, dim3(256), 0, s, a);
  else VV_LAUNCH((k_reduce_sgd<BF16>), grid, dim3(256), 0, s, a);
  return nblk;
}

__global__ __launch_bounds__(256) void k_absmax(const float* x, int64_t n, unsigned* out_bits) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));
}
void launch_absmax(const float* x, int64_t n, unsigned* out_bits, hipStream_t s) {
  hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, s, x, n, out_bits);
}

template <typename T>
__global__ __launch_bounds__(256) void k_w_convert(const float* W, uint16_t* Wh, int D, int F,
                                                   int Fp, Scales* sc) {
  const float sw = sc->sw_next;
  const int64_t nW = (int64_t)D * F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nW; i += (int64_t)gridDim.x * 256) {
    const int d = (int)(i / F), f = (int)(i % F);
    Wh[(int64_t)d * Fp + f] = T::from_float(W[i] * sw);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) sc->sw_cur = sw;
}
void launch_w_convert(int prec, const float* W, uint16_t* Wh, int D, int F, int Dp, int Fp,
                      Scales* sc, hipStream_t s) {
  (void)Dp;
  if (prec == 0) hipLaunchKernelGGL(k_w_convert<F16>, dim3(1024), dim3(256), 0, s, W, Wh, D, F, Fp, sc);
  else hipLaunchKernelGGL(k_w_convert<BF16>, dim3(1024), dim3(256), 0, s, W, Wh, D, F, Fp, sc);
}

// ------------------------------------------------------------------------------- table --------
template <typename T>
__global__ __launch_bounds__(256) void k_table_convert(const float* src, uint16_t* dst,
                                                       int64_t n_rows, int F, int Fp, float sx) {
  const int64_t n = n_rows * F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / F; const int f = (int)(i % F);
    dst[r * Fp + f] = T::from_float(src[i] * sx);
  }
}
void launch_table_convert(int prec, const float* src, uint16_t* dst, int64_t n_rows, int F, int Fp,
                          float sx, hipStream_t s) {
  if (prec == 0) hipLaunchKernelGGL(k_table_convert<F16>, dim3(2048), dim3(256), 0, s, src, dst, n_rows, F, Fp, sx);
  else hipLaunchKernelGGL(k_table_convert<BF16>, dim3(2048), dim3(256), 0, s, src, dst, n_rows, F, Fp, sx);
}

// synthetic features, bit-identical to videovector_amd/synth.py:feature_rows
template <typename T>
__global__ __launch_bounds__(256) void k_table_synth(uint16_t* dst, uint64_t seed, int64_t n_rows,
                                                     int F, int Fp, float sx) {
  const int64_t n = n_rows * F;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / F; const int f = (int)(i % F);
    const uint64_t h = mix64(seed, (1ull << 40) + (uint64_t)i);
    const int s = (int)(h & 15) + (int)((h >> 4) & 15) + (int)((h >> 8) & 15) + (int)((h >> 12) & 15);
    const float x = (float)(s > 30 ? s - 30 : 0) * 0.125f;
    dst[r * Fp + f] = T::from_float(x * sx);
  }
}
void launch_table_synth(int prec, uint16_t* dst, uint64_t seed, int64_t n_rows, int F, int Fp,
                        float sx, hipStream_t s) {
  if (prec == 0) hipLaunchKernelGGL(k_table_synth<F16>, dim3(4096), dim3(256), 0, s, dst, seed, n_rows, F, Fp, sx);
  else hipLaunchKernelGGL(k_table_synth<BF16>, dim3(4096), dim3(256), 0, s, dst, seed, n_rows, F, Fp, sx);
}

template <typename T>
__global__ __launch_bounds__(256) void k_table_read(const uint16_t* table, const int32_t* rows,
                                                    int64_t n, int F, int Fp, float inv_sx,
                                                    float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n * F; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / F; const int f = (int)(i % F);
    const int64_t src = rows ? rows[r] : r;
    out[i] = T::to_float(table[src * Fp + f]) * inv_sx;
  }
}
void launch_table_read(int prec, const uint16_t* table, const int32_t* rows, int64_t n, int F,
                       int Fp, float inv_sx, float* out, hipStream_t s) {
  if (prec == 0) hipLaunchKernelGGL(k_table_read<F16>, dim3(1024), dim3(256), 0, s, table, rows, n, F, Fp, inv_sx, out);
  else hipLaunchKernelGGL(k_table_read<BF16>, dim3(1024), dim3(256), 0, s, table, rows, n, F, Fp, inv_sx, out);
}

// composite rows for quirk Q1: row first_row+p = features 0..F-2 of desc[2p], feature F-1 of desc[2p+1]
__global__ __launch_bounds__(256) void k_patch_rows(uint16_t* table, const int32_t* desc, int64_t first_row,
                                                    int F, int Fp) {
  const int64_t p = blockIdx.x;
  const uint16_t* src = table + (int64_t)desc[2 * p] * Fp;
  const int32_t ls = desc[2 * p + 1];
  uint16_t* dst = table + (first_row + p) * Fp;
  for (int f = threadIdx.x; f < Fp; f += 256) {
    uint16_t v = src[f];
    if (f == F - 1) v = ls >= 0 ? table[(int64_t)ls * Fp + f] : (uint16_t)0;
    dst[f] = v;
  }
}
void launch_patch_rows(uint16_t* table, const int32_t* desc, int64_t n_patch, int64_t first_row, int F,
                       int Fp, hipStream_t s) {
  hipLaunchKernelGGL(k_patch_rows, dim3((unsigned)n_patch), dim3(256), 0, s, table, desc, first_row, F, Fp);
}

// scratch row first_row+i = sum_j coeff[j] * table[rows[i*k+j]]   (TEST branch: average_for_test)
template <typename T>
__global__ __launch_bounds__(256) void k_mean_rows(uint16_t* table, const int32_t* rows, int k, const float* coeff,
                                                   int64_t first_row, int Fp) {
  const int64_t i = blockIdx.x;
  uint16_t* dst = table + (first_row + i) * Fp;
  for (int f = threadIdx.x; f < Fp; f += 256) {
    float s = 0.f;
    for (int j = 0; j < k; ++j) s += coeff[j] * T::to_float(table[(int64_t)rows[i * k + j] * Fp + f]);
    dst[f] = T::from_float(s);
  }
}
void launch_mean_rows(int prec, uint16_t* table, const int32_t* rows, int64_t n, int k, const float* coeff,
                      int64_t first_row, int Fp, hipStream_t s) {
  if (prec == 0) hipLaunchKernelGGL(k_mean_rows<F16>, dim3((unsigned)n), dim3(256), 0, s, table, rows, k, coeff, first_row, Fp);
  else hipLaunchKernelGGL(k_mean_rows<BF16>, dim3((unsigned)n), dim3(256), 0, s, table, rows, k, coeff, first_row, Fp);
}

// out[i][j] = alpha * sum_d x[i][d] x[j][d]  (fp32; retrieval_stats_layer.cpp:208-209 uses alpha = -2).
// 64x64 tile per 256-thread block, 4x4 outputs per thread; n is a few hundred to a few thousand.
__global__ __launch_bounds__(256) void k_gram(const float* x, int n, int dim, float alpha, float* out) {
  __shared__ float As[16][64 + 1], Bs[16][64 + 1];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < dim; k0 += 16) {
    for (int t = threadIdx.x; t < 64 * 16; t += 256) {
      const int r = t >> 4, c = t & 15;
      As[c][r] = (i0 + r < n && k0 + c < dim) ? x[(int64_t)(i0 + r) * dim + k0 + c] : 0.f;
      Bs[c][r] = (j0 + r < n && k0 + c < dim) ? x[(int64_t)(j0 + r) * dim + k0 + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { a[u] = As[c][ty * 4 + u]; b[u] = Bs[c][tx * 4 + u]; }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] += a[u] * b[v];
    }
    __syncthreads();
  }
  for (int u = 0; u < 4; ++u)
    for (int v = 0; v < 4; ++v) {
      const int i = i0 + ty * 4 + u, j = j0 + tx * 4 + v;
      if (i < n && j < n) out[(int64_t)i * n + j] = alpha * acc[u][v];
    }
}
void launch_gram(const float* x, int n, int dim, float alpha, float* out, hipStream_t s) {
  const dim3 grid((n + 63) / 64, (n + 63) / 64);
  hipLaunchKernelGGL(k_gram, grid, dim3(256), 0, s, x, n, dim, alpha, out);
}

// idx (data-layer layout, -1 = empty slot) -> table rows, padded to Rp with the all-zero row
// An index outside [0, row_limit) -- -1 is the documented "empty slot", anything else can only come from a device
// index array the host never saw -- reads the all-zero row instead of memory outside the table.
__global__ void k_map_rows(const int32_t* idx, int32_t* rows, int R, int Rp, int32_t zero_row, int32_t row_limit) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < Rp) rows[i] = (i < R && idx[i] >= 0 && idx[i] < row_limit) ? idx[i] : zero_row;
}
void launch_map_rows(const int32_t* idx, int32_t* rows, int R, int Rp, int32_t zero_row, int32_t row_limit,
                     hipStream_t s) {
  hipLaunchKernelGGL(k_map_rows, dim3((Rp + 255) / 256), dim3(256), 0, s, idx, rows, R, Rp, zero_row, row_limit);
}

template <typename T>
__global__ void k_dyh_to_float(const uint16_t* dYh, int R, int D, int Dp, float inv_sg, float* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)R * D; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / D; const int d = (int)(i % D);
    out[i] = T::to_float(dYh[r * Dp + d]) * inv_sg;
  }
}
void launch_dyh_to_float(int prec, const uint16_t* dYh, int R, int D, int Dp, float inv_sg,
                         float* out, hipStream_t s) {
  if (prec == 0) hipLaunchKernelGGL(k_dyh_to_float<F16>, dim3(1024), dim3(256), 0, s, dYh, R, D, Dp, inv_sg, out);
  else hipLaunchKernelGGL(k_dyh_to_float<BF16>, dim3(1024), dim3(256), 0, s, dYh, R, D, Dp, inv_sg, out);
}

// NORMALIZATION forward on rows of x in place (normalization_layer.cu:10-45): y = x/(|x|+1e-10)
__global__ __launch_bounds__(256) void k_row_normalize(float* x, int n, int D) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= n) return;
  float* p = x + (int64_t)r * D;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s += p[d] * p[d];
  s = wave_sum(s);
  const float inv = 1.f / (sqrtf(s) + 1e-10f);
  for (int d = lane; d < D; d += 64) p[d] *= inv;
}
void launch_row_normalize(float* x, int n, int D, hipStream_t s) {
  hipLaunchKernelGGL(k_row_normalize, dim3((n + 3) / 4), dim3(256), 0, s, x, n, D);
}

}  // namespace vv
