// fwd_stream_lab: the LDS-DMA stream of k_fwd_gemm_ph alone (no fragment reads, no MFMA), at the de-duplicated size of the
// benchmark (20 650 distinct rows, 192-row tiles x 2 column halves = 216 workgroups), to find what sets its rate:
//   mode 0  as the kernel issues it today (the two column halves of a row tile read the same rows at the same time)
//   mode 1  every gathered row is the L2-hot zero row
//   mode 2  the two column halves read DIFFERENT rows (no line is asked for twice)
//   LEAD n  column half 0 asks for its A_lo half-tile n K-tiles early and column half 1 for its A_hi half-tile: each line's
//           first (missing) request comes from one workgroup and the sibling's request, n K-tiles later, finds it in L2
//   both    control: both siblings lead with the SAME half (a deeper prefetch of A only, still simultaneous)
// Build: hipcc --offload-arch=gfx950 -O2 fwd_stream_lab.hip -o fwd_stream_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_wave_base));
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(gsrc) : "m0", "memory");
}
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int x = bid & 7, q = nblk >> 3, rem = nblk & 7;
  return x * q + (x < rem ? x : rem) + (bid >> 3);
}
constexpr int SLOT = 16384;

struct Args {
  const unsigned short* table; const unsigned short* Wh; const int* rows; int zero_row; int Fp; int R; int mode; int lead; int both; int wait8; int dist;
  unsigned* sink;
};

__global__ __launch_bounds__(512) void k_stream(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HROWS = 96, BMT = 192, BK = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;
  const int nact = ((a.R + BMT - 1) / BMT) * 2;
  if ((int)blockIdx.x >= nact) return;
  const int L = xcd_remap(blockIdx.x, nact);
  const int m0 = (L / 2) * BMT, nh = L & 1, n0 = nh * 256;
  const int Fp = a.Fp;
  const unsigned short* srcA[2][2];
  const unsigned short* srcB[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 8 + (lane >> 3), lc = (lane & 7) ^ (row & 7);
      const int grow = m0 + hf * HROWS + row;
      int trow = a.zero_row;
      if (row < HROWS && grow < a.R && a.mode != 1) trow = a.rows[grow + (a.mode == 2 ? nh * a.R : 0)];
      srcA[hf][i] = a.table + (long)trow * Fp + lc * 8;
      srcB[hf][i] = a.Wh + (long)(n0 + hf * 128 + row) * Fp + lc * 8;
    }
  const int nk = 4096 / BK;
  const int lead_q = a.both ? 0 : (nh == 0 ? 0 : 3);     // which A half this workgroup asks for early
  auto issue = [&](int kt, int q) {
    const unsigned short* const* src = q == 0 ? srcA[0] : q == 1 ? srcB[0] : q == 2 ? srcB[1] : srcA[1];
    int k = kt, slot = (4 * kt + q) & 7;
    if (a.lead && q == lead_q) { k = kt + a.lead; slot = 8 + (k & 1); if (k >= nk) return false; }
    unsigned char* dst = smem + slot * SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(src[i] + k * BK, dst + (i * 8 + wave) * 1024);
    return true;
  };
  // the leading half's first `lead` K-tiles
  if (a.lead)
    for (int k = 0; k < a.lead; ++k) {
      const unsigned short* const* src = lead_q == 0 ? srcA[0] : srcA[1];
#pragma unroll
      for (int i = 0; i < 2; ++i) glds16(src[i] + k * BK, smem + (8 + (k & 1)) * SLOT + (i * 8 + wave) * 1024);
    }
  // prologue: half-tiles 0..5
  issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3); issue(1, 0); issue(1, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  const int H = 4 * nk;
  unsigned acc = 0;
  for (int t = 0; t < nk; ++t) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int h = 4 * t + p + 6;
      if (h < H) {
        issue(h >> 2, h & 3);
        if (p != 2) { if (a.wait8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); }
      } else if (p != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      acc += *(volatile unsigned*)(smem + ((4 * t + p) & 7) * SLOT + tid * 4);     // one token read per phase
      __builtin_amdgcn_s_barrier();
    }
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  if (acc == 0x12345678u) a.sink[0] = acc;
}

// the same stream issued DIST half-tiles ahead of the phase that needs them; the counted wait leaves DIST - 2 in flight
template <int DIST>
__global__ __launch_bounds__(512) void k_stream_d(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int HROWS = 96, BMT = 192, BK = 64;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2;
  const int nact = ((a.R + BMT - 1) / BMT) * 2;
  if ((int)blockIdx.x >= nact) return;
  const int L = xcd_remap(blockIdx.x, nact);
  const int m0 = (L / 2) * BMT, nh = L & 1, n0 = nh * 256;
  const int Fp = a.Fp;
  const unsigned short* srcA[2][2];
  const unsigned short* srcB[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = (i * 8 + wave) * 8 + (lane >> 3), lc = (lane & 7) ^ (row & 7);
      const int grow = m0 + hf * HROWS + row;
      int trow = a.zero_row;
      if (row < HROWS && grow < a.R && a.mode != 1) trow = a.rows[grow];
      srcA[hf][i] = a.table + (long)trow * Fp + lc * 8;
      srcB[hf][i] = a.Wh + (long)(n0 + hf * 128 + row) * Fp + lc * 8;
    }
  const int nk = 4096 / BK, H = 4 * nk;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem)) + wave * 1024;
  auto issue = [&](int h) {
    const int kt = h >> 2, q = h & 3, slot = h % 10;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned short* src = q == 0 ? srcA[0][i] : q == 1 ? srcB[0][i] : q == 2 ? srcB[1][i] : srcA[1][i];
      const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + slot * SLOT + i * 8192);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(src + kt * BK) : "m0", "memory");
    }
  };
#pragma unroll
  for (int h = 0; h < DIST; ++h) issue(h);
  constexpr int W = 2 * (DIST - 2);
  if constexpr (W == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (W == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (W == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();
  unsigned acc = 0;
  for (int P = 0; P < H; ++P) {
    const int h = P + DIST;
    if (h < H) {
      issue(h);
      if ((P & 3) != 2) {
        if constexpr (W == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (W == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (W == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
      }
    } else if ((P & 3) != 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += *(volatile unsigned*)(smem + (P % 10) * SLOT + tid * 4);
    __builtin_amdgcn_s_barrier();
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();
  if (acc == 0x12345678u) a.sink[0] = acc;
}

int main() {
  const int F = 4096, n_rows = 60000, R = 20650, D = 512;
  unsigned short *table, *Wh; int* rows; unsigned* sink;
  CHK(hipMalloc(&table, (size_t)(n_rows + 1) * F * 2)); CHK(hipMemset(table, 1, (size_t)(n_rows + 1) * F * 2));
  CHK(hipMalloc(&Wh, (size_t)D * F * 2)); CHK(hipMemset(Wh, 1, (size_t)D * F * 2));
  CHK(hipMalloc(&sink, 64));
  std::vector<int> perm(n_rows);
  for (int i = 0; i < n_rows; ++i) perm[i] = i;
  unsigned long long s = 88172645463325252ull;
  for (int i = n_rows - 1; i > 0; --i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(perm[i], perm[(int)(s % (unsigned)(i + 1))]); }
  CHK(hipMalloc(&rows, (size_t)2 * R * 4)); CHK(hipMemcpy(rows, perm.data(), (size_t)2 * R * 4, hipMemcpyHostToDevice));   // 2R distinct rows
  hipStream_t st; CHK(hipStreamCreate(&st));
  const int lds = 10 * SLOT;
  CHK(hipFuncSetAttribute((const void*)k_stream, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = ((R + 191) / 192) * 2;
  auto run = [&](const char* name, int mode, int lead, int both, int wait8) {
    Args a{table, Wh, rows, n_rows, F, R, mode, lead, both, wait8, 6, sink};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(512), lds, st, a);
    double best = 1e9, sum = 0;
    const int reps = 10;
    for (int w = 0; w < reps; ++w) {
      CHK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(k_stream, dim3(grid), dim3(512), lds, st, a);
      CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, (double)ms); sum += ms;
    }
    printf("%-58s min %.1f us  mean %.1f us  (%.1f GB/s per workgroup)\n", name, best * 1e3, sum / reps * 1e3, 4.0 * 1048576 / (best * 1e-3) / 1e9);
  };
  auto run_d = [&](const char* name, int dist, int mode) {
    Args a{table, Wh, rows, n_rows, F, R, mode, 0, 0, 1, dist, sink};
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto launch = [&]() {
      if (dist == 6) hipLaunchKernelGGL(k_stream_d<6>, dim3(grid), dim3(512), lds, st, a);
      else if (dist == 8) hipLaunchKernelGGL(k_stream_d<8>, dim3(grid), dim3(512), lds, st, a);
      else if (dist == 10) hipLaunchKernelGGL(k_stream_d<10>, dim3(grid), dim3(512), lds, st, a);
      else hipLaunchKernelGGL(k_stream_d<12>, dim3(grid), dim3(512), lds, st, a);
    };
    for (int w = 0; w < 3; ++w) launch();
    double best = 1e9, sum = 0;
    const int reps = 10;
    for (int w = 0; w < reps; ++w) {
      CHK(hipEventRecord(e0, st)); launch(); CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, (double)ms); sum += ms;
    }
    printf("%-58s min %.1f us  mean %.1f us  (%.1f GB/s per workgroup)\n", name, best * 1e3, sum / reps * 1e3, 4.0 * 1048576 / (best * 1e-3) / 1e9);
  };
  CHK(hipFuncSetAttribute((const void*)k_stream_d<6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHK(hipFuncSetAttribute((const void*)k_stream_d<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHK(hipFuncSetAttribute((const void*)k_stream_d<10>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CHK(hipFuncSetAttribute((const void*)k_stream_d<12>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  printf("%d workgroups, 4 MiB of LDS-DMA each\n", grid);
  for (int rep = 0; rep < 2; ++rep) {
    run_d("issued 6 ahead, 4 in flight past the wait, real rows", 6, 0);
    run_d("issued 8 ahead, 6 in flight, real rows", 8, 0);
    run_d("issued 10 ahead, 8 in flight, real rows", 10, 0);
    run_d("issued 12 ahead, 10 in flight, real rows", 12, 0);
    run_d("issued 6 ahead, L2-hot rows", 6, 1);
    run_d("issued 8 ahead, L2-hot rows", 8, 1);
    run_d("issued 10 ahead, L2-hot rows", 10, 1);
    run_d("issued 12 ahead, L2-hot rows", 12, 1);
  }
  for (int rep = 0; rep < 1; ++rep) {
    run("mode 0 (today)", 0, 0, 0, 1);
    run("mode 1 (every row L2-hot)", 1, 0, 0, 1);
    run("mode 2 (siblings read different rows)", 2, 0, 0, 1);
    run("lead 1, siblings lead different halves", 0, 1, 0, 1);
    run("lead 2, siblings lead different halves", 0, 2, 0, 1);
    run("lead 3, siblings lead different halves", 0, 3, 0, 1);
    run("lead 4, siblings lead different halves", 0, 4, 0, 1);
    run("lead 2, both siblings lead A_lo (control)", 0, 2, 1, 1);
    run("lead 4, both siblings lead A_lo (control)", 0, 4, 1, 1);
    run("mode 0, vmcnt(12) (six half-tiles in flight)", 0, 0, 0, 0);
    run("lead 2, different halves, vmcnt(12)", 0, 2, 0, 0);
  }
  return 0;
}
