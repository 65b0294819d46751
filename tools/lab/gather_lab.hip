// gather_lab: how fast can a CU fill LDS with 128-B row pieces gathered from a large table?
// Sweeps ring depth (bytes in flight), column rotation (HBM channel spread) and piece width.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// each step: ROWS rows x PIECE bytes per WG.  512 threads; NL = ROWS*PIECE/16/512 loads per thread
template <int DEPTH, int PIECE, int ROWS, bool ROT>
__global__ __launch_bounds__(512) void k_gather(const unsigned short* table, const int* rows, int Fp,
                                                int nsteps, int tiles_per_wg, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int STEP_BYTES = ROWS * PIECE;
  constexpr int NL = STEP_BYTES / 16 / 512;
  constexpr int CPR = PIECE / 16;   // chunks per row
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned acc = 0;
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x * tiles_per_wg + t;
    const unsigned short* src[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int c = (i * 8 + wave) * 64 + lane;
      const int row = c / CPR, ch = c % CPR;
      src[i] = table + (long)rows[tile * ROWS + row] * Fp + ch * 8;
    }
    const int rot = ROT ? (tile * 37) % nsteps : 0;
    auto stage = [&](int s) {
      int ks = s + rot; if (ks >= nsteps) ks -= nsteps;
      unsigned char* dst = smem + (s % DEPTH) * STEP_BYTES;
#pragma unroll
      for (int i = 0; i < NL; ++i)
        __builtin_amdgcn_global_load_lds(GLB_PTR(src[i] + ks * (PIECE / 2)), LDS_PTR(dst + (i * 8 + wave) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) stage(s);
    for (int s = 0; s < nsteps; ++s) {
      if (s + DEPTH - 1 < nsteps) stage(s + DEPTH - 1);
      // wait until step s has landed: allow NL*(DEPTH-1) newer loads in flight (fewer near the end)
      if (s + DEPTH - 1 < nsteps) {
        if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if constexpr (NL * (DEPTH - 1) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      acc += *(unsigned*)(smem + (s % DEPTH) * STEP_BYTES + tid * 4);
      __builtin_amdgcn_s_barrier();
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int DEPTH, int PIECE, int ROWS, bool ROT>
double run(const unsigned short* table, const int* rows, int Fp, int ntiles, int grid, hipStream_t st, unsigned* sink, int kext = 0) {
  const int lds = DEPTH * ROWS * PIECE;
  CHK(hipFuncSetAttribute((const void*)k_gather<DEPTH, PIECE, ROWS, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int nsteps = (kext ? kext : Fp) * 2 / PIECE;
  const int tpw = ntiles / grid;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_gather<DEPTH, PIECE, ROWS, ROT>), dim3(grid), dim3(512), lds, st, table, rows, Fp, nsteps, tpw, sink);
  CHK(hipEventRecord(e0, st));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_gather<DEPTH, PIECE, ROWS, ROT>), dim3(grid), dim3(512), lds, st, table, rows, Fp, nsteps, tpw, sink);
  CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int Fp = 4096, n_rows = 81914, ROWS = 256;
  const int ntiles = 512;    // 512 tiles x 256 rows = 131072 gathered rows (1.07 GB of row bytes)
  unsigned short* table; int* rows; unsigned* sink;
  CHK(hipMalloc(&table, (size_t)n_rows * (Fp + 256) * 2)); CHK(hipMemset(table, 1, (size_t)n_rows * (Fp + 256) * 2));
  CHK(hipMalloc(&sink, 64));
  std::vector<int> h(ntiles * ROWS);
  unsigned long long s = 88172645463325252ull;
  for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (int)(s % n_rows); }
  CHK(hipMalloc(&rows, h.size() * 4)); CHK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  hipStream_t st; CHK(hipStreamCreate(&st));
  const double bytes = (double)ntiles * ROWS * Fp * 2;
#define RUN(D, P, R, ROTB, G) { double ms = run<D, P, R, ROTB>(table, rows, Fp, ntiles, G, st, sink); \
    printf("depth %d piece %4d rot %d grid %3d inflight %3d KB : %.3f ms  %.2f TB/s  %.1f GB/s/WG\n", D, P, (int)ROTB, G, (D - 1 > 0 ? D - 1 : 1) * R * P / 1024, ms, bytes / ms / 1e9, bytes / ms / 1e6 / G); }
  printf("random rows, 8 KiB rows (1.07 GB gathered per launch)\n");
  RUN(2, 128, 256, false, 256) RUN(2, 128, 256, true, 256)
  RUN(3, 128, 256, false, 256) RUN(3, 128, 256, true, 256)
  RUN(4, 128, 256, false, 256) RUN(4, 128, 256, true, 256)
  RUN(5, 128, 256, true, 256)
  RUN(2, 256, 256, false, 256) RUN(2, 256, 256, true, 256)
  RUN(2, 512, 64, false, 256) RUN(2, 512, 64, true, 256) RUN(4, 512, 64, true, 256)
  RUN(2, 128, 256, true, 128) RUN(4, 128, 256, true, 128)
  RUN(2, 128, 256, true, 512) RUN(4, 128, 256, false, 512)
  // L2-hot operand (the W side of the forward GEMM): 512 rows shared by every workgroup, K extent 4096, row stride ld halves
  {
    const int nW = 512;
    std::vector<int> hw(ntiles * ROWS);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (int)(i % nW);
    CHK(hipMemcpy(rows, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    printf("512 shared rows (L2), K extent 4096, by row stride\n");
    for (int ld : {4096, 4096 + 32, 4096 + 64, 4096 + 128, 4096 + 192}) {
#define RUNW(D, P, R, ROTB, G) { double ms = run<D, P, R, ROTB>(table, rows, ld, ntiles, G, st, sink, 4096); \
    printf("stride %d B depth %d piece %4d rot %d grid %3d : %.3f ms  %.2f TB/s  %.1f GB/s/WG\n", ld * 2, D, P, (int)ROTB, G, ms, bytes / ms / 1e9, bytes / ms / 1e6 / G); }
      RUNW(4, 128, 256, false, 256) RUNW(4, 128, 256, true, 256) RUNW(4, 128, 256, false, 216)
    }
    // random rows of the big table again, by stride
    CHK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    printf("random rows of the 670 MB table, K extent 4096, by row stride\n");
    for (int ld : {4096, 4096 + 64, 4096 + 128}) { RUNW(4, 128, 256, false, 256) RUNW(4, 128, 256, true, 256) }
  }
  return 0;
}
