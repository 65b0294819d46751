// gather_burst_lab: does the gathered stream of k_fwd_gemm_ph (20 650 distinct rows of 8 KiB from a 671 MB table, 216 pullers) run at
// ~3 TB/s BECAUSE every row is visited 64 times, 128 bytes a visit (one K-tile), about a microsecond apart?  The same bytes, the same
// pullers, the same number of requests in flight -- but each visit takes BURST contiguous bytes of the row (128 = the kernel, 256, 512,
// 1024, 2048): fewer, longer visits per row.  LDS-DMA into a scratch image, as the kernel stages.  Row sets rotate: every launch is cold.
// Build: hipcc --offload-arch=gfx950 -O2 gather_burst_lab.hip -o gather_burst_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(lds_off), "v"(gsrc) : "m0", "memory");
}
struct Args { const unsigned short* table; const int* rows; int R; int Fp; int burst; int window; int dup; };

template <int WINDOW>
__global__ __launch_bounds__(512) void k_gather(Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem)) + wave * 16384;
  const int nact = ((a.R + 191) / 192) * 2;
  if ((int)blockIdx.x >= nact) return;
  const int x = blockIdx.x & 7, q = nact >> 3, rem = nact & 7;
  const int L = x * q + (x < rem ? x : rem) + (blockIdx.x >> 3);
  const int m0 = (L / 2) * 192, sib = L & 1;
  // a wave instruction = 1 KiB = rpi rows x burst bytes; a step = every row of the tile advanced by `burst` bytes: 192 / rpi instructions,
  // spread over the 8 waves (wave w takes instructions w, w + 8, ...)
  const int burst = a.burst, lpr = burst / 16, rpi = 64 / lpr;        // lanes per row, rows per instruction
  const int ipstep = 192 / rpi, steps = a.Fp * 2 / burst;
  // the rows' table offsets first (in the kernel they sit in registers before the loop starts): up to 24 instructions per wave and step
  long off[24];
#pragma unroll
  for (int j = 0; j < 24; ++j) {
    const int i = wave + 8 * j;
    int r = i * rpi + lane / lpr;
    if (!a.dup) r = (r + sib * 96) % 192;                    // (dup 0: siblings start half a tile apart -- same lines, different moments)
    const int grow = m0 + r;
    off[j] = i < ipstep ? (long)(grow < a.R ? a.rows[grow] : a.rows[0]) * a.Fp + (lane % lpr) * 8 : 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int issued = 0;
  for (int s = 0; s < steps; ++s) {
#pragma unroll
    for (int j = 0; j < 24; ++j) {
      if (wave + 8 * j >= ipstep) break;
      glds16(a.table + off[j] + (long)s * (burst / 2), lds0 + (issued % 16) * 1024);
      ++issued;
      if (issued >= WINDOW) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WINDOW - 1) : "memory");
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

int main(int argc, char** argv) {
  const int n_rows = 81920, Fp = 4096, R = 20650, launches = 24;
  unsigned short* table; CHK(hipMalloc(&table, (size_t)n_rows * Fp * 2)); CHK(hipMemset(table, 1, (size_t)n_rows * Fp * 2));
  std::mt19937 rng(7);
  std::vector<int> perm(n_rows); for (int i = 0; i < n_rows; ++i) perm[i] = i;
  int* rows; CHK(hipMalloc(&rows, (size_t)launches * R * 4));
  std::vector<int> h((size_t)launches * R);
  for (int l = 0; l < launches; ++l) {            // a fresh random row set per launch (3 of them cover most of the table: nothing stays cached)
    std::shuffle(perm.begin(), perm.end(), rng);
    std::copy(perm.begin(), perm.begin() + R, h.begin() + (size_t)l * R);
    std::sort(h.begin() + (size_t)l * R, h.begin() + (size_t)(l + 1) * R);       // (the de-duplicated rows arrive in table order)
  }
  if (argc > 1 && atoi(argv[1]) == 0) for (int l = 0; l < launches; ++l) std::shuffle(h.begin() + (size_t)l * R, h.begin() + (size_t)(l + 1) * R, rng);
  CHK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CHK(hipFuncSetAttribute((const void*)k_gather<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CHK(hipFuncSetAttribute((const void*)k_gather<12>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  CHK(hipFuncSetAttribute((const void*)k_gather<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int grid = ((R + 191) / 192) * 2;
  printf("rows %s; %d pullers x 192 rows x 8 KiB (siblings ask for the same rows); us per launch, GB/s of DISTINCT bytes (169 MB)\n", (argc > 1 && atoi(argv[1]) == 0) ? "in random order" : "in table order", grid);
  for (int dup = 1; dup >= 0; --dup)
  for (int window : {8, 12, 16})
    for (int burst : {128, 256, 512, 1024}) {
      float best = 1e9f, sum = 0;
      for (int l = 0; l < launches; ++l) {
        Args a{table, rows + (size_t)l * R, R, Fp, burst, window, dup};
        CHK(hipEventRecord(e0));
        if (window == 8) hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(512), 131072, 0, a);
        else if (window == 12) hipLaunchKernelGGL(k_gather<12>, dim3(grid), dim3(512), 131072, 0, a);
        else hipLaunchKernelGGL(k_gather<16>, dim3(grid), dim3(512), 131072, 0, a);
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (l >= 4) { best = std::min(best, ms); sum += ms; }
      }
      const float avg = sum / (launches - 4);
      printf("siblings %s  window %2d instr/wave  burst %4d B: avg %6.1f us  min %6.1f us  -> %5.0f GB/s\n", dup ? "together " : "staggered", window, burst, avg * 1e3, best * 1e3, 169.2e6 / (avg * 1e-3) / 1e9);
    }
  return 0;
}
