// wgrad_ts_lab: k_wgrad_gemm_ph at the benchmark's de-duplicated size (dW = dYu^T X over U = 20 650 distinct rows, 512 x 4096, split-K 8)
// with per-wave time stamps (ABL bit 9, kernels_gemm_ph.hip): where a phase's clocks go, prologue (row ids -> LDS) and epilogue (slab stores).
// Row sets rotate between launches (4 x 169 MB out of a 671 MB table).  Build: tools/lab/wgrad_ts_lab.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <random>
#include "../../videovector_amd/csrc/vv_internal.h"
namespace vv { thread_local ProfPair g_prof; thread_local const KernelOpts* g_ko = nullptr; }
#include "../../videovector_amd/csrc/kernels_gemm_ph.hip"
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
using namespace vv;

__global__ void k_fill16(uint16_t* t, int64_t n, uint64_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix64(seed, (uint64_t)i);
    float v = ((float)((h >> 8) & 0xffff) * (1.0f / 32768.f) - 1.0f) * scale;
    if (scale < 0.f) v = (h & 3) ? (float)((h >> 8) & 0xffff) * (1.0f / 65536.f) : 0.f;
    t[i] = F16::from_float(v);
  }
}
template <int ABL>
static void launch(const WgradArgs& a, hipStream_t s) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_wgrad_gemm_ph<F16, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, PH_WG_LDS_BYTES), true);
  (void)once;
  const dim3 grid((a.Dp / BM) * (a.Fp / BN) * a.S), block(GEMM_THREADS);
  hipLaunchKernelGGL((k_wgrad_gemm_ph<F16, ABL>), grid, block, PH_WG_LDS_BYTES, s, a);
}
// (round 6) the LEAN instantiations (fp32 slabs): k_wgrad_gemm_ph<F16, ABL, false, false, true>
template <int ABL>
static void launch_lean(const WgradArgs& a, hipStream_t s) {
  static bool once = ((void)hipFuncSetAttribute((const void*)k_wgrad_gemm_ph<F16, ABL, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PH_WG_LDS_BYTES), true);
  (void)once;
  const dim3 grid((a.Dp / BM) * (a.Fp / BN) * a.S), block(GEMM_THREADS);
  hipLaunchKernelGGL((k_wgrad_gemm_ph<F16, ABL, false, false, true>), grid, block, PH_WG_LDS_BYTES, s, a);
}
int main(int argc, char** argv) {
  const int U = argc > 1 ? atoi(argv[1]) : 20650;
  const int iters = argc > 2 ? atoi(argv[2]) : 40;
  const int rounds = argc > 3 ? atoi(argv[3]) : 2;
  const int hot_ic = argc > 4 ? atoi(argv[4]) : 0;
  const int Fp = 4096, Dp = 512, S = 8;
  const int64_t n_rows = 81914;
  const int NSETS = 4;
  const int Rp = (int)round_up(U, 256) + 256;
  uint16_t *table, *dY; float* slabs; int32_t* rows; int32_t* ndev;
  CHK(hipMalloc(&table, (n_rows + 1) * (int64_t)Fp * 2));
  CHK(hipMalloc(&dY, (int64_t)(Rp + BK) * Dp * 2));
  CHK(hipMalloc(&slabs, (int64_t)S * Dp * Fp * 4 + (1 << 20)));      // + room for the stamps behind the last slab
  CHK(hipMalloc(&rows, (int64_t)NSETS * Rp * 4));
  CHK(hipMalloc(&ndev, 64));
  hipLaunchKernelGGL(k_fill16, dim3(4096), dim3(256), 0, 0, table, n_rows * (int64_t)Fp, 7ull, -1.0f);     // (scale < 0: fc7-like -- non-negative, a quarter zeros)
  CHK(hipMemset(table + n_rows * (int64_t)Fp, 0, Fp * 2));
  hipLaunchKernelGGL(k_fill16, dim3(1024), dim3(256), 0, 0, dY, (int64_t)(Rp + BK) * Dp, 13ull, 4096.f);
  CHK(hipMemcpy(ndev, &U, 4, hipMemcpyHostToDevice));
  std::mt19937_64 rng(5);
  std::vector<int32_t> hr((size_t)NSETS * Rp, (int32_t)n_rows);
  {
    std::vector<int32_t> perm(n_rows); for (int64_t i = 0; i < n_rows; ++i) perm[i] = (int32_t)i;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int s = 0; s < NSETS; ++s) for (int i = 0; i < U; ++i) hr[(size_t)s * Rp + i] = hot_ic ? perm[i] : perm[((size_t)s * U + i) % n_rows];
  }
  CHK(hipMemcpy(rows, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
  CHK(hipDeviceSynchronize());
  WgradArgs base{};
  base.dYh = dY; base.table = table; base.slabs = slabs; base.Rp = Rp; base.Dp = Dp; base.Fp = Fp; base.S = S;
  base.ksteps_per_split = (Rp / BK + S - 1) / S; base.n_dev = ndev; base.zero_row = (int32_t)n_rows;
  hipStream_t st; CHK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  const int nwg = (Dp / BM) * (Fp / BN) * S;
  struct Var { const char* name; int kind; };
  const Var vars[] = {{"wgrad", 0}, {"wgrad_ts", 1}, {"wgrad_ts_nostream", 2}, {"wgrad_ts_nomm", 3}, {"wgrad_ts_nostore", 4}, {"wgrad_ts_hot", 5}, {"wgrad_sf", 6}, {"wgrad_ts_sf", 7},
                      {"wgrad_lean", 8}, {"wgrad_lean_ts", 9}, {"wgrad_lean_ts_hot", 10}};
  {   // the lean build against the 64-bit-address build, bit for bit
    std::vector<float> s0((size_t)S * Dp * Fp), s1(s0.size());
    WgradArgs a = base; a.rows = rows;
    launch<0>(a, st); CHK(hipStreamSynchronize(st)); CHK(hipMemcpy(s0.data(), slabs, s0.size() * 4, hipMemcpyDeviceToHost));
    CHK(hipMemsetAsync(slabs, 0xff, s0.size() * 4, st));
    launch_lean<0>(a, st); CHK(hipStreamSynchronize(st)); CHK(hipMemcpy(s1.data(), slabs, s1.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (size_t i = 0; i < s0.size(); ++i) bad += memcmp(&s0[i], &s1[i], 4) != 0;
    printf("check wgrad_lean: %ld of %zu slab values differ\n", bad, s0.size());
  }
  {   // the stream-first build against the product build, bit for bit (slab 0 .. S-1)
    std::vector<float> s0((size_t)S * Dp * Fp), s1(s0.size());
    WgradArgs a = base; a.rows = rows;
    launch<0>(a, st); CHK(hipStreamSynchronize(st)); CHK(hipMemcpy(s0.data(), slabs, s0.size() * 4, hipMemcpyDeviceToHost));
    CHK(hipMemsetAsync(slabs, 0xff, s0.size() * 4, st));
    launch<1024>(a, st); CHK(hipStreamSynchronize(st)); CHK(hipMemcpy(s1.data(), slabs, s1.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0; for (size_t i = 0; i < s0.size(); ++i) bad += memcmp(&s0[i], &s1[i], 4) != 0;
    printf("check wgrad_sf: %ld of %zu slab values differ\n", bad, s0.size());
  }
  int set = 0;
  for (int r = 0; r < rounds; ++r)
    for (const Var& v : vars) {
      auto run = [&]() {
        WgradArgs a = base; a.rows = rows + (size_t)set * Rp; set = (set + 1) % NSETS;
        switch (v.kind) {
          case 0: launch<0>(a, st); break;
          case 1: launch<512>(a, st); break;
          case 2: launch<512 + 1>(a, st); break;
          case 3: launch<512 + 2>(a, st); break;
          case 4: launch<512 + 64>(a, st); break;
          case 5: launch<512 + 8>(a, st); break;
          case 6: launch<1024>(a, st); break;                 // r05: the phase's LDS-DMA in front of its fragment reads
          case 7: launch<1024 + 512>(a, st); break;
          case 8: launch_lean<0>(a, st); break;               // r06: the lean instantiation
          case 9: launch_lean<512>(a, st); break;
          case 10: launch_lean<512 + 8>(a, st); break;
        }
      };
      for (int i = 0; i < 5; ++i) run();
      CHK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) run();
      CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
      printf("round %d  %-18s %8.2f us\n", r, v.name, ms * 1000.f / iters);
      if (v.kind >= 1 && v.kind != 6 && v.kind != 8) {
        std::vector<uint32_t> hb((size_t)nwg * 8 * 12);
        CHK(hipMemcpy(hb.data(), slabs + (int64_t)S * Dp * Fp, hb.size() * 4, hipMemcpyDeviceToHost));
        for (int grp = 0; grp < 2; ++grp) {
          double sum[12] = {0};
          for (int w = 0; w < nwg; ++w) for (int wv = grp * 4; wv < grp * 4 + 4; ++wv)
            for (int j = 0; j < 12; ++j) sum[j] += hb[((size_t)w * 8 + wv) * 12 + j];
          const double n = nwg * 4.0, nph = sum[10] / n * 4.0;
          const double mhz = (sum[9] / n) / ((sum[8] / n) / 100.0);
          printf("   stamps waves %d-%d: load+wait %.0f  bar_after_load %.0f  mfma %.0f  bar_after_mfma %.0f | start->loop %.0f  loop_end %.0f  end %.0f clocks; %.0f MHz; K-tiles %.1f;"
                 " per phase: load %.0f bar1 %.0f mfma %.0f bar2 %.0f\n", grp * 4, grp * 4 + 3, sum[0] / n, sum[2] / n, sum[3] / n, sum[4] / n,
                 sum[5] / n, sum[6] / n, sum[7] / n, mhz, sum[10] / n, sum[0] / n / nph, sum[2] / n / nph, sum[3] / n / nph, sum[4] / n / nph);
        }
      }
      fflush(stdout);
    }
  return 0;
}
