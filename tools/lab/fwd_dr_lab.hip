// fwd_dr_lab: k_fwd_gemm_dr (W fragments straight into registers, kernels_gemm_dr.hip) against k_fwd_gemm_ph (both operands
// through LDS, kernels_gemm_ph.hip) at the benchmark's de-duplicated size: U distinct rows gathered from a 671 MB table,
// 4096 -> 512.  Row sets rotate between launches (4 sets x 169 MB: a launch does not find its rows in the Infinity Cache).
// Outputs are compared bit for bit.  Build: see tools/lab/fwd_dr_lab.sh
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
#include "kernels_gemm_dr.hip"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
using namespace vv;

__global__ void k_fill_table(uint16_t* t, int64_t n, uint64_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix64(seed, (uint64_t)i);
    const float v = (h & 3) ? (float)((h >> 8) & 0xffff) * (1.0f / 65536.f) : 0.f;   // fc7-like: non-negative, a quarter zeros
    t[i] = F16::from_float(v);
  }
}
__global__ void k_fill_w(uint16_t* w, uint16_t* wq, int D, int Fp, int nk, uint64_t seed) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < (int64_t)D * Fp; i += (int64_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / Fp), k = (int)(i % Fp);
    const uint64_t h = mix64(seed, (uint64_t)i);
    const float v = ((float)((h >> 8) & 0xffff) * (1.0f / 32768.f) - 1.0f) * 2048.f;
    const uint16_t x = F16::from_float(v);
    w[i] = x; wq[wq_index(n, k, nk)] = x;
  }
}

// r05: the product forward kernel with time stamps (ABL bit 9, kernels_gemm_ph.hip): 192-row tiles, with / without the sibling lead
template <int ABL, int LEAD, int MRG = 0>
static void launch_ph_ts(const FwdArgs& a, hipStream_t s) {
  constexpr int LDS = LEAD ? 10 * PH_SLOT : PH_LDS_BYTES;
  static bool once = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_ph<F16, 0, true, 3, ABL, false, 0, LEAD, MRG>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
  (void)once;
  const dim3 grid(((a.R + 191) / 192) * 2), block(GEMM_THREADS);
  hipLaunchKernelGGL((k_fwd_gemm_ph<F16, 0, true, 3, ABL, false, 0, LEAD, MRG>), grid, block, LDS, s, a);
}

template <int MT, int P, int ABL, int TD = 0, int OPT = 0>
static void launch_dr(const FwdArgs& a, hipStream_t s) {
  constexpr int LDS = (P + 1) * 16 * MT * 128 + (TD ? 2048 : 0);
  static bool once = ((void)hipFuncSetAttribute((const void*)k_fwd_gemm_dr<F16, MT, P, ABL, TD, true, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS), true);
  (void)once;
  const int Dp = (int)round_up(a.D, D_ALIGN);
  const int nrt = (a.R + 16 * MT - 1) / (16 * MT), tn = Dp / BN;
  const dim3 grid((OPT & 2) ? 8 * ((nrt + 8 / tn - 1) / (8 / tn)) : nrt * tn), block(GEMM_THREADS);
  hipLaunchKernelGGL((k_fwd_gemm_dr<F16, MT, P, ABL, TD, true, OPT>), grid, block, LDS, s, a);
}

int main(int argc, char** argv) {
  const int U = argc > 1 ? atoi(argv[1]) : 20650;
  const int iters = argc > 2 ? atoi(argv[2]) : 40;
  const int rounds = argc > 3 ? atoi(argv[3]) : 3;
  const char* only = argc > 4 ? argv[4] : "";
  const int hot_ic = argc > 5 ? atoi(argv[5]) : 0;     // 1: one row set from the first U table rows, re-used by every launch (Infinity-Cache resident)
  const int F = 4096, D = 512, Fp = 4096, Dp = 512, nk = Fp / 64;
  const int64_t n_rows = 81914;
  const int NSETS = 4;
  uint16_t *table, *Wh, *Wq; float *bias, *H0, *H1; Scales* sc; int32_t* rows;
  CHK(hipMalloc(&table, (n_rows + 1) * (int64_t)Fp * 2));
  CHK(hipMalloc(&Wh, (int64_t)Dp * Fp * 2)); CHK(hipMalloc(&Wq, (int64_t)Dp * Fp * 2));
  CHK(hipMalloc(&bias, D * 4)); CHK(hipMalloc(&sc, sizeof(Scales)));
  const int Rp = (int)round_up(U, 256) + 256;
  CHK(hipMalloc(&H0, (int64_t)Rp * D * 4)); CHK(hipMalloc(&H1, (int64_t)Rp * D * 4));
  CHK(hipMalloc(&rows, (int64_t)NSETS * Rp * 4));
  hipLaunchKernelGGL(k_fill_table, dim3(4096), dim3(256), 0, 0, table, n_rows * (int64_t)Fp, 7ull);
  CHK(hipMemset(table + n_rows * (int64_t)Fp, 0, Fp * 2));
  hipLaunchKernelGGL(k_fill_w, dim3(1024), dim3(256), 0, 0, Wh, Wq, D, Fp, nk, 11ull);
  std::vector<float> hb(D); for (int i = 0; i < D; ++i) hb[i] = 0.01f * (i % 17) - 0.05f;
  CHK(hipMemcpy(bias, hb.data(), D * 4, hipMemcpyHostToDevice));
  Scales hs; hs.sx = 1.f; hs.sw_cur = 2048.f; hs.sw_next = 2048.f; hs.wmax_bits = 0;
  CHK(hipMemcpy(sc, &hs, sizeof(hs), hipMemcpyHostToDevice));
  std::mt19937_64 rng(5);
  std::vector<int32_t> hr((size_t)NSETS * Rp, (int32_t)n_rows);
  {
    std::vector<int32_t> perm(n_rows); for (int64_t i = 0; i < n_rows; ++i) perm[i] = (int32_t)i;
    std::shuffle(perm.begin(), perm.end(), rng);
    for (int s = 0; s < NSETS; ++s) for (int i = 0; i < U; ++i) hr[(size_t)s * Rp + i] = perm[((size_t)s * U + i) % n_rows];
    if (hot_ic) {
      std::vector<int32_t> p2(U); for (int i = 0; i < U; ++i) p2[i] = i;
      std::shuffle(p2.begin(), p2.end(), rng);
      for (int s = 0; s < NSETS; ++s) for (int i = 0; i < U; ++i) hr[(size_t)s * Rp + i] = p2[i];
    }
  }
  CHK(hipMemcpy(rows, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
  CHK(hipDeviceSynchronize());

  FwdArgs base{};
  base.table = table; base.Wh = Wh; base.bias = bias; base.scales = sc; base.R = U; base.D = D; base.Fp = Fp;
  base.zero_row = (int32_t)n_rows; base.relu = 1; base.drop_ratio = 0.f; base.mask = nullptr; base.B = 1; base.CN = 1;
  hipStream_t st; CHK(hipStreamCreate(&st));
  const int ts_wg = ((U + 191) / 192) * 2;
  uint32_t* ts_buf; CHK(hipMalloc(&ts_buf, (size_t)ts_wg * 8 * 12 * 4)); CHK(hipMemset(ts_buf, 0, (size_t)ts_wg * 8 * 12 * 4));
  struct Var { const char* name; int kind; };
  std::vector<Var> vars = {{"ph_lead", 0}, {"dr12p4", 1}, {"dr12p2", 2}, {"dr12p4_noA", 3}, {"dr12p4_noMM", 4}, {"dr12p4_noRD", 5},
                           {"dr12p4_hotA", 6}, {"dr12p4_noB", 7}, {"dr12p4_noAnoB", 8}, {"dr12p4_onlyMM", 9}, {"dr16p4", 10}, {"dr8p4", 11},
                           {"dr12_onlyA", 12}, {"dr12_onlyB", 13}, {"dr12_onlyAB", 14}, {"dr12_t4", 15}, {"dr12_t8", 16}, {"dr12_lgkm", 17},
                           {"dr12_t8_onlyA", 18}, {"dr12_t8_noB", 19},
                           {"dr12_xcd", 20}, {"dr12_xcd_onlyA", 21}, {"dr12_sc1", 22}, {"dr12_nt", 23}, {"dr12_sc01", 24}, {"dr12_sc1_onlyA", 25}, {"dr12_xcd_onlyB", 26},
                           {"dr12_run256_onlyA", 27}, {"dr12_run256_onlyAB", 28}, {"dr12_run256_full", 29},
                           {"dr12_rot1", 30}, {"dr12_rot2", 31}, {"dr12_rot4", 32}, {"dr12_rot1_onlyA", 33}, {"dr12_rot2_onlyA", 34}, {"dr12_rot4_onlyA", 35},
                           {"dr12_run512_onlyA", 36}, {"dr12_run1k_onlyA", 37},
                           {"dr12_bl1_onlyA", 38}, {"dr12_bl2_onlyA", 39}, {"dr12_bl4_onlyA", 40}, {"dr12_bl2_onlyAB", 41}, {"dr12_bl2_noRD", 42},
                           {"dr12_spread_onlyA", 43}, {"dr12_spread_full", 44}, {"dr12_spread_noRD", 45}, {"dr12_spread_run1k_onlyA", 46},
                           {"dr12_even_onlyA", 47}, {"dr12_even_onlyAB", 48}, {"dr12_split_onlyA", 49}, {"dr12_split_onlyAB", 50},
                           {"ph_ts_lead", 60}, {"ph_ts_plain", 61}, {"ph_ts_lead_hotA", 62}, {"ph_ts_lead_nostream", 63}, {"ph_ts_lead_nomm", 64},
                           {"ph_ts_lead_nostore", 65}, {"ph_plain", 66}, {"ph_sf_plain", 67}, {"ph_sf_lead", 68}, {"ph_lead2", 69},
                           {"marks_lead_4ph", 70}, {"marks_lead_merged", 71}, {"marks_plain_4ph", 72}, {"marks_plain_merged", 73},
                           {"marks_lead_4ph_hotA", 74}, {"marks_lead_merged_hotA", 75}, {"marks_lead_4ph_nostream", 76}, {"marks_lead_merged_nostream", 77}};
  auto run = [&](int kind, int set, float* Hout) {
    FwdArgs a = base; a.rows = rows + (size_t)set * Rp; a.H = Hout;
    if (kind == 0) { a.Wh = Wh; launch_fwd_gemm_ph(0, a, st); return; }
    if (kind >= 60) {
      a.Wh = Wh; a.mask = (const uint8_t*)ts_buf;
      switch (kind) {
        case 60: launch_ph_ts<512, 1>(a, st); break;
        case 61: launch_ph_ts<512, 0>(a, st); break;
        case 62: launch_ph_ts<512 + 8, 1>(a, st); break;
        case 63: launch_ph_ts<512 + 1, 1>(a, st); break;
        case 64: launch_ph_ts<512 + 2, 1>(a, st); break;
        case 65: launch_ph_ts<512 + 64, 1>(a, st); break;
        case 66: launch_ph_ts<0, 0>(a, st); break;
        case 67: launch_ph_ts<1024, 0>(a, st); break;       // r05: the phase's LDS-DMA in front of its fragment reads
        case 68: launch_ph_ts<1024, 1>(a, st); break;
        case 69: launch_ph_ts<0, 1>(a, st); break;
        case 70: launch_ph_ts<2048, 1, 0>(a, st); break;     // r05: the kernel's four marks only; four phases per K-tile ...
        case 71: launch_ph_ts<2048, 1, 1>(a, st); break;     // ... and two merged ones
        case 72: launch_ph_ts<2048, 0, 0>(a, st); break;
        case 73: launch_ph_ts<2048, 0, 1>(a, st); break;
        case 74: launch_ph_ts<2048 + 8, 1, 0>(a, st); break; // rows = the L2-hot zero row
        case 75: launch_ph_ts<2048 + 8, 1, 1>(a, st); break;
        case 76: launch_ph_ts<2048 + 1, 1, 0>(a, st); break; // no LDS-DMA stream in the loop
        case 77: launch_ph_ts<2048 + 1, 1, 1>(a, st); break;
      }
      return;
    }
    a.Wh = Wq;
    switch (kind) {
      case 1: launch_dr<12, 4, 0>(a, st); break;
      case 2: launch_dr<12, 2, 0>(a, st); break;
      case 3: launch_dr<12, 4, 1>(a, st); break;
      case 4: launch_dr<12, 4, 2>(a, st); break;
      case 5: launch_dr<12, 4, 4>(a, st); break;
      case 6: launch_dr<12, 4, 8>(a, st); break;
      case 7: launch_dr<12, 4, 16>(a, st); break;
      case 8: launch_dr<12, 4, 17>(a, st); break;
      case 9: launch_dr<12, 4, 21>(a, st); break;
      case 10: launch_dr<16, 4, 0>(a, st); break;
      case 11: launch_dr<8, 4, 0>(a, st); break;
      case 12: launch_dr<12, 4, 2 + 4 + 16>(a, st); break;
      case 13: launch_dr<12, 4, 2 + 4 + 1>(a, st); break;
      case 14: launch_dr<12, 4, 2 + 4>(a, st); break;
      case 15: launch_dr<12, 4, 0, 4>(a, st); break;
      case 16: launch_dr<12, 4, 0, 8>(a, st); break;
      case 17: launch_dr<12, 4, 0, 0, 1>(a, st); break;
      case 18: launch_dr<12, 4, 2 + 4 + 16, 8>(a, st); break;
      case 19: launch_dr<12, 4, 16, 8>(a, st); break;
      case 20: launch_dr<12, 4, 0, 0, 1 + 2>(a, st); break;
      case 21: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 2>(a, st); break;
      case 22: launch_dr<12, 4, 0, 0, 1 + 4>(a, st); break;
      case 23: launch_dr<12, 4, 0, 0, 1 + 8>(a, st); break;
      case 24: launch_dr<12, 4, 0, 0, 1 + 12>(a, st); break;
      case 25: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 4>(a, st); break;
      case 26: launch_dr<12, 4, 2 + 4 + 1, 0, 1 + 2>(a, st); break;
      case 27: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 16>(a, st); break;
      case 28: launch_dr<12, 4, 2 + 4, 0, 1 + 16>(a, st); break;
      case 29: launch_dr<12, 4, 0, 0, 1 + 16>(a, st); break;
      case 30: launch_dr<12, 4, 0, 0, 1 + 32>(a, st); break;
      case 31: launch_dr<12, 4, 0, 0, 1 + 64>(a, st); break;
      case 32: launch_dr<12, 4, 0, 0, 1 + 96>(a, st); break;
      case 33: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 32>(a, st); break;
      case 34: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 64>(a, st); break;
      case 35: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 96>(a, st); break;
      case 36: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 16 + 128>(a, st); break;
      case 37: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 16 + 256>(a, st); break;
      case 38: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 512>(a, st); break;
      case 39: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 1024>(a, st); break;
      case 40: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 1536>(a, st); break;
      case 41: launch_dr<12, 4, 2 + 4, 0, 1 + 1024>(a, st); break;
      case 42: launch_dr<12, 4, 4, 0, 1 + 1024>(a, st); break;
      case 43: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 2048>(a, st); break;      // gathered rows alone, K spread over the row tiles
      case 44: launch_dr<12, 4, 0, 0, 1 + 2048>(a, st); break;               // full kernel, K spread
      case 45: launch_dr<12, 4, 4, 0, 1 + 2048>(a, st); break;               // both streams, no fragment reads
      case 46: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 2048 + 16 + 256>(a, st); break;   // rows alone, K spread, 1-KiB runs
      case 47: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 4096>(a, st); break;      // r05: rows alone, only the even column tile asks (no duplicates, 108 pullers)
      case 48: launch_dr<12, 4, 2 + 4, 0, 1 + 4096>(a, st); break;           //      ... + W on every workgroup
      case 49: launch_dr<12, 4, 2 + 4 + 16, 0, 1 + 8192>(a, st); break;      // r05: rows alone, every sibling only its half of the rows (no duplicates, 216 pullers)
      case 50: launch_dr<12, 4, 2 + 4, 0, 1 + 8192>(a, st); break;           //      ... + W
    }
  };
  // correctness: the real variants against the LDS kernel, bit for bit, on every row set
  std::vector<float> h0((size_t)U * D), h1((size_t)U * D);
  for (int kind : {1, 17, 67, 68, 69}) {
    long bad = 0; double maxd = 0;
    for (int set = 0; set < NSETS; ++set) {
      CHK(hipMemsetAsync(H0, 0xff, (int64_t)Rp * D * 4, st)); CHK(hipMemsetAsync(H1, 0xff, (int64_t)Rp * D * 4, st));
      run(0, set, H0); run(kind, set, H1);
      CHK(hipStreamSynchronize(st));
      CHK(hipMemcpy(h0.data(), H0, h0.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(h1.data(), H1, h1.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < h0.size(); ++i) if (memcmp(&h0[i], &h1[i], 4)) { ++bad; maxd = std::max(maxd, (double)fabsf(h0[i] - h1[i])); }
    }
    double s = 0; for (size_t i = 0; i < h0.size(); ++i) s += h0[i];
    const char* nm = "?"; for (auto& v : vars) if (v.kind == kind) nm = v.name;
    printf("check %-10s: %ld of %zu x %d values differ (max |d| %.3g), mean out %.4f\n", nm, bad, h0.size(), NSETS, maxd, s / h0.size());
  }
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  int set = 0;
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) {
      if (only[0] && !strstr(only, v.name)) continue;
      for (int i = 0; i < 5; ++i) { run(v.kind, set, H1); set = (set + 1) % NSETS; }
      CHK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; ++i) { run(v.kind, set, H1); set = (set + 1) % NSETS; }
      CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
      float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
      printf("round %d  %-16s %8.2f us\n", r, v.name, ms * 1000.f / iters);
      if ((v.kind >= 60 && v.kind <= 65) || (v.kind >= 70 && v.kind <= 77)) {
        // the stamps of the LAST launch: mean over the workgroups, waves 0-3 (the leading group) and 4-7 apart
        std::vector<uint32_t> hb((size_t)ts_wg * 8 * 12);
        CHK(hipMemcpy(hb.data(), ts_buf, hb.size() * 4, hipMemcpyDeviceToHost));
        for (int grp = 0; grp < 2; ++grp) {
          double sum[12] = {0}; double mx7 = 0, mn7 = 1e18;
          for (int w = 0; w < ts_wg; ++w) for (int wv = grp * 4; wv < grp * 4 + 4; ++wv) {
            const uint32_t* o = &hb[((size_t)w * 8 + wv) * 12];
            for (int j = 0; j < 12; ++j) sum[j] += o[j];
            mx7 = std::max(mx7, (double)o[7]); mn7 = std::min(mn7, (double)o[7]);
          }
          const double n = ts_wg * 4.0;
          const double mhz = (sum[9] / n) / ((sum[8] / n) / 100.0);           // shader clocks per us of real time
          printf("   stamps waves %d-%d: load %.0f  vmwait %.0f  bar_after_load %.0f  mfma %.0f  bar_after_mfma %.0f  | prologue %.0f  loop_end %.0f  end %.0f (min %.0f max %.0f) clocks;  %.0f MHz;"
                 "  per phase (256): load %.0f vm %.0f bar1 %.0f mfma %.0f bar2 %.0f\n", grp * 4, grp * 4 + 3,
                 sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n, sum[5] / n, sum[6] / n, sum[7] / n, mn7, mx7, mhz,
                 sum[0] / n / 256, sum[1] / n / 256, sum[2] / n / 256, sum[3] / n / 256, sum[4] / n / 256);
        }
      }
      fflush(stdout);
    }
  return 0;
}
