// lds_issue_lab: what does ONE LDS fragment read cost the wave that issues it, in the geometry of the phase-staggered GEMMs
// (512-thread workgroup, one per CU; waves 4-7 in a LOAD segment while waves 0-3 run an MFMA segment between the same two barriers)?
// profiles/r05_wgrad_stamps.txt puts k_wgrad_gemm_ph's load segment at ~18 clocks per ds_read_b64_tr_b16 (512 B) -- the same as the
// forward kernel's ds_read_b128 (1 KiB), which is the LDS array's own 256 B/clk with four waves reading.  Why is the half-width read not
// faster, and does the partner's MFMA shape matter (a 16x16x32 MFMA holds its SIMD's issue port 8 clocks of 16, a 32x32x16 one 8 of 32)?
//   read kind : tr = ds_read_b64_tr_b16 (k_wgrad_gemm_ph's addresses) | b128 = ds_read_b128 (k_fwd_gemm_ph's) | b64 = ds_read_b64
//   readers   : waves 4-7 (one per SIMD) | all eight waves
//   partner   : idle at the barrier | 16 x v_mfma_f32_16x16x32_f16 | 8 x v_mfma_f32_32x32x16_f16 (256 clocks of matrix pipe either way)
// A burst = 16 reads from 4 address registers (offsets 0, 1024, 8192, 9216: one X half of k_wgrad_gemm_ph).  Per burst the reader stamps
// s_memtime before the first read, after the last one was ISSUED, and after s_waitcnt lgkmcnt(0).
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-inline-asm lds_issue_lab.hip -o lds_issue_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int REPS = 256;

#define RD4(op, o0, o1, o2, o3, an)                                   \
  op " %[r" #o0 "], %[" #an "]\n\t"                                   \
  op " %[r" #o1 "], %[" #an "] offset:1024\n\t"                       \
  op " %[r" #o2 "], %[" #an "] offset:8192\n\t"                       \
  op " %[r" #o3 "], %[" #an "] offset:9216\n\t"

// one burst of 16 reads; returns (issue clocks, issue + landed clocks)
#define BURST64(op)                                                                                                   \
  {                                                                                                                    \
    unsigned long long t0, t1, t2;                                                                                     \
    unsigned long long r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;                           \
    asm volatile("s_memtime %[t0]\n\t"                                                                                 \
                 RD4(op, 0, 1, 2, 3, a0) RD4(op, 4, 5, 6, 7, a1) RD4(op, 8, 9, 10, 11, a2) RD4(op, 12, 13, 14, 15, a3) \
                 "s_memtime %[t1]\n\t"                                                                                 \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                            \
                 "s_memtime %[t2]\n\t"                                                                                 \
                 "s_waitcnt lgkmcnt(0)"                                                                                \
                 : [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2),                                                     \
                   [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6), [r7] "=&v"(r7), \
                   [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11), [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14), [r15] "=&v"(r15) \
                 : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3) : "memory");                                 \
    sink ^= r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7 ^ r8 ^ r9 ^ r10 ^ r11 ^ r12 ^ r13 ^ r14 ^ r15;                       \
    s_issue += t1 - t0; s_land += t2 - t0;                                                                             \
  }
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define BURST128()                                                                                                     \
  {                                                                                                                    \
    unsigned long long t0, t1, t2;                                                                                     \
    u32x4 r0, r1, r2, r3, r4, r5, r6, r7, r8, r9, r10, r11, r12, r13, r14, r15;                                        \
    asm volatile("s_memtime %[t0]\n\t"                                                                                 \
                 RD4("ds_read_b128", 0, 1, 2, 3, a0) RD4("ds_read_b128", 4, 5, 6, 7, a1) RD4("ds_read_b128", 8, 9, 10, 11, a2) RD4("ds_read_b128", 12, 13, 14, 15, a3) \
                 "s_memtime %[t1]\n\t"                                                                                 \
                 "s_waitcnt lgkmcnt(0)\n\t"                                                                            \
                 "s_memtime %[t2]\n\t"                                                                                 \
                 "s_waitcnt lgkmcnt(0)"                                                                                \
                 : [t0] "=&s"(t0), [t1] "=&s"(t1), [t2] "=&s"(t2),                                                     \
                   [r0] "=&v"(r0), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [r4] "=&v"(r4), [r5] "=&v"(r5), [r6] "=&v"(r6), [r7] "=&v"(r7), \
                   [r8] "=&v"(r8), [r9] "=&v"(r9), [r10] "=&v"(r10), [r11] "=&v"(r11), [r12] "=&v"(r12), [r13] "=&v"(r13), [r14] "=&v"(r14), [r15] "=&v"(r15) \
                 : [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3) : "memory");                                 \
    sink ^= r0[0] ^ r1[1] ^ r2[2] ^ r3[3] ^ r4[0] ^ r5[1] ^ r6[2] ^ r7[3] ^ r8[0] ^ r9[1] ^ r10[2] ^ r11[3] ^ r12[0] ^ r13[1] ^ r14[2] ^ r15[3]; \
    s_issue += t1 - t0; s_land += t2 - t0;                                                                             \
  }

// KIND 0 tr, 1 b128, 2 b64 | ALL8: every wave reads | PARTNER 0 idle, 1 16x16x32, 2 32x32x16
template <int KIND, bool ALL8, int PARTNER>
__global__ __launch_bounds__(512) void k_lds(unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  for (int i = tid; i < 65536 / 4; i += 512) ((unsigned*)smem)[i] = i * 2654435761u;
  __syncthreads();
  unsigned a0, a1, a2, a3;
  if (KIND == 0 || KIND == 2) {
    // k_wgrad_gemm_ph's fragment addresses (64 k-rows x 128 columns, 256-B rows, chunk' = chunk ^ (h(row) << 1))
    const int g = lane >> 4, li = lane & 15, q4 = li >> 2, pp = li & 3;
    const int row = 8 * g + q4;
    const int hx = ((row & 3) | (((row >> 3) & 1) << 2)) << 1;
    const int rd = row * 256 + (pp & 1) * 8;
    a0 = rd + (((wm * 8 + 0 + (pp >> 1)) ^ hx) << 4);
    a1 = rd + (((wm * 8 + 2 + (pp >> 1)) ^ hx) << 4);
    a2 = rd + (((wm * 8 + 4 + (pp >> 1)) ^ hx) << 4);
    a3 = rd + (((wm * 8 + 6 + (pp >> 1)) ^ hx) << 4);
  } else {
    // k_fwd_gemm_ph's: [128 rows][64 halves], 128-B rows, chunk' = chunk ^ (row & 7); four 16-row tiles
    const int frow = lane & 15, fq = lane >> 4, sw = frow & 7;
    const int base = (wn * 32 + frow) * 128 + ((fq ^ sw) << 4);
    a0 = base; a1 = base + 2048; a2 = base + 16384; a3 = base + 16384 + 2048;
  }
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) void*)smem);
  a0 += lds0; a1 += lds0; a2 += lds0; a3 += lds0;
  unsigned long long sink = 0, s_issue = 0, s_land = 0;
  f16x8 fa, fb;
  for (int j = 0; j < 8; ++j) { fa[j] = (_Float16)(0.01f * (lane + j)); fb[j] = (_Float16)(0.02f * (lane - j)); }
  f32x4 c4[8];
  f32x16 c16[2];
  for (int j = 0; j < 8; ++j) c4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int j = 0; j < 2; ++j) for (int k = 0; k < 16; ++k) c16[j][k] = 0.f;
  const bool reader = ALL8 || wm == 1;
  const unsigned long long tk0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
  for (int rep = 0; rep < REPS; ++rep) {
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (reader) {
      if (KIND == 0) BURST64("ds_read_b64_tr_b16")
      else if (KIND == 2) BURST64("ds_read_b64")
      else BURST128()
    } else if (PARTNER == 1) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) c4[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, c4[j], 0, 0, 0);
    } else if (PARTNER == 2) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j) c16[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, c16[j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const unsigned long long tk1 = __builtin_amdgcn_s_memtime();
  const unsigned long long tr1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  for (int j = 0; j < 8; ++j) acc += c4[j][0] + c4[j][3];
  for (int j = 0; j < 2; ++j) acc += c16[j][0] + c16[j][15];
  if (lane == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * 8 + wave) * 6;
    o[0] = s_issue; o[1] = s_land; o[2] = tk1 - tk0; o[3] = tr1 - tr0; o[4] = sink + (unsigned long long)(acc == 12345.f); o[5] = reader;
  }
}

template <int KIND, bool ALL8, int PARTNER>
static void run(const char* name, unsigned long long* dout) {
  CHK(hipFuncSetAttribute((const void*)k_lds<KIND, ALL8, PARTNER>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  std::vector<unsigned long long> h(256 * 8 * 6);
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL((k_lds<KIND, ALL8, PARTNER>), dim3(256), dim3(512), 65536, 0, dout);
    CHK(hipDeviceSynchronize());
  }
  CHK(hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> iss, land, mhz, per;
  for (int b = 0; b < 256; ++b)
    for (int w = 0; w < 8; ++w) {
      const unsigned long long* o = &h[((size_t)b * 8 + w) * 6];
      if (o[5]) { iss.push_back((double)o[0] / REPS / 16); land.push_back((double)o[1] / REPS / 16); }
      if (w == 0) { mhz.push_back((double)o[2] / (double)o[3] * 100.0); per.push_back((double)o[2] / REPS); }
    }
  auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%-34s clocks per read: issue %6.2f  issue+landed %6.2f | clocks per rep (barrier to barrier) %7.1f | %4.0f MHz\n", name, med(iss), med(land), med(per), med(mhz));
}

int main() {
  unsigned long long* dout; CHK(hipMalloc(&dout, 256 * 8 * 6 * 8));
  printf("lds_issue_lab: bursts of 16 LDS reads per wave between two workgroup barriers, 256 workgroups x 512 threads, %d reps; medians over waves\n", REPS);
  run<0, false, 0>("tr_b64   readers 4-7, partner idle", dout);
  run<0, false, 1>("tr_b64   readers 4-7, 16x16x32 x16", dout);
  run<0, false, 2>("tr_b64   readers 4-7, 32x32x16 x8", dout);
  run<0, true, 0>("tr_b64   all 8 waves read", dout);
  run<2, false, 0>("b64      readers 4-7, partner idle", dout);
  run<2, false, 1>("b64      readers 4-7, 16x16x32 x16", dout);
  run<2, true, 0>("b64      all 8 waves read", dout);
  run<1, false, 0>("b128     readers 4-7, partner idle", dout);
  run<1, false, 1>("b128     readers 4-7, 16x16x32 x16", dout);
  run<1, false, 2>("b128     readers 4-7, 32x32x16 x8", dout);
  run<1, true, 0>("b128     all 8 waves read", dout);
  return 0;
}
