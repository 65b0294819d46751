// slab_pitch_lab: k_reduce_sgd reads eight split-K slabs at the same offset -- eight streams exactly 8 MiB apart.  Does that power-of-two
// distance cost anything in the memory system's interleave?  Sum of S slabs of N floats, slab pitch = N + pad floats; GB/s of slab bytes.
// Build: hipcc --offload-arch=gfx950 -O3 slab_pitch_lab.hip -o slab_pitch_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int S>
__global__ __launch_bounds__(256) void k_sum(const float* slabs, size_t pitch, float* out, size_t n4) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    f4 v[S];
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = __builtin_nontemporal_load((const f4*)(slabs + s * pitch) + i);
    f4 a = v[0];
#pragma unroll
    for (int s = 1; s < S; ++s) a += v[s];
    __builtin_nontemporal_store(a, (f4*)out + i);
  }
}
int main() {
  const size_t N = 512 * 4096;                       // floats per slab (8 MiB)
  const int S = 8;
  const size_t max_pad = 1 << 20;
  float* slabs; CHK(hipMalloc(&slabs, (S * (N + max_pad) + 1024) * 4)); CHK(hipMemset(slabs, 0, (S * (N + max_pad) + 1024) * 4));
  float* out; CHK(hipMalloc(&out, N * 4));
  float* trash; CHK(hipMalloc(&trash, (size_t)512 << 20));      // evict the caches between launches
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
  for (size_t pad : {(size_t)0, (size_t)64, (size_t)1024, (size_t)4096 + 64, (size_t)16384 + 64, (size_t)65536 + 1024 + 64, (size_t)262144 + 4096 + 64}) {
    std::vector<float> t;
    for (int it = 0; it < 12; ++it) {
      CHK(hipMemsetAsync(trash, it, (size_t)512 << 20));
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_sum<S>, dim3(2048), dim3(256), 0, 0, slabs, N + pad, out, N / 4);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      if (it >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("slab pitch 8 MiB + %7zu B: median %6.1f us  min %6.1f us  -> %5.0f GB/s of slab reads (+ %zu MB written)\n", pad * 4, t[t.size() / 2] * 1e3, t[0] * 1e3,
           S * N * 4 / (t[t.size() / 2] * 1e-3) / 1e9, N * 4 >> 20);
  }
  return 0;
}
