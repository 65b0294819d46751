// Cost of the links of a dependent chain on one HIP stream (gfx950): kernel -> tiny kernel, kernel -> hipStreamWriteValue32,
// event record + cross-stream wait.  Build: hipcc --offload-arch=gfx950 -O2 stream_ops.hip -o stream_ops
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(float* p, int n) { for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = p[i] * 1.0001f + 1.f; }
__global__ void k_pub(int* f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float* p; int* f; int n = 1 << 22;
  CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4)); CK(hipMalloc(&f, 64));
  hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  hipEvent_t e1, e2; CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
  const int it = 2000;
  auto run = [&](const char* name, auto body) {
    for (int i = 0; i < 50; ++i) body(i);
    hipDeviceSynchronize();
    const double t0 = now();
    for (int i = 0; i < it; ++i) body(i);
    hipDeviceSynchronize();
    printf("%-60s %.2f us per iteration\n", name, (now() - t0) / it * 1e6);
    return 0;
  };
  run("4 x work kernel (256 blocks)", [&](int) { for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, a, p, n / 4); });
  run("4 x (work + publish kernel)", [&](int i) { for (int k = 0; k < 4; ++k) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, a, p, n / 4); hipLaunchKernelGGL(k_pub, dim3(1), dim3(1), 0, a, f + k, i); } });
  run("4 x (work + hipStreamWriteValue32)", [&](int i) { for (int k = 0; k < 4; ++k) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, a, p, n / 4); if (hipStreamWriteValue32(a, f + k, i, 0) != hipSuccess) { printf("hipStreamWriteValue32 failed\n"); } } });
  run("work on a; record e1; b waits e1; 4 x work on b", [&](int) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, a, p, n / 4); hipEventRecord(e1, a); hipStreamWaitEvent(b, e1, 0); for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, b, p + n / 2, n / 4); hipEventRecord(e2, b); hipStreamWaitEvent(a, e2, 0); });
  run("the same without the join back (a never waits for b)", [&](int) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, a, p, n / 4); hipEventRecord(e1, a); hipStreamWaitEvent(b, e1, 0); for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, b, p + n / 2, n / 4); });
  run("5 x work on a (reference for the two above)", [&](int) { for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, a, p, n / 4); });
  return 0;
}
