// Sanitizer driver for the host sampler (tools/san_sampler.sh): the serial path, the 1/2/3-thread prefetch pipelines and the
// shared-memory ring with two consumer threads must all produce the same index stream; run under ThreadSanitizer (the
// lock-free stage rings, the batch ring) and under AddressSanitizer + UBSan (the bit-parallel walk reads whole vectors).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>
#include "../include/videovec.h"

static uint64_t mix(uint64_t a, uint64_t b) { uint64_t x = a * 0x9E3779B97F4A7C15ull + b; x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29; return x; }

int main() {
  const int V = 300;
  std::vector<int32_t> vid(V), ns(V); std::vector<int64_t> rb(V);
  int64_t rows = 0;
  for (int v = 0; v < V; ++v) { vid[v] = 1000 + v; ns[v] = 1 + (int)(mix(7, v) % 70); rb[v] = rows; rows += ns[v]; }
  int failures = 0;
  const int ctypes[] = {VV_CONTEXT_WINDOW, VV_CONTEXT_PAST, VV_CONTEXT_PAST_CONTINUOUS, VV_CONTEXT_PAIRWISE};
  for (int ct : ctypes) for (int same = 0; same <= 4; same += 4) {
    vv_sampler_param p; vv_sampler_param_default(&p);
    p.batch_size = 64; p.context_size = 5; p.num_negative_samples = 12; p.max_buffer_size = 400; p.context_type = ct;
    p.max_same_video_negs = ct == VV_CONTEXT_PAIRWISE ? 0 : same;
    const int CN = (ct == VV_CONTEXT_PAIRWISE ? 2 : 5) + 12, n = 64 * CN;
    vv_sampler* ref = nullptr;
    if (vv_sampler_create(&p, V, vid.data(), ns.data(), rb.data(), nullptr, &ref)) { printf("create failed\n"); return 2; }
    std::vector<std::vector<int32_t>> want(40, std::vector<int32_t>(n));
    for (auto& w : want) vv_sampler_next(ref, w.data(), nullptr, nullptr);
    vv_sampler_destroy(ref);
    for (int threads = 1; threads <= 4; ++threads) {
      vv_sampler* s = nullptr;
      vv_sampler_create(&p, V, vid.data(), ns.data(), rb.data(), nullptr, &s);
      vv_sampler_prefetch_start(s, 3, threads, nullptr, 1);
      std::vector<int32_t> got(n);
      for (auto& w : want) { vv_sampler_next(s, got.data(), nullptr, nullptr); if (got != w) ++failures; }
      vv_sampler_prefetch_stop(s);
      vv_sampler_destroy(s);
    }
    {   // shared-memory ring, two consumers each taking half of every batch
      char name[64]; snprintf(name, sizeof name, "vv_san_%d_%d_%d", (int)getpid(), ct, same);
      vv_sampler* s = nullptr;
      vv_sampler_create(&p, V, vid.data(), ns.data(), rb.data(), nullptr, &s);
      vv_sampler_prefetch_start(s, 4, 4, name, 2);
      int bad[2] = {0, 0};
      auto consumer = [&](int c) {
        vv_batch_ring* r = nullptr;
        if (vv_batch_ring_attach(name, 10.0, &r)) { bad[c] = 1000; return; }
        std::vector<int32_t> half(32 * CN);
        for (auto& w : want) {
          if (vv_batch_ring_next(r, c, c * 32, 32, half.data(), nullptr, 10.0)) { bad[c] += 100; break; }
          if (memcmp(half.data(), w.data() + (size_t)c * 32 * CN, half.size() * 4)) ++bad[c];
        }
        vv_batch_ring_detach(r);
      };
      std::thread t0(consumer, 0), t1(consumer, 1);
      t0.join(); t1.join();
      failures += bad[0] + bad[1];
      vv_sampler_prefetch_stop(s);
      vv_sampler_destroy(s);
    }
  }
  printf("sampler sanitizer run: %d mismatches\n", failures);
  return failures ? 1 : 0;
}
