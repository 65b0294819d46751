// Lab: the sampler's walk stage alone (select_item + swap_item on the calling thread), ns per item, and its parts.
//   g++ -O3 -march=x86-64-v3 -std=c++17 -pthread -I include tools/lab/walk_bench.cc -o /tmp/walk_bench
#include "../../videovector_amd/csrc/sampler.cc"
#include <chrono>
#include <cstdio>
int main(int argc, char** argv) {
  const int V = getenv("NV") ? atoi(getenv("NV")) : 2048, items = argc > 1 ? atoi(argv[1]) : 2000000, Nn = argc > 2 ? atoi(argv[2]) : 50;
  std::vector<int32_t> vid(V), ns(V); std::vector<int64_t> rb(V);
  uint64_t z = 88172645463325252ull; int64_t tot = 0;
  for (int v = 0; v < V; ++v) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; vid[v] = v; ns[v] = getenv("FIXN") ? atoi(getenv("FIXN")) : 16 + (int)(z % 49); rb[v] = tot; tot += ns[v]; }
#ifdef VV_WALK_LAB
  g_lab = getenv("LAB") ? atoi(getenv("LAB")) : 0;
#endif
  vv_sampler_param p; vv_sampler_param_default(&p);
  p.batch_size = 1024; p.context_size = 5; p.num_negative_samples = Nn; p.max_buffer_size = 5000; p.negative_swap_percentage = 50;
  vv_sampler* s = nullptr;
  if (vv_sampler_create(&p, V, vid.data(), ns.data(), rb.data(), nullptr, &s) != VV_OK) { printf("create failed\n"); return 1; }
  if (const char* e = getenv("PIN")) {        // "a,b": this thread on CPU a, the stream thread on CPU b
    const int a = atoi(e), b = strchr(e, ',') ? atoi(strchr(e, ',') + 1) : -1;
    cpu_set_t sa; CPU_ZERO(&sa); CPU_SET(a, &sa); sched_setaffinity(0, sizeof(sa), &sa);
    if (b >= 0) { static cpu_set_t sb; CPU_ZERO(&sb); CPU_SET(b, &sb); s->rng.pin_helper_like_caller(&sb); }
  }
  if (getenv("HELPER")) s->rng.start_helper();       // the stream (and its side arrays) from a second thread, as in the 4-thread pipeline
  uint32_t* rec = s->rec1.data();
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
#ifdef VV_WALK_PROF
    memset(g_wp, 0, sizeof(g_wp));
    for (int i = 0; i < items; ++i) { WP(0, s->select_item(rec)); WP(3, s->swap<false>(rec, s->buf_row.data(), nullptr, 0, nullptr)); }
    printf("generation: %.2f ticks per word (%.0f words per item)\n", (double)g_wp[6] / (double)g_wp[7], (double)g_wp[7] / items);
    printf("ticks per item: select %.0f (ensure %.0f)  swap %.0f (prefix %.0f  positions %.0f  taken loop %.0f)\n", (double)g_wp[0] / items, (double)g_wp[4] / items, (double)g_wp[3] / items, (double)g_wp[1] / items, (double)g_wp[5] / items, (double)g_wp[2] / items);
#else
    if (getenv("LOGEV")) {                    // with the event log and a record ring, as the pipeline's walk thread writes them (nobody reads)
      static std::vector<vv_sampler::Event> evr(1 << 20); static std::vector<uint32_t> recs((size_t)32768 * s->rec_words);
      uint64_t evh = 0;
      for (int i = 0; i < items; ++i) { uint32_t* rc = recs.data() + (size_t)(i % 32768) * s->rec_words; s->select_item(rc); s->swap<true>(rc, s->buf_row.data(), evr.data(), evr.size() - 1, &evh); }
    } else
    for (int i = 0; i < items; ++i) { s->select_item(rec); s->swap<false>(rec, s->buf_row.data(), nullptr, 0, nullptr); }
#endif
    auto t1 = std::chrono::steady_clock::now();
    printf("walk: %.1f ns per item (restarts %lld)\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / items, (long long)s->stat_restarts);
  }
  {
    std::vector<int32_t> out(64);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < items; ++i) { rec[4 + s->CA + (i % Nn)] += 0x9E3779B9u; s->negs(rec, out.data(), s->buf_row.data()); }
    auto t1 = std::chrono::steady_clock::now();
    printf("slot draw alone: %.1f ns per item\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / items);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < items; ++i) { rec[4 + (i % 5)] += 0x9E3779B9u; rec[0] = i % V; s->frames_item(rec, out.data(), nullptr); }
    t1 = std::chrono::steady_clock::now();
    printf("frame draw alone: %.1f ns per item\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / items);
    std::vector<int32_t> idx(1024 * (5 + Nn));
    t0 = std::chrono::steady_clock::now();
    for (int b = 0; b < items / 1024; ++b) vv_sampler_next(s, idx.data(), nullptr, nullptr);
    t1 = std::chrono::steady_clock::now();
    printf("calling-thread path: %.1f ns per item\n", std::chrono::duration<double, std::nano>(t1 - t0).count() / (items / 1024 * 1024));
  }
  uint64_t chk = 0; for (int i = 0; i < 5000; ++i) chk = chk * 1000003u + (uint32_t)s->buf_row[i];
  printf("buffer checksum %016llx cursor %d\n", (unsigned long long)chk, s->cursor);
  if (getenv("HELPER")) s->rng.stop_helper();
  vv_sampler_destroy(s);
  return 0;
}
