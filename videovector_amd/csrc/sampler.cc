// sampler.cc -- host-side triplet sampler of the MI355X videovec path (product code; the test
// oracle under oracle/ is a separate, independent restatement and is never linked here).
//
// Reproduces, with table-row indices in place of feature copies, the reference's
// VideoSampledShotsDataLayer:
//   setup   src/caffe/layers/video_sampled_shots_data_layer.cpp:64-369
//   batch   ...:768-909 (InternalThreadEntry), :371-393,425-757 (AddSamplesToTop), :24-44
//   prefetch src/caffe/layers/base_data_layer.cpp:52-95 (a batch is produced while the previous one is consumed)
// including the exact consumption order of the C library's rand() stream.
//
// Structure.  The reference's sampler is ONE sequential stream: every draw is rand() % k, the number of draws an
// item consumes depends on the data (how many of its video's shots are swapped into the negative buffer), and the
// negative slots come from a persistent permutation.  Two things make it fast here without changing a single index:
//   * the libc stream x[k] = x[k-31] + x[k-3] does not depend on how it is consumed, so it is generated a block at a
//     time together with the swap-in predicate ((x >> 1) % 100 < negative_swap_percentage) of every value;
//   * the work of an item splits into three chains that touch disjoint state --
//       walk  : which stream positions the item owns + the swap-in of its video's shots into the buffer
//               (state: stream position, DB cursor, buffer contents and the membership bitmap),
//       negs  : the partial Fisher-Yates draw of its negative slots (state: the persistent slot permutation),
//       frames: its target/context frames (no state at all) --
//     which run either back to back in the calling thread (vv_sampler_next) or as a three-stage pipeline of threads
//     that hands finished batches to the consumer(s) through a ring (vv_sampler_prefetch_start), optionally in POSIX
//     shared memory so that ONE sampler serves every rank of a node.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <fcntl.h>
#include <immintrin.h>
#include <sched.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/videovec.h"

#ifdef VV_WALK_PROF
extern uint64_t g_wp[8];
#endif
namespace {

#define VV_T512 __attribute__((target("avx512f,avx512bw,avx512dq,avx512vl,avx512vbmi2,bmi,bmi2,lzcnt,popcnt")))
VV_T512 static inline uint32_t swap_flags16v(__m512i words, __m512i vsw) {      // ((word >> 1) % 100 < vsw) of sixteen stream words, as mask bits
  const __m512i t = _mm512_srli_epi32(words, 1);
  const __m512i magic = _mm512_set1_epi32(0x51EB851F);
  const __m512i pe = _mm512_srli_epi64(_mm512_mul_epu32(t, magic), 37);                           // even lanes: t / 100
  const __m512i po = _mm512_srli_epi64(_mm512_mul_epu32(_mm512_srli_epi64(t, 32), magic), 37);   // odd lanes
  const __m512i q = _mm512_or_si512(pe, _mm512_slli_epi64(po, 32));
  const __m512i r = _mm512_sub_epi32(t, _mm512_mullo_epi32(q, _mm512_set1_epi32(100)));
  return (uint32_t)_mm512_cmplt_epu32_mask(r, vsw);
}
// (a >> 1) % d of sixteen stream words, d < 2^31 (see above)
VV_T512 static inline __m512i mod16(__m512i words, __m512d vinv, __m256i vd) {
  const __m512i a = _mm512_srli_epi32(words, 1);
  const __m256i a0 = _mm512_castsi512_si256(a), a1 = _mm512_extracti64x4_epi64(a, 1);
  const __m256i q0 = _mm512_cvttpd_epi32(_mm512_mul_pd(_mm512_cvtepi32_pd(a0), vinv));
  const __m256i q1 = _mm512_cvttpd_epi32(_mm512_mul_pd(_mm512_cvtepi32_pd(a1), vinv));
  __m256i r0 = _mm256_sub_epi32(a0, _mm256_mullo_epi32(q0, vd)), r1 = _mm256_sub_epi32(a1, _mm256_mullo_epi32(q1, vd));
  r0 = _mm256_mask_sub_epi32(r0, _mm256_cmpeq_epi32_mask(r0, vd), r0, vd);
  r1 = _mm256_mask_sub_epi32(r1, _mm256_cmpeq_epi32_mask(r1, vd), r1, vd);
  return _mm512_inserti64x4(_mm512_castsi256_si512(r0), r1, 1);
}

// glibc rand() == random() with the default TYPE_3 state: 31 words seeded from seed 1 by the
// 16807 Lehmer recurrence, then state[f] += state[f-3] walking f cyclically; output is the new
// word >> 1; the first 310 outputs are discarded by srandom.  (glibc 2.35 stdlib/random_r.c)
class LibcRand {
 public:
  void init(int block, int swap_pct, uint32_t seed = 1, bool wide = false, int mb = 0) {
    wide_ = wide;
    if (const char* e = getenv("VV_SAMPLER_PREFETCH")) pf_ = atoi(e);
    aux_ = wide && mb > 0 && swap_pct > 0; mb_ = mb; inv_mb_ = mb > 0 ? 1.0 / (double)mb : 0.0;
    uint32_t st[31];
    if (seed == 0) seed = 1;             // srandom_r: seed 0 is seed 1 (the never-seeded stream)
    int64_t w = seed;
    st[0] = seed;
    for (int i = 1; i < 31; ++i) {
      w = (16807 * w) % 2147483647;     // exact in 64 bits; equals glibc's overflow-free form
      st[i] = (uint32_t)w;
    }
    int f = 3, r = 0;
    for (int i = 0; i < 310; ++i) {     // srandom discards the first 310 outputs
      st[f] += st[r];
      f = f == 30 ? 0 : f + 1;
      r = r == 30 ? 0 : r + 1;
    }
    // Linear history: the word at st[f] is the oldest, x[k-31].  From here on x[k] = x[k-31] + x[k-3].
    // Substituting the recurrence into itself twice gives x[k] = x[k-9] + x[k-31] + x[k-34] + x[k-37]: no operand
    // closer than 9 positions, so eight values are produced per vector step instead of one per store-to-load round trip.
    n_ = block; swap_ = (uint32_t)swap_pct;
    h_.assign((size_t)kHist + n_ + 8, 0u);
    if (aux_) { hf_.assign((size_t)kHist + n_ + 64, 0); hp_.assign((size_t)kHist + n_ + 16, 0); }
    uint32_t* h = h_.data();
    for (int i = 0; i < 31; ++i) h[kHist - 31 + i] = st[(f + i) % 31];
    for (int i = kHist - 32; i >= 0; --i) h[i] = h[i + 31] - h[i + 28];     // run the recurrence backwards for the extra history
    generate(h_.data() + kHist, aux_ ? hf_.data() + kHist : nullptr, aux_ ? hp_.data() + kHist : nullptr, 0);
    base_ = h_.data() + kHist; pos_ = 0; avail_ = n_;
    if (aux_) { basef_ = hf_.data() + kHist; basep_ = hp_.data() + kHist; }
  }
  ~LibcRand() { if (helper_on_) { helper_stop_.store(1); if (helper_.joinable()) helper_.join(); } }
  int capacity() const { return n_; }
  // make the next k values addressable as peek(0..k-1) (k <= capacity())
  void ensure(int k) {
    if (pos_ + k <= avail_) return;
    if (helper_on_) { switch_block(); return; }
    const int m = n_ - pos_;                       // unconsumed values
    memmove(h_.data(), h_.data() + pos_, (size_t)(kHist + m) * sizeof(uint32_t));
    if (aux_) { memmove(hf_.data(), hf_.data() + pos_, (size_t)(kHist + m)); memmove(hp_.data(), hp_.data() + pos_, (size_t)(kHist + m) * sizeof(int32_t)); }
#ifdef VV_WALK_PROF
    const uint64_t tg_ = __rdtsc();
#endif
    generate(h_.data() + kHist, aux_ ? hf_.data() + kHist : nullptr, aux_ ? hp_.data() + kHist : nullptr, m);
#ifdef VV_WALK_PROF
    g_wp[6] += __rdtsc() - tg_; g_wp[7] += (uint64_t)(n_ - m);
#endif
    pos_ = 0;
  }
  const uint32_t* peek() const { return base_ + pos_; }     // raw words: rand() = word >> 1
  // with the 512-bit forms, beside every word: its swap-in predicate as a byte (0xFF: (word >> 1) % 100 < negative_swap_percentage)
  // and (word >> 1) % max_buffer_size -- computed where the words are generated (the helper thread, when there is one) instead of
  // inside the walk's dependent chain
  bool has_aux() const { return aux_; }
  // With the stream thread on another core every line the walk reads comes out of that core's L2 (the helper wrote it there): ask for
  // the lines `pf_` words ahead of the cursor, one item's worth per item (VV_SAMPLER_PREFETCH words; 0 = off).
  void prefetch_ahead(int words) {
    if (!helper_on_ || pf_ <= 0) return;
    const int a = pos_ + pf_, b = std::min(a + words, avail_);
    for (int i = a & ~15; i < b; i += 16) {
      _mm_prefetch((const char*)(base_ + i), _MM_HINT_T0);
      if (aux_) { _mm_prefetch((const char*)(basep_ + i), _MM_HINT_T0); if ((i & 63) == 0) _mm_prefetch((const char*)(basef_ + i), _MM_HINT_T0); }
    }
  }
  const uint8_t* peek_flags() const { return basef_ + pos_; }
  const int32_t* peek_posmod() const { return basep_ + pos_; }
  void skip(int k) { pos_ += k; }
  int32_t next() { ensure(1); return (int32_t)(base_[pos_++] >> 1); }
  void discard(int64_t k) {
    while (k > 0) { const int step = (int)std::min<int64_t>(k, n_ / 2); ensure(step); pos_ += step; k -= step; }
  }

  // ---- generation off the consumer's thread.  The stream is a pure function of the seed, so the NEXT block can be
  // produced while the current one is consumed: two block buffers, each [n_ words of front slack][kHist][n_ values]; a
  // helper thread fills the idle one with the continuation of the other (its history = the other's last kHist values).
  // When the consumer runs out it copies its unconsumed tail (with the history in front of it) into the slack right in
  // front of the next block's values -- the run stays contiguous for peek() -- and hands the exhausted buffer back to the
  // helper.  Only the consumer's thread touches pos_ / base_ / avail_; the buffers change hands through fill_ / ready_.
  void start_helper() {
    if (helper_on_) return;
    for (int b = 0; b < 2; ++b) {
      blk_[b].assign((size_t)n_ + kHist + n_ + 8, 0u);
      if (aux_) { blkf_[b].assign((size_t)n_ + kHist + n_ + 64, 0); blkp_[b].assign((size_t)n_ + kHist + n_ + 16, 0); }
    }
    // the single buffer's state becomes block 0's: history + values at the same offsets, the cursor where it was
    memcpy(blk_[0].data() + n_, h_.data(), (size_t)(kHist + n_) * sizeof(uint32_t));
    cur_ = 0; base_ = blk_[0].data() + n_ + kHist; avail_ = n_;        // (pos_ unchanged)
    if (aux_) {
      memcpy(blkf_[0].data() + n_, hf_.data(), (size_t)(kHist + n_)); memcpy(blkp_[0].data() + n_, hp_.data(), (size_t)(kHist + n_) * sizeof(int32_t));
      basef_ = blkf_[0].data() + n_ + kHist; basep_ = blkp_[0].data() + n_ + kHist;
    }
    ready_[0].store(1); ready_[1].store(0);
    helper_stop_.store(0);
    fill_.store(1, std::memory_order_release);                        // fill block 1 as the continuation of block 0
    helper_on_ = true;
    helper_ = std::thread([this]() { helper_loop(); });
  }
  void stop_helper() {
    if (!helper_on_) return;
    helper_stop_.store(1);
    if (helper_.joinable()) helper_.join();
    helper_on_ = false;
    // back to the single buffer: the unconsumed run (with its history) to the front, the rest generated here
    const int m = avail_ - pos_;
    const int keep = std::min(m, n_);
    memcpy(h_.data(), base_ + pos_ - kHist, (size_t)(kHist + keep) * sizeof(uint32_t));
    if (aux_) { memcpy(hf_.data(), basef_ + pos_ - kHist, (size_t)(kHist + keep)); memcpy(hp_.data(), basep_ + pos_ - kHist, (size_t)(kHist + keep) * sizeof(int32_t)); }
    generate(h_.data() + kHist, aux_ ? hf_.data() + kHist : nullptr, aux_ ? hp_.data() + kHist : nullptr, keep);
    base_ = h_.data() + kHist; pos_ = 0; avail_ = n_;
    if (aux_) { basef_ = hf_.data() + kHist; basep_ = hp_.data() + kHist; }
    // (m > n_ cannot happen: a switch leaves at most n_ - 1 + n_ values and the next ensure() only runs it down)
    for (int b = 0; b < 2; ++b) { blk_[b].clear(); blk_[b].shrink_to_fit(); blkf_[b].clear(); blkf_[b].shrink_to_fit(); blkp_[b].clear(); blkp_[b].shrink_to_fit(); }
  }
  uint64_t wait_ticks() const { return wait_ticks_; }         // time-stamp-counter ticks the consumer waited for the helper's next block
  void pin_helper_like_caller(const cpu_set_t* set) { helper_set_ = set ? *set : cpu_set_t(); helper_pin_ = set != nullptr; }

 private:
  static constexpr int kHist = 48;               // words kept in front of the first unconsumed value (>= 46: the 512-bit generator's farthest operand; three whole vectors)
  // Eight words per step from x[k] = x[k-9] + x[k-31] + x[k-34] + x[k-37].  (Scalar on purpose: 256-bit loads of the operands would
  // straddle the stores of the last few steps -- no store forwarding, each step waits for them to drain -- and measured slower, ~2 against
  // 1.6 ticks per word.)
  void generate(uint32_t* __restrict h, uint8_t* f, int32_t* pm, int from) {       // h = the values' start, kHist words of history in front of it
    if (wide_) { generate_wide(h, f, pm, from); return; }
    const int n = n_;
    int i = from;
    for (; i + 8 <= n; i += 8) {                 // every operand of a block lies in front of the block
      uint32_t t[8];
      for (int j = 0; j < 8; ++j) t[j] = h[i + j - 9] + h[i + j - 31] + h[i + j - 34] + h[i + j - 37];
      for (int j = 0; j < 8; ++j) h[i + j] = t[j];
    }
    for (; i < n; ++i) h[i] = h[i - 31] + h[i - 3];
  }
  // 512-bit form: sixteen words per step need every operand at least sixteen positions back.  Substituting the recurrence into its
  // nearest term four more times gives x[k] = x[k-18] + x[k-31] + x[k-34] + x[k-37] + x[k-40] + x[k-43] + x[k-46]; the last 48 words stay
  // in three registers and the seven operand windows are cut out of them by VALIGND (read back from memory they would be loads that
  // straddle the stores of the last steps -- no store forwarding).  In the same pass, while the new words are in a register: their
  // swap-in predicate bytes and slot remainders (see peek_flags).
  template <bool AUX>
  VV_T512 int generate_wide_loop(uint32_t* __restrict h, uint8_t* __restrict f, int32_t* __restrict pm, int i) {
    const int n = n_;
    if (i + 16 > n) return i;
    const __m512i vsw = _mm512_set1_epi32((int)swap_);
    const __m512d vinv = _mm512_set1_pd(inv_mb_);
    const __m256i vd = _mm256_set1_epi32(mb_ > 0 ? mb_ : 1);
    __m512i v1 = _mm512_loadu_si512((const void*)(h + i - 16)), v2 = _mm512_loadu_si512((const void*)(h + i - 32)), v3 = _mm512_loadu_si512((const void*)(h + i - 48));
    for (; i + 16 <= n; i += 16) {
      const __m512i far = _mm512_add_epi32(_mm512_add_epi32(_mm512_add_epi32(_mm512_alignr_epi32(v2, v3, 14), _mm512_alignr_epi32(v2, v3, 11)),
                                                            _mm512_add_epi32(_mm512_alignr_epi32(v2, v3, 8), _mm512_alignr_epi32(v2, v3, 5))),
                                           _mm512_alignr_epi32(v2, v3, 2));
      const __m512i nv = _mm512_add_epi32(far, _mm512_add_epi32(_mm512_alignr_epi32(v1, v2, 14), _mm512_alignr_epi32(v1, v2, 1)));   // (the operands that depend on the last step join last)
      _mm512_storeu_si512((void*)(h + i), nv);
      if (AUX) {
        _mm_storeu_si128((__m128i*)(f + i), _mm_movm_epi8((__mmask16)swap_flags16v(nv, vsw)));
        _mm512_storeu_si512((void*)(pm + i), mod16(nv, vinv, vd));
      }
      v3 = v2; v2 = v1; v1 = nv;
    }
    return i;
  }
  void generate_wide(uint32_t* h, uint8_t* f, int32_t* pm, int from) {
    int i = f ? generate_wide_loop<true>(h, f, pm, from) : generate_wide_loop<false>(h, nullptr, nullptr, from);
    for (; i < n_; ++i) {
      h[i] = h[i - 31] + h[i - 3];
      if (f) { const uint32_t t = h[i] >> 1; f[i] = t % 100u < swap_ ? 0xFF : 0; pm[i] = (int32_t)(t % (uint32_t)mb_); }
    }
  }
  void helper_loop() {
    if (helper_pin_) (void)sched_setaffinity(0, sizeof(helper_set_), &helper_set_);
    unsigned spins = 0;
    while (!helper_stop_.load(std::memory_order_relaxed)) {
      const int b = fill_.load(std::memory_order_acquire);
      if (b < 0) {
        if (++spins < 2000) __builtin_ia32_pause(); else if (spins < 4000) sched_yield(); else std::this_thread::sleep_for(std::chrono::microseconds(20));
        continue;
      }
      spins = 0;
      uint32_t* dst = blk_[b].data() + n_ + kHist;                     // values of block b
      const uint32_t* src = blk_[1 - b].data() + n_ + kHist + n_ - kHist;   // the other block's last kHist values
      memcpy(dst - kHist, src, (size_t)kHist * sizeof(uint32_t));
      generate(dst, aux_ ? blkf_[b].data() + n_ + kHist : nullptr, aux_ ? blkp_[b].data() + n_ + kHist : nullptr, 0);
      fill_.store(-1, std::memory_order_relaxed);
      ready_[b].store(1, std::memory_order_release);
    }
  }
  void switch_block() {
    const int nb = 1 - cur_;
    unsigned spins = 0;
    if (!ready_[nb].load(std::memory_order_acquire)) {               // the helper has not finished the next block: the consumer outruns it
      const uint64_t t0 = __rdtsc();
      while (!ready_[nb].load(std::memory_order_acquire)) { if (++spins < 100000) __builtin_ia32_pause(); else sched_yield(); }
      wait_ticks_ += __rdtsc() - t0;
    }
    const int m = avail_ - pos_;                                       // unconsumed values of the current run
    uint32_t* nv = blk_[nb].data() + n_ + kHist;
    memcpy(nv - m - kHist, base_ + pos_ - kHist, (size_t)(kHist + m) * sizeof(uint32_t));   // (the copy's end rewrites the block's own history with the same words)
    if (aux_) {
      uint8_t* nf = blkf_[nb].data() + n_ + kHist; int32_t* np = blkp_[nb].data() + n_ + kHist;
      memcpy(nf - m, basef_ + pos_, (size_t)m); memcpy(np - m, basep_ + pos_, (size_t)m * sizeof(int32_t));
      basef_ = nf - m; basep_ = np - m;
    }
    base_ = nv - m; pos_ = 0; avail_ = m + n_;
    ready_[cur_].store(0, std::memory_order_relaxed);
    fill_.store(cur_, std::memory_order_release);                      // the exhausted block: continuation of block nb
    cur_ = nb;
  }
  std::vector<uint32_t> h_;
  const uint32_t* base_ = nullptr;
  int n_ = 0, pos_ = 0, avail_ = 0;
  uint32_t swap_ = 0;
  bool wide_ = false, aux_ = false;
  uint64_t wait_ticks_ = 0;
  int pf_ = 0;
  int mb_ = 0; double inv_mb_ = 0.0;
  std::vector<uint8_t> hf_; std::vector<int32_t> hp_;
  const uint8_t* basef_ = nullptr; const int32_t* basep_ = nullptr;
  // helper mode
  std::vector<uint32_t> blk_[2];
  std::vector<uint8_t> blkf_[2]; std::vector<int32_t> blkp_[2];
  std::thread helper_;
  bool helper_on_ = false; int cur_ = 0;
  std::atomic<int> fill_{-1}, helper_stop_{0};
  std::atomic<int> ready_[2];
  cpu_set_t helper_set_; bool helper_pin_ = false;
};

struct Slot { int32_t row = -1, last = -1; };

// a % d for 0 <= a < 2^32 and the divisors this sampler meets (1 .. max(max_buffer_size, longest video, 100)),
// by two multiplications with a precomputed reciprocal instead of a hardware division (Lemire, Kaser, Kurz:
// "Faster remainder by direct computation", 2019): M = floor((2^64 - 1) / d) + 1, a % d = floor(((M * a) mod 2^64) * d / 2^64).
class FastMod {
 public:
  void init(int dmax) {
    m_.resize((size_t)dmax + 1);
    for (int d = 1; d <= dmax; ++d) m_[d] = UINT64_C(0xFFFFFFFFFFFFFFFF) / (uint64_t)d + 1;
  }
  int32_t mod(int32_t a, int32_t d) const { return modm(m_[d], a, d); }
  uint64_t magic(int d) const { return m_[d]; }
  const uint64_t* table() const { return m_.data(); }
  static int32_t modm(uint64_t M, int32_t a, int32_t d) {
    const uint64_t low = M * (uint64_t)(uint32_t)a;
    return (int32_t)(((unsigned __int128)low * (uint64_t)d) >> 64);
  }
 private:
  std::vector<uint64_t> m_;
};

// Stage threads next to the thread that starts them: the stages hand cache lines to each other every few items, and
// the sampler's state was first touched by the caller, so a stage scheduled on another CCD or socket runs several
// times slower (measured on the 2 x 64-core EPYC 9575F hosts of the MI355X boxes: 0.17 ms per 1024 items with the
// stages beside the caller, 0.5-0.7 ms wherever the scheduler puts them).  The threads are confined to the caller's
// aligned group of 8 CPUs (one CCD = one L3 on those hosts) plus the SMT siblings the kernel reports for them -- a set,
// not one CPU each, so that a runtime helper thread landing on one of them cannot hold a stage up for a time slice.
// VV_SAMPLER_CPUS="a,b,c,..." overrides the set, VV_SAMPLER_PIN=0 disables.
bool stage_cpu_set(cpu_set_t* out, cpu_set_t* per_stage = nullptr, int* n_per_stage = nullptr, int* first_cpu = nullptr) {
  CPU_ZERO(out);
  if (n_per_stage) *n_per_stage = 0;
  const char* off = getenv("VV_SAMPLER_PIN");
  if (off && atoi(off) == 0) return false;
  if (const char* e = getenv("VV_SAMPLER_CPUS")) {
    int n = 0;
    for (const char* q = e; *q;) { const int c = atoi(q); if (c >= 0 && c < CPU_SETSIZE) { CPU_SET(c, out); ++n; } while (*q && *q != ',') ++q; if (*q) ++q; }
    return n > 0;
  }
  cpu_set_t allowed;
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
  const int cur = sched_getcpu();
  if (cur < 0) return false;
  const int g0 = cur & ~7;
  int n = 0;
  for (int c = g0; c < g0 + 8 && c < CPU_SETSIZE; ++c) {
    if (!CPU_ISSET(c, &allowed)) continue;
    CPU_SET(c, out); ++n;
    // a core of its own for each stage thread (the CPU and its SMT siblings), not the caller's: see the note at the function's end
    cpu_set_t* mine = (per_stage && n_per_stage && c != cur && *n_per_stage < 8) ? &per_stage[*n_per_stage] : nullptr;
    if (mine) { CPU_ZERO(mine); CPU_SET(c, mine); if (first_cpu) first_cpu[*n_per_stage] = c; }
    bool with_caller = false;
    char path[128];                                    // SMT siblings of c
    snprintf(path, sizeof(path), "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
    if (FILE* f = fopen(path, "r")) {
      char buf[128];
      if (fgets(buf, sizeof(buf), f))
        for (const char* q = buf; *q && *q != '\n';) {
          const int a = atoi(q); int b = a;
          while (*q && *q != ',' && *q != '-' && *q != '\n') ++q;
          if (*q == '-') { ++q; b = atoi(q); while (*q && *q != ',' && *q != '\n') ++q; }
          for (int k = a; k <= b && k < CPU_SETSIZE; ++k) {
            if (k >= 0 && CPU_ISSET(k, &allowed)) { CPU_SET(k, out); if (mine) CPU_SET(k, mine); }
            if (mine && first_cpu && k >= 0 && k < first_cpu[*n_per_stage]) first_cpu[*n_per_stage] = k;   // the core's name for claim_cores: its lowest CPU, whichever sibling the group holds
            if (k == cur) with_caller = true;
          }
          if (*q == ',') ++q;
        }
      fclose(f);
    }
    if (mine && !with_caller) ++*n_per_stage;
  }
  // Round 5: inside the common set the scheduler kept moving the stage threads (and now and then put two on the siblings of one
  // core): ten samplers of the 8192-item batch spread over 0.83 .. 1.34 ms; with the four threads on four cores of their own
  // 0.759 .. 0.768 (profiles/r05_sampler_place.txt; siblings sharing a core were slower in every pairing tried).  So when the
  // group has four cores beside the caller's THAT NO OTHER SAMPLER HOLDS, each stage thread gets one (per_stage; claimed by the caller
  // through claim_cores below); otherwise the common set as before.
  return n >= 4;                                       // too few neighbours: leave placement to the scheduler
}
// Two samplers whose callers sit in the same group of eight (two ranks of an unbound job, two samplers of one process) must not pin their
// stage threads to the same cores -- a spinning stage thread per core is the whole point.  A core is claimed by holding an exclusive
// flock on /dev/shm/vv_sampler_cpu_<n> for as long as the pipeline runs (the lock dies with the descriptor, i.e. with the process).
// The lock files: empty, one per core ever claimed on this host, never unlinked (unlinking a file another process is about to lock would
// give two holders of "the same" core); opened READ-ONLY -- flock works on any descriptor -- and made world-readable whatever the umask,
// so that samplers of different users on a shared host see each other's claims (as O_RDWR a second user's open failed, every
// claim with it, and the sampler fell back to the common set without a trace but vv_sampler_stat 9 == 0).
// -> number of cores claimed (their sets moved to the front of per_stage), descriptors in fds.
int claim_cores(cpu_set_t* per_stage, const int* first_cpu, int n, int want, int* fds) {
  int got = 0;
  for (int i = 0; i < n && got < want; ++i) {
    char path[64];
    snprintf(path, sizeof(path), "/dev/shm/vv_sampler_cpu_%d", first_cpu[i]);
    int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) {
      fd = open(path, O_CREAT | O_RDONLY | O_CLOEXEC, 0444);
      if (fd >= 0) (void)fchmod(fd, 0444);               // (the creator's umask may have taken the others' read bit)
    }
    if (fd < 0) continue;
    if (flock(fd, LOCK_EX | LOCK_NB) != 0) { close(fd); continue; }
    fds[got] = fd;
    per_stage[got] = per_stage[i];
    ++got;
  }
  if (got < want) { for (int i = 0; i < got; ++i) close(fds[i]); return 0; }
  return got;
}
void pin_self(const cpu_set_t* set) {
  if (set) (void)sched_setaffinity(0, sizeof(*set), set);
}

// Waiting for another stage / the consumer / the producer.  A sleep costs its quantum PLUS the timer slack and the wake-up
// (50 us asked = 100-130 us observed), a yield a trip through the scheduler: a stage that is a few items ahead of its
// producer, or a consumer whose batch is 20 us from complete, must not pay that -- the light frame stage oversleeping at
// a batch's end held every batch back by a sleep quantum (0.257 ms per 1024-item batch with three stage threads against
// 0.174 ms with two, profiles/r02_sampler_rates.txt; both figures also carried the CONSUMER's own oversleep).  So: spin on
// `pause` for ~100 us (these are dedicated threads on cores of their own, stage_cpu_set), then yield for a few ms, and
// only a wait longer than that (a full ring in front of a GPU-bound consumer, an idle pipeline) sleeps.
void backoff(unsigned& spins) {
  if (++spins < 6000) { __builtin_ia32_pause(); return; }
  if (spins < 12000) { sched_yield(); return; }
  std::this_thread::sleep_for(std::chrono::microseconds(50));
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------
// Batch ring: finished batches between the producer (the sampler's threads) and up to VV_RING_MAX_CONSUMERS
// consumers (the ranks of a node), in private memory or in a POSIX shared-memory object.
// ------------------------------------------------------------------------------------------------------------
enum { VV_RING_MAX_CONSUMERS = 64 };
struct RingHdr {
  uint64_t magic;
  int32_t depth, batch_size, cn, consumers, has_last, pad_;
  uint64_t batch_bytes;                          // bytes of one batch record
  // items finished by the negative-slot stage and by the frame stage: batch k is complete when both are past its end
  alignas(64) std::atomic<int64_t> done_negs;
  alignas(64) std::atomic<int64_t> done_frames;
  alignas(64) std::atomic<int32_t> closed;       // producer gone
  struct alignas(64) Rel { std::atomic<int64_t> v; } released[VV_RING_MAX_CONSUMERS];
};
static constexpr uint64_t kRingMagic = 0x5656524e47303031ull;   // "VVRNG001"

struct vv_batch_ring {
  RingHdr* hdr = nullptr;
  unsigned char* data = nullptr;
  size_t map_bytes = 0;
  bool shm = false, owner = false;
  std::string name;
  int64_t next_k[VV_RING_MAX_CONSUMERS];
  int32_t* idx_of(int64_t k) const { return (int32_t*)(data + (size_t)(k % hdr->depth) * hdr->batch_bytes); }
  int32_t* label_of(int64_t k) const { return idx_of(k) + (size_t)hdr->batch_size * hdr->cn; }
  int32_t* last_of(int64_t k) const { return label_of(k) + hdr->batch_size; }
  bool ready(int64_t k) const {
    const int64_t need = (k + 1) * (int64_t)hdr->batch_size;
    return hdr->done_negs.load(std::memory_order_acquire) >= need && hdr->done_frames.load(std::memory_order_acquire) >= need;
  }
  int64_t min_released() const {
    int64_t m = INT64_MAX;
    for (int i = 0; i < hdr->consumers; ++i) m = std::min(m, hdr->released[i].v.load(std::memory_order_acquire));
    return m;
  }
};

static size_t ring_hdr_bytes() { return (sizeof(RingHdr) + 4095) / 4096 * 4096; }

static vv_batch_ring* ring_create(const char* shm_name, int depth, int B, int CN, int consumers, bool has_last) {
  vv_batch_ring* r = new (std::nothrow) vv_batch_ring();
  if (!r) return nullptr;
  const size_t rec = ((size_t)B * CN * (has_last ? 2 : 1) + B) * sizeof(int32_t);
  const size_t rec_al = (rec + 63) / 64 * 64;
  r->map_bytes = ring_hdr_bytes() + rec_al * depth;
  void* mem = nullptr;
  if (shm_name && *shm_name) {
    r->name = shm_name[0] == '/' ? shm_name : std::string("/") + shm_name;
    shm_unlink(r->name.c_str());
    const int fd = shm_open(r->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) { delete r; return nullptr; }
    if (ftruncate(fd, (off_t)r->map_bytes) != 0) { close(fd); shm_unlink(r->name.c_str()); delete r; return nullptr; }
    mem = mmap(nullptr, r->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) { shm_unlink(r->name.c_str()); delete r; return nullptr; }
    r->shm = true;
  } else {
    mem = mmap(nullptr, r->map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (mem == MAP_FAILED) { delete r; return nullptr; }
  }
  r->owner = true;
  r->hdr = new (mem) RingHdr();
  r->data = (unsigned char*)mem + ring_hdr_bytes();
  r->hdr->depth = depth; r->hdr->batch_size = B; r->hdr->cn = CN; r->hdr->consumers = consumers;
  r->hdr->has_last = has_last ? 1 : 0; r->hdr->batch_bytes = rec_al;
  r->hdr->done_negs.store(0); r->hdr->done_frames.store(0); r->hdr->closed.store(0);
  for (int i = 0; i < VV_RING_MAX_CONSUMERS; ++i) { r->hdr->released[i].v.store(0); r->next_k[i] = 0; }
  std::atomic_thread_fence(std::memory_order_release);
  r->hdr->magic = kRingMagic;
  return r;
}

static void ring_free(vv_batch_ring* r) {
  if (!r) return;
  if (r->hdr) {
    if (r->owner) r->hdr->closed.store(1, std::memory_order_release);
    munmap((void*)r->hdr, r->map_bytes);
  }
  if (r->owner && r->shm) shm_unlink(r->name.c_str());
  delete r;
}

// ------------------------------------------------------------------------------------------------------------
struct vv_sampler {
  vv_sampler_param p;
  std::vector<int32_t> video_id, n_shots, shot_ids;
  std::vector<int64_t> row_base, shot_off;
  bool has_ids = false;
  int max_n = 1;
  LibcRand rng;
  FastMod fm;
  int32_t rmod(int32_t d) { return fm.mod(rng.next(), d); }     // rand() % d
  int32_t cursor = 0;
  std::vector<int32_t> buffer_ids;            // persistent permutation (…data_layer.cpp:81-83)
  std::vector<int32_t> buf_row;               // slot -> table row
  std::vector<int32_t> buf_row_negs;          // the negs stage's own view of the buffer (pipelined mode)
  std::vector<uint64_t> buf_key;              // slot -> (video_id, shot_id)
  std::unordered_set<uint64_t> keys;          // negative_keys_set_ (general case)
  // fast path: when every (video_id, shot_id) key names exactly one table row, membership in the key
  // set is a bitmap over table rows (padded, so that vector reads past a video's rows stay inside)
  bool dense_keys = false;
  std::vector<uint8_t> row_in_buf;
  int64_t row_min = 0;
  bool contains(int v, int j) const {
    return dense_keys ? row_in_buf[(size_t)(row_base[v] - row_min + j)] != 0 : keys.count(key(video_id[v], shot_id(v, j))) != 0;
  }
  void insert_key(int v, int j) {
    if (dense_keys) row_in_buf[(size_t)(row_base[v] - row_min + j)] = 1; else keys.insert(key(video_id[v], shot_id(v, j)));
  }
  std::vector<Slot> slots;                    // persistent prefetch_data_ contents [B][C+Nn] (general path)
  std::vector<int32_t> perm;

  static uint64_t key(int32_t vid, int32_t shot) { return ((uint64_t)(uint32_t)vid << 32) | (uint32_t)shot; }
  int32_t shot_id(int v, int j) const { return has_ids ? shot_ids[shot_off[v] + j] : j; }
  // …data_layer.cpp:412-415: the frame distance, or max_shot_distance (a float bound held in an int) from there on
  int32_t pair_label(int f0, int f1) const {
    const int d = f0 > f1 ? f0 - f1 : f1 - f0;
    return (float)d >= p.max_shot_distance ? (int32_t)p.max_shot_distance : d;
  }

  // include/caffe/util/rng.hpp:43-54
  void random_unique(std::vector<int32_t>& a, int n) {
    int left = (int)a.size();
    for (int first = 0; first < n; ++first, --left) std::swap(a[first], a[first + rmod(left)]);
  }

  // ---- staged fast path (dense keys, no same-video negatives)
  bool fast = false;
  int CA = 0;                                 // stream values the frames stage looks at (C, 2 or 0 by context type)
  int rec_words = 0;                          // words of one item record: {v, n_events, ev_off_lo, ev_off_hi, vals[CA + Nn]}
  struct Event { int32_t pos, row; };
  int next_general(int32_t* idx, int32_t* last_src, int32_t* label);
  void select_item(uint32_t* rec);
  template <bool LOG> void swap_item(uint32_t* rec, int32_t* brow, Event* ev_ring, uint64_t ev_mask, uint64_t* ev_head);
  template <bool LOG> void swap_item_512(uint32_t* rec, int32_t* brow, Event* ev_ring, uint64_t ev_mask, uint64_t* ev_head);
  template <bool LOG> void swap(uint32_t* rec, int32_t* brow, Event* ev_ring, uint64_t evm, uint64_t* ev_head) {
    if (wide && rng.has_aux()) swap_item_512<LOG>(rec, brow, ev_ring, evm, ev_head); else swap_item<LOG>(rec, brow, ev_ring, evm, ev_head);
  }
  bool wide = false;                          // AVX-512 (F, BW, DQ, VL, VBMI2) forms of the walk and of the stream generation; VV_SAMPLER_AVX512=0 disables
  double inv_mb = 0.0;                        // 1.0 / max_buffer_size
  int cur_v = 0, cur_n = 0, cur_a_total = 0;  // the item between select_item and swap_item
  int64_t stat_restarts = 0;
  int sample_batch(int32_t* idx, int32_t* last_src, int32_t* label);
  void negs_item(const uint32_t* rec, int32_t* out, const int32_t* brow);
  void negs(const uint32_t* rec, int32_t* out, const int32_t* brow) { negs_item(rec, out, brow); }   // (a 512-bit form -- the Nn remainders in double precision first -- measured SLOWER than this loop: 114 against 86 ns per 50 slots)
  void frames_item(const uint32_t* rec, int32_t* out, int32_t* label);
  std::vector<uint32_t> rec1;                 // the serial path's one item record

  // ---- prefetch pipeline
  vv_batch_ring* ring = nullptr;
  std::vector<std::thread> threads;
  int core_fds[4] = {-1, -1, -1, -1}, n_core_fds = 0;     // the stage threads' cores, held while the pipeline runs (claim_cores)
  std::atomic<int> stop{0};
  int n_stage_threads = 0;
  std::vector<uint32_t> recs; int64_t ring_items = 0;
  std::vector<Event> events; uint64_t ev_mask = 0;
  alignas(64) std::atomic<int64_t> walked{0};
  alignas(64) std::atomic<uint64_t> ev_tail{0};
  // time-stamp-counter ticks each stage spent waiting (for its producer, its consumers, ring space); vv_sampler_stat 3..6
  alignas(64) std::atomic<uint64_t> wait_walk{0};
  alignas(64) std::atomic<uint64_t> wait_negs{0};
  alignas(64) std::atomic<uint64_t> wait_frames{0};
  uint64_t tsc_start = 0;
  void run_batches();
  void run_walk();
  void run_negs(bool also_frames);
  void run_frames();
};

// ---- the three chains of one item -----------------------------------------------------------------------------
// walk = select_item + swap_item.  select: pick the item's record (…data_layer.cpp:796-848: a record with fewer than
// 2 or fewer than C shots adds nothing and consumes no draw) and hand the frames / negs stages the raw stream words they
// own.  swap: the swap-in loop (:888-906 with AddToBuffer :24-37), branch-free: a shot that is not swapped in writes
// to a dummy slot / dummy row instead of branching on a coin flip.
#ifdef VV_WALK_PROF
uint64_t g_wp[8];
#endif
#ifdef VV_WALK_LAB
int g_lab = 0;
#endif
#ifdef VV_SAMPLER_LAB
static int g_plab = 0;       // (lab build only: parts of the pipeline's hand-over switched off, results wrong by design)
#endif
void vv_sampler::select_item(uint32_t* rec) {
  const int C = p.context_size, Nn = p.num_negative_samples, V = (int)video_id.size();
  int v, n;
  for (;;) {
    v = cursor; n = n_shots[v];
    cursor = cursor + 1 == V ? 0 : cursor + 1;
    if (n >= 2 && n >= C) break;
  }
  const bool shuffled = (p.context_type == VV_CONTEXT_WINDOW || p.context_type == VV_CONTEXT_PAST) && Nn > 0 && n > C;
  const int a_total = CA + (shuffled ? n - C - 1 : 0);     // :432/:517 random_unique, :482/:566 random_shuffle
#ifdef VV_WALK_PROF
  { const uint64_t t_ = __rdtsc(); rng.ensure(a_total + Nn + 2 * n + 160); g_wp[4] += __rdtsc() - t_; }
#else
  rng.ensure(a_total + Nn + 2 * n + 160);     // the swap-in reads whole groups of 8 / 16 words, the 512-bit form 128 predicate bytes from its position
#endif
  rec[0] = (uint32_t)v;
  rng.prefetch_ahead(a_total + Nn + 2 * n);
  const uint32_t* hv = rng.peek();
#ifdef VV_SAMPLER_LAB
  if (!(g_plab & 8))
#endif
  {
  memcpy(rec + 4, hv, (size_t)CA * 4);
  memcpy(rec + 4 + CA, hv + a_total, (size_t)Nn * 4);
  }
  cur_v = v; cur_n = n; cur_a_total = a_total;
}

// (rand() % 100 < swap) for 8 consecutive stream words -> 8 mask bits.  rand() = word >> 1 < 2^31;
// t / 100 = (t * 0x51EB851F) >> 37 exactly for every 32-bit t.
static inline uint32_t swap_flags8(const uint32_t* w, __m256i vsw) {
  const __m256i t = _mm256_srli_epi32(_mm256_loadu_si256((const __m256i*)w), 1);
  const __m256i magic = _mm256_set1_epi32(0x51EB851F);
  const __m256i pe = _mm256_srli_epi64(_mm256_mul_epu32(t, magic), 37);                           // lanes 0,2,4,6
  const __m256i po = _mm256_srli_epi64(_mm256_mul_epu32(_mm256_srli_epi64(t, 32), magic), 37);   // lanes 1,3,5,7
  const __m256i q = _mm256_blend_epi32(pe, _mm256_slli_epi64(po, 32), 0xAA);
  const __m256i r = _mm256_sub_epi32(t, _mm256_mullo_epi32(q, _mm256_set1_epi32(100)));
  return (uint32_t)_mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(vsw, r)));
}

typedef unsigned __int128 u128;
static inline int select128(u128 x, int k) {        // position of the k-th (1-based) set bit; k <= popcount(x)
  const uint64_t lo = (uint64_t)x, hi = (uint64_t)(x >> 64);
  const int c0 = __builtin_popcountll(lo);
  // both halves, then a select: which half holds the k-th bit depends on the video's length -- as a branch it is mispredicted often
  const int in_lo = (int)_tzcnt_u64(_pdep_u64(1ull << ((k - 1) & 63), lo));
  const int in_hi = 64 + (int)_tzcnt_u64(_pdep_u64(1ull << ((k - 1 - c0) & 63), hi));
  return k <= c0 ? in_lo : in_hi;
}

// The swap-in of one video (:888-906 with AddToBuffer :24-37).  The reference walks the video's shots in order; a shot
// whose key is not in the buffer draws rand() % 100 (a "test") and, below negative_swap_percentage, a second value
// rand() % max_buffer_size for the slot it replaces.  Which stream positions are tests follows from the predicate bits
// F alone: a position is a test unless its predecessor is a taken test; inside a run of ones the tests alternate from
// the run's start.  With E / O the runs that start at an even / odd position (carry trick: F & ~(F + even starts)),
//   TEST = ~(F << 1) | ((E << 1) & EVEN) | ((O << 1) & ODD).
// The k-th shot that is not in the buffer owns the k-th test; the taken ones are pdep(pext(F, TEST), notin).  Only the
// taken shots are then visited.  A swap-in that evicts a LATER shot of the same video changes that shot's membership:
// the walk restarts behind the evicting shot (rare; counted in stat_restarts).
#ifdef VV_WALK_PROF
#define WP(i, expr) { const uint64_t t_ = __rdtsc(); expr; g_wp[i] += __rdtsc() - t_; }
#define WPT(i) { const uint64_t t_ = __rdtsc(); g_wp[i] += t_ - wp_t; wp_t = t_; }
#else
#define WPT(i)
#endif
template <bool LOG>
__attribute__((noinline)) void vv_sampler::swap_item(uint32_t* rec, int32_t* brow, Event* ev_ring, uint64_t evm, uint64_t* ev_head) {
  const int Nn = p.num_negative_samples, v = cur_v, n = cur_n, a_total = cur_a_total;
  const uint32_t* hv = rng.peek() + a_total + Nn;
  int q = 0;
  uint32_t nev = 0;
  const uint64_t ev0 = LOG ? *ev_head : 0;
#ifdef VV_WALK_PROF
  uint64_t wp_t = __rdtsc();
#endif
  if (Nn > 0 && p.negative_swap_percentage > 0) {
    uint8_t* inb = row_in_buf.data() - row_min;              // indexed by table row
    const int mb = p.max_buffer_size;
    const uint64_t M = fm.magic(mb);
    const int64_t base = row_base[v];
    const __m256i vsw = _mm256_set1_epi32(p.negative_swap_percentage);
    int j0 = 0;
    while (j0 < n) {
      const int cnt = std::min(n - j0, 64);
      // shots j0 .. j0+cnt-1 that are not in the buffer (the bitmap is padded: reading 64 bytes is always in bounds)
      const uint8_t* ib = inb + base + j0;
      const __m256i z = _mm256_setzero_si256();
      uint64_t notin = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)ib), z));
      if (cnt > 32) notin |= (uint64_t)(uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_loadu_si256((const __m256i*)(ib + 32)), z)) << 32;
      if (cnt < 64) notin &= (1ull << cnt) - 1;
      const int m = __builtin_popcountll(notin);
      if (m == 0) { j0 += cnt; continue; }
      // predicate bits of stream positions q .. q + 2m - 1
      const uint32_t* w = hv + q;
      uint64_t f0 = 0, f1 = 0;
      const int groups = (2 * m + 7) >> 3;
      for (int g = 0; g < groups && g < 8; ++g) f0 |= (uint64_t)swap_flags8(w + 8 * g, vsw) << (8 * g);
      for (int g = 8; g < groups; ++g) f1 |= (uint64_t)swap_flags8(w + 8 * g, vsw) << (8 * (g - 8));
      const u128 F = ((u128)f1 << 64) | f0;
      const u128 EVEN = ((u128)0x5555555555555555ull << 64) | 0x5555555555555555ull;
      const u128 S = F & ~(F << 1);
      const u128 E = F & ~(F + (S & EVEN));
      const u128 O = F & ~E;
      const u128 TEST = ~(F << 1) | ((E << 1) & EVEN) | ((O << 1) & ~EVEN);
      const int pm = select128(TEST, m);                     // position of the last test of this chunk
      const u128 upto = pm == 127 ? ~(u128)0 : (((u128)1 << (pm + 1)) - 1);
      u128 TK = TEST & F & upto;                             // taken tests, in stream order
      const uint64_t t_lo = (uint64_t)TEST, t_hi = (uint64_t)(TEST >> 64);
      const int c0 = __builtin_popcountll(t_lo);
      uint64_t comp = _pext_u64((uint64_t)F, t_lo);          // predicate of the k-th test at bit k
      if (c0 < 64) comp |= _pext_u64((uint64_t)(F >> 64), t_hi) << c0;
      uint64_t TS = _pdep_u64(comp, notin);                  // taken shots, in shot order (same count as TK)
      int q_end = q + pm + 1 + (int)((F >> pm) & 1);
      bool restarted = false;
      WPT(1)
      while (TS) {
        const int j = j0 + (int)_tzcnt_u64(TS); TS &= TS - 1;
        const uint64_t k_lo = (uint64_t)TK;
        const int tp = k_lo ? (int)_tzcnt_u64(k_lo) : 64 + (int)_tzcnt_u64((uint64_t)(TK >> 64));
        TK &= TK - 1;
        const int32_t r = (int32_t)(base + j);
        const int32_t pos = FastMod::modm(M, (int32_t)(w[tp + 1] >> 1), mb);   // :29 rand() % max_buffer_size
        const int32_t old = brow[pos];
        inb[old] = 0; inb[r] = 1; brow[pos] = r;
        if (LOG) { ev_ring[(ev0 + nev) & evm] = Event{pos, r}; ++nev; }
        // (ONE unsigned compare for r < old < end of chunk: written as two, the first -- is the evicted row behind this one in the
        // table? -- is a coin flip the branch predictor loses half the time: it was 10 of the loop's 12 cycles per swap-in)
        if ((uint32_t)(old - r - 1) < (uint32_t)((int32_t)(base + j0 + cnt) - r - 1)) {
          // a later shot of this video left the buffer: its membership bit above is stale
          j0 = j + 1; q = q + tp + 2; restarted = true; ++stat_restarts;
          break;
        }
      }
      if (!restarted) { q = q_end; j0 += cnt; }
      WPT(2)
    }
  }
  rng.skip(a_total + Nn + q);
  rec[1] = nev; rec[2] = (uint32_t)ev0; rec[3] = (uint32_t)(ev0 >> 32);
  if (LOG) *ev_head = ev0 + nev;
}

// ---- the same swap-in with 512-bit vectors (Zen 4/5, Sapphire Rapids ...): sixteen predicate bits per compare-into-mask, the
// membership bits of 64 shots from one load, and -- the larger part -- the taken tests' buffer positions computed for the whole
// chunk at once instead of inside the loop that applies them: the stream words behind the taken tests are packed by VPCOMPRESSD
// (mask = the taken tests' bits moved up by one), the taken shots' numbers by VPCOMPRESSB, and rand() % max_buffer_size is taken in
// double precision: for 0 <= a < 2^31 the rounded product a * (1/d) is within a/d * 2^-52 < 1/d of a/d, so its integer part is the
// quotient except when a is a multiple of d and the product fell just below it (remainder d instead of 0: one compare).  The loop that
// is left reads (position, shot) pairs and moves rows -- the part that has to stay sequential (two tests may name the same slot).
template <bool LOG>
VV_T512 __attribute__((noinline)) void vv_sampler::swap_item_512(uint32_t* rec, int32_t* brow, Event* ev_ring, uint64_t evm, uint64_t* ev_head) {
  const int Nn = p.num_negative_samples, v = cur_v, n = cur_n, a_total = cur_a_total;
  const uint8_t* hf = rng.peek_flags() + a_total + Nn;       // the stream's predicate bytes and slot remainders from this item's first swap-in word on
  const int32_t* hp = rng.peek_posmod() + a_total + Nn;
  int q = 0;
  uint32_t nev = 0;
  const uint64_t ev0 = LOG ? *ev_head : 0;
#ifdef VV_WALK_PROF
  uint64_t wp_t = __rdtsc();
#endif
  if (Nn > 0 && p.negative_swap_percentage > 0) {
    uint8_t* inb = row_in_buf.data() - row_min;              // indexed by table row
    const int64_t base = row_base[v];
    const __m512i iota = _mm512_set_epi8(63, 62, 61, 60, 59, 58, 57, 56, 55, 54, 53, 52, 51, 50, 49, 48, 47, 46, 45, 44, 43, 42, 41, 40, 39, 38, 37, 36, 35, 34, 33, 32,
                                         31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0);
    alignas(64) int32_t posv[64 + 16];
    alignas(64) uint8_t jb[64];
    int j0 = 0;
    while (j0 < n) {
      const int cnt = std::min(n - j0, 64);
      // shots j0 .. j0+cnt-1 that are not in the buffer (the bitmap is padded: reading 64 bytes is always in bounds)
      uint64_t notin = _mm512_cmpeq_epi8_mask(_mm512_loadu_si512((const void*)(inb + base + j0)), _mm512_setzero_si512());
      if (cnt < 64) notin &= (1ull << cnt) - 1;
      const int m = __builtin_popcountll(notin);
      if (m == 0) { j0 += cnt; continue; }
      // predicate bits of stream positions q .. q + 127 (the stream carries them as bytes: LibcRand::generate_wide)
      const __m512i fb0 = _mm512_loadu_si512((const void*)(hf + q)), fb1 = _mm512_loadu_si512((const void*)(hf + q + 64));
      const uint64_t f0 = _mm512_test_epi8_mask(fb0, fb0), f1 = _mm512_test_epi8_mask(fb1, fb1);
      const int32_t* w = hp + q;
      const u128 F = ((u128)f1 << 64) | f0;
      const u128 EVEN = ((u128)0x5555555555555555ull << 64) | 0x5555555555555555ull;
      const u128 S = F & ~(F << 1);
      const u128 E = F & ~(F + (S & EVEN));
      const u128 O = F & ~E;
      const u128 TEST = ~(F << 1) | ((E << 1) & EVEN) | ((O << 1) & ~EVEN);
      const int pm = select128(TEST, m);                     // position of the last test of this chunk
      const u128 upto = pm == 127 ? ~(u128)0 : (((u128)1 << (pm + 1)) - 1);
      const u128 TK = TEST & F & upto;                       // taken tests, in stream order
      const uint64_t t_lo = (uint64_t)TEST, t_hi = (uint64_t)(TEST >> 64);
      const int c0 = __builtin_popcountll(t_lo);
      uint64_t comp = _pext_u64((uint64_t)F, t_lo);          // predicate of the k-th test at bit k
      if (c0 < 64) comp |= _pext_u64((uint64_t)(F >> 64), t_hi) << c0;
      const uint64_t TS = _pdep_u64(comp, notin);            // taken shots, in shot order (same count as TK)
      const int q_end = q + pm + 1 + (int)((F >> pm) & 1);
      // the chunk's (slot, shot) pairs: the slot remainder of the word behind each taken test; the shots' numbers
      const uint64_t k_lo = (uint64_t)TK, k_hi = (uint64_t)(TK >> 64);
      int nt = 0;
      WPT(1)
      {
        // (the m tests of a chunk end at position 2(m-1) <= 126 at the latest: the last position word is word 127.)  Four groups of
        // sixteen without looking at their masks -- an empty group packs nothing -- then the upper four if any taken test lies there
        uint64_t pw = k_lo << 1;                             // bit i: word i is a position word
        for (int g = 0; g < 4; ++g, pw >>= 16) {
          const __mmask16 mk = (__mmask16)pw;
          _mm512_storeu_si512((void*)(posv + nt), _mm512_maskz_compress_epi32(mk, _mm512_loadu_si512((const void*)(w + 16 * g))));
          nt += __builtin_popcount(mk);
        }
        if ((k_lo >> 63) | k_hi) {
          pw = (k_hi << 1) | (k_lo >> 63);
          for (int g = 4; g < 8; ++g, pw >>= 16) {
            const __mmask16 mk = (__mmask16)pw;
            _mm512_storeu_si512((void*)(posv + nt), _mm512_maskz_compress_epi32(mk, _mm512_loadu_si512((const void*)(w + 16 * g))));
            nt += __builtin_popcount(mk);
          }
        }
      }
      _mm512_store_si512((void*)jb, _mm512_maskz_compress_epi8((__mmask64)TS, iota));
      bool restarted = false;
      WPT(5)
      const int32_t hi_row = (int32_t)(base + j0 + cnt);
#ifdef VV_WALK_LAB
      if (g_lab & 1) nt = 0;                                 // (lab: no buffer updates at all)
      static int32_t posv2[80]; static uint8_t jb2[64];
      if (g_lab & 16) {                                      // (lab: slots / shots of the PREVIOUS chunk -- no store-to-load forwarding from the vector stores)
        for (int k = 0; k < nt; ++k) {
          const int32_t pos = posv2[k];
          const int32_t r = (int32_t)(base + j0 + (jb2[k] % cnt));
          const int32_t old = brow[pos];
          inb[old] = 0; inb[r] = 1; brow[pos] = r;
        }
        memcpy(posv2, posv, sizeof(posv2)); memcpy(jb2, jb, 64);
        nt = 0;
      }
#endif
      for (int k = 0; k < nt; ++k) {
        const int32_t pos = posv[k];
        const int32_t r = (int32_t)(base + j0 + jb[k]);
#ifdef VV_WALK_LAB
        if (g_lab & 2) { const int32_t old = brow[pos]; brow[pos] = r; if ((uint32_t)(old - r - 1) < (uint32_t)(hi_row - r - 1)) ++stat_restarts; continue; }   // no membership stores
        if (g_lab & 32) { const int32_t old = brow[pos]; if ((uint32_t)(old - r - 1) < (uint32_t)(hi_row - r - 1)) ++stat_restarts; continue; }   // loads only
        if (g_lab & 64) { brow[pos] = r; continue; }                                                // slot stores only
        if (g_lab & 128) { const int32_t old = brow[pos]; inb[old] = 0; if ((uint32_t)(old - r - 1) < (uint32_t)(hi_row - r - 1)) ++stat_restarts; continue; }   // load + evicted row's byte
        if (g_lab & 256) { inb[(uint32_t)(pos * 16) % 80000u] = 0; continue; }                       // a random byte store alone
        if (g_lab & 4) { inb[r] = 1; continue; }                                                   // no slot exchange
        if (g_lab & 8) { const int32_t old = brow[pos]; inb[old & 0xfff] = 0; inb[r] = 1; brow[pos] = r; continue; }   // evicted row's byte in a 4-KiB window
#endif
        const int32_t old = brow[pos];
        inb[old] = 0; inb[r] = 1; brow[pos] = r;
        if (LOG) { ev_ring[(ev0 + nev) & evm] = Event{pos, r}; ++nev; }
        if ((uint32_t)(old - r - 1) < (uint32_t)(hi_row - r - 1)) {      // r < old < hi_row in one compare (see swap_item)
          // a later shot of this video left the buffer: its membership bit above is stale
          const int tp = select128(TK, k + 1);
          j0 = j0 + jb[k] + 1; q = q + tp + 2; restarted = true; ++stat_restarts;
          break;
        }
      }
      if (!restarted) { q = q_end; j0 += cnt; }
      WPT(2)
    }
  }
  rng.skip(a_total + Nn + q);
  rec[1] = nev; rec[2] = (uint32_t)ev0; rec[3] = (uint32_t)(ev0 >> 32);
  if (LOG) *ev_head = ev0 + nev;
}

// negs: :855 random_unique over the persistent slot permutation, fused with :856-875 (the drawn slots' rows)
void vv_sampler::negs_item(const uint32_t* rec, int32_t* out, const int32_t* brow) {
  const int Nn = p.num_negative_samples;
  if (Nn <= 0) return;
  int32_t* ids = buffer_ids.data();
  const uint32_t* x = rec + 4 + CA;
  const uint64_t* M = fm.table();
  int left = p.max_buffer_size;
  for (int first = 0; first < Nn; ++first, --left) {
    const int r = first + FastMod::modm(M[left], (int32_t)(x[first] >> 1), left);
    const int32_t t = ids[r]; ids[r] = ids[first]; ids[first] = t;
    out[first] = brow[t];
  }
}

// frames: AddSamplesToTop without same-video negatives (:425-453, :510-538, :599-640, :677-718)
void vv_sampler::frames_item(const uint32_t* rec, int32_t* out, int32_t* label) {
  const int C = p.context_size, v = (int)rec[0], n = n_shots[v];
  const int64_t base = row_base[v];
  const uint32_t* x = rec + 4;
  if (label) *label = video_id[v];                                                   // :879
  if (p.context_type == VV_CONTEXT_PAIRWISE) {
    // :397 random_unique(perm, 2) on the identity permutation: perm[0] = j0, then perm[1] <-> perm[j1]
    const int j0 = fm.mod((int32_t)(x[0] >> 1), n);
    const int j1 = 1 + fm.mod((int32_t)(x[1] >> 1), n - 1);
    const int f0 = j0, f1 = j1 == j0 ? 0 : j1;
    out[0] = (int32_t)(base + f0); out[1] = (int32_t)(base + f1);                    // :400-405, draw order, not sorted
    if (label && p.output_shot_distance) *label = pair_label(f0, f1);               // :407-415
  } else if (p.context_type == VV_CONTEXT_WINDOW || p.context_type == VV_CONTEXT_PAST) {
    // random_unique(perm, C) on the identity permutation of 0..n-1, kept sparse.  Step `first` exchanges perm[first] and perm[j],
    // j = first + rand() % left >= first: perm[first] is final from then on and is never read again, so only the entries that were
    // some earlier step's j matter -- (J[k], W[k]) = "perm[J[k]] holds W[k]", the latest k wins.  The look-ups run over all earlier
    // steps without branching on the comparison (at C = 5 ten compare-selects per item instead of data-dependent searches), and the
    // sort (:437, :522) is an odd-even transposition network of min / max: the frame stage's time was mostly mispredicted branches.
    int32_t Jb[32], Wb[32], fr[32];
    std::vector<int32_t> big;
    int32_t *J = Jb, *W = Wb, *pf = fr;
    if (C > 32) { big.resize(3 * (size_t)C); J = big.data(); W = J + C; pf = W + C; }
    for (int first = 0, left = n; first < C; ++first, --left) {
      const int j = first + fm.mod((int32_t)(x[first] >> 1), left);
      int32_t vf = first, vj = j;
      for (int k = 0; k < first; ++k) { vf = J[k] == first ? W[k] : vf; vj = J[k] == j ? W[k] : vj; }
      vj = j == first ? vf : vj;
      pf[first] = vj; J[first] = j; W[first] = vf;
    }
    for (int round = 0; round < C; ++round)
      for (int i = round & 1; i + 1 < C; i += 2) { const int32_t a = pf[i], b = pf[i + 1]; pf[i] = a < b ? a : b; pf[i + 1] = a < b ? b : a; }
    if (p.context_type == VV_CONTEXT_WINDOW) {
      const int half = C / 2;
      for (int i = 0, ctx = 0; i < C; ++i) {                                         // :439-453: the middle one is the target
        if (i == half) out[0] = (int32_t)(base + pf[i]); else out[++ctx] = (int32_t)(base + pf[i]);
      }
    } else {
      for (int i = 0; i < C; ++i) out[i == C - 1 ? 0 : i + 1] = (int32_t)(base + pf[i]);   // :524-538: the last one
    }
  } else {
    const int msl = (n - C) / (C - 1);                                               // :609, :687
    int sl, begin;
    if (p.context_type == VV_CONTEXT_PAST_CONTINUOUS) {
      sl = fm.mod((int32_t)(x[0] >> 1), msl + 1);                                    // :610
      begin = fm.mod((int32_t)(x[1] >> 1), n - (C - 1) * sl - C + 1);                // :612-613
    } else {
      sl = msl >= 1 ? msl - 1 : 0;                                                   // :688
      begin = n - (C - 1) * sl - C;                                                  // :690-691
    }
    for (int i = 0; i < C; ++i) out[i == C - 1 ? 0 : i + 1] = (int32_t)(base + begin + i * (sl + 1));
  }
}

// ---- general path: same-video negatives (quirk Q1 needs the persistent slot contents) or keys that are not rows ----
int vv_sampler::next_general(int32_t* idx, int32_t* last_src, int32_t* label) {
  vv_sampler* s = this;
  const int C = p.context_size, Nn = p.num_negative_samples, CN = C + Nn, half = C / 2;
  const int V = (int)s->video_id.size();
  for (int item = 0; item < p.batch_size;) {
    const int v = s->cursor;
    const int n = s->n_shots[v];
    const int64_t base = s->row_base[v];
    Slot* sl = &s->slots[(size_t)item * CN];
    int added = 0;
    const int max_same = std::min(p.max_same_video_negs, Nn);                        // never past the item's Nn slots
    const bool ok = n >= 2 && n >= C;                                              // :387,:427,:512,:601,:679
    int32_t lab = s->video_id[v];
    if (ok && p.context_type == VV_CONTEXT_PAIRWISE) {                             // :396-422
      std::vector<int32_t>& perm = s->perm;
      perm.resize(n);
      for (int i = 0; i < n; ++i) perm[i] = i;
      s->random_unique(perm, 2);                                                   // :397
      sl[0].row = sl[0].last = (int32_t)(base + perm[0]);
      sl[1].row = sl[1].last = (int32_t)(base + perm[1]);
      if (p.output_shot_distance) lab = s->pair_label(perm[0], perm[1]);
    } else if (ok && p.context_type != VV_CONTEXT_WINDOW) {
      // :510-757 -- target = the last of the C frames, context = the C-1 before it, in time order
      std::vector<int32_t>& perm = s->perm;
      perm.resize(n);
      for (int i = 0; i < n; ++i) perm[i] = i;
      int begin = 0, stride = 1;
      if (p.context_type == VV_CONTEXT_PAST) {
        s->random_unique(perm, C);                                                 // :517
        std::sort(perm.begin(), perm.begin() + C);                                 // :522
      } else {
        const int msl = (n - C) / (C - 1);                                         // :609,:687
        int sl;
        if (p.context_type == VV_CONTEXT_PAST_CONTINUOUS) {
          sl = s->rmod(msl + 1);                                                   // :610
          begin = s->rmod(n - (C - 1) * sl - C + 1);                               // :612-613
        } else {
          sl = msl >= 1 ? msl - 1 : 0;                                             // :688
          begin = n - (C - 1) * sl - C;                                            // :690-691
        }
        stride = sl + 1;
      }
      for (int i = 0; i < C; ++i) {
        const int frame = p.context_type == VV_CONTEXT_PAST ? perm[i] : begin + i * stride;
        Slot& d = (i == C - 1) ? sl[0] : sl[i + 1];
        d.row = d.last = (int32_t)(base + frame);
      }
      if (p.context_type == VV_CONTEXT_PAST) {
        if (Nn > 0 && n > C && max_same <= 0) {
          s->rng.discard(n - C - 1);
        } else if (Nn > 0 && n > C) {                                              // :563-583
          for (int i = C + 1; i < n; ++i) {
            const int j = C + s->rmod(i - C + 1);
            if (i != j) std::swap(perm[i], perm[j]);
          }
          for (int nid = C; nid < n && added < max_same; ++nid)
            if (perm[nid] < perm[1]) sl[C + added++].row = (int32_t)(base + perm[nid]);   // :570-577, F-1 values
        }
      } else if (Nn > 0 && begin > 0) {                                            // :652-670, :730-748
        for (int nid = begin - 1; nid >= 0 && added < max_same; --nid)
          sl[C + added++].row = (int32_t)(base + nid);
      }
    } else if (ok) {
      std::vector<int32_t>& perm = s->perm;
      perm.resize(n);
      for (int i = 0; i < n; ++i) perm[i] = i;
      s->random_unique(perm, C);                                                   // :432
      std::sort(perm.begin(), perm.begin() + C);                                   // :437
      for (int i = 0, ctx = 0; i < C; ++i) {                                       // :439-453
        const int32_t r = (int32_t)(base + perm[i]);
        Slot& d = (i == half) ? sl[0] : sl[++ctx];
        d.row = d.last = r;
      }
      if (Nn > 0 && n > C && max_same <= 0) {
        s->rng.discard(n - C - 1);                // the shuffle below draws n-C-1 values; its result is unused here
      } else if (Nn > 0 && n > C) {                                                // :479-503
        for (int i = C + 1; i < n; ++i) {         // std::random_shuffle(perm + C, perm + n)
          const int j = C + s->rmod(i - C + 1);
          if (i != j) std::swap(perm[i], perm[j]);
        }
        for (int nid = C; nid < n && added < max_same; ++nid)
          if (perm[nid] < perm[half - 1] || perm[nid] > perm[half + 1])
            sl[C + added++].row = (int32_t)(base + perm[nid]);   // F-1 values copied: .last stays
      }
    }
    s->cursor = (s->cursor + 1) % V;                                               // :826-846
    if (!ok) continue;                                                             // :848
    if (Nn > 0) {
      // :855 random_unique over the persistent slot permutation, fused with :856-875 (read the drawn slots' rows)
      int32_t* ids = s->buffer_ids.data();
      const int32_t* brow = s->buf_row.data();
      int left = p.max_buffer_size;
      for (int first = 0; first < Nn - added; ++first, --left) {
        const int r = first + s->rmod(left);
        const int32_t t = ids[r]; ids[r] = ids[first]; ids[first] = t;
        sl[C + added + first].row = sl[C + added + first].last = brow[t];
      }
    }
    if (label) label[item] = lab;                                                  // :879
    ++item;
    if (Nn > 0 && p.negative_swap_percentage > 0 && s->dense_keys) {               // :888-906, bitmap key set
      uint8_t* inb = s->row_in_buf.data() - s->row_min;      // indexed by table row
      int32_t* brow = s->buf_row.data();
      const int swap = p.negative_swap_percentage, mb = p.max_buffer_size;
      for (int j = 0; j < n; ++j) {
        const int64_t r = base + j;
        if (inb[r]) continue;
        if (s->rng.next() % 100 < swap) {                                          // :27
          const int pos = s->rmod(mb);                                             // :29
          inb[brow[pos]] = 0;
          inb[r] = 1;
          brow[pos] = (int32_t)r;
        }
      }
    } else if (Nn > 0 && p.negative_swap_percentage > 0) {                         // :888-906, general key set
      for (int j = 0; j < n; ++j) {
        if (s->contains(v, j)) continue;
        if (s->rng.next() % 100 < p.negative_swap_percentage) {                    // :27
          const int pos = s->rmod(p.max_buffer_size);                              // :29
          s->keys.erase(s->buf_key[pos]);
          s->insert_key(v, j);
          s->buf_key[pos] = vv_sampler::key(s->video_id[v], s->shot_id(v, j));
          s->buf_row[pos] = (int32_t)(base + j);
        }
      }
    }
  }
  const size_t n = s->slots.size();
  const Slot* sp = s->slots.data();
  if (idx && last_src) for (size_t i = 0; i < n; ++i) { idx[i] = sp[i].row; last_src[i] = sp[i].last; }
  else if (idx) for (size_t i = 0; i < n; ++i) idx[i] = sp[i].row;
  else if (last_src) for (size_t i = 0; i < n; ++i) last_src[i] = sp[i].last;
  return VV_OK;
}

// ---- prefetch threads -------------------------------------------------------------------------------------------
// One thread running whole batches (general path, or threads == 1).
void vv_sampler::run_batches() {
  RingHdr* h = ring->hdr;
  for (int64_t k = 0; !stop.load(std::memory_order_relaxed); ++k) {
    unsigned spins = 0;
    while (k - ring->min_released() >= h->depth) { if (stop.load(std::memory_order_relaxed)) return; backoff(spins); }
    sample_batch(ring->idx_of(k), h->has_last ? ring->last_of(k) : nullptr, ring->label_of(k));
    h->done_negs.store((k + 1) * (int64_t)h->batch_size, std::memory_order_release);
    h->done_frames.store((k + 1) * (int64_t)h->batch_size, std::memory_order_release);
  }
}

void vv_sampler::run_walk() {
  const int B = p.batch_size;
  uint64_t ev_head = 0;
  int64_t it = 0, published = 0, done_seen = 0;
  uint64_t tail_seen = 0;
  const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(32, B));
  const uint64_t ev_cap = ev_mask + 1;
  while (!stop.load(std::memory_order_relaxed)) {
    unsigned spins = 0;
    // the record slot must have been consumed by both later stages, its batch buffer released by every consumer,
    // and the event ring must have room for one more video
    uint64_t tw = 0;
    // (the later stages' counters as last read: they only grow, so a stale value errs on the safe side -- they are read again only
    // when it says "no room".  Read for every item, each was a line the other core had just written: a miss on the walk's path per item.)
    for (;;) {
      if (it - done_seen < ring_items && ev_head + (uint64_t)max_n <= tail_seen + ev_cap) break;
      done_seen = std::min(ring->hdr->done_negs.load(std::memory_order_acquire), ring->hdr->done_frames.load(std::memory_order_acquire));
      tail_seen = ev_tail.load(std::memory_order_acquire);
      if (it - done_seen < ring_items && ev_head + (uint64_t)max_n <= tail_seen + ev_cap) break;
      if (published < it) { walked.store(it, std::memory_order_release); published = it; }
      if (stop.load(std::memory_order_relaxed)) return;
      if (!tw) tw = __rdtsc();
      backoff(spins);
    }
    if (tw) wait_walk.store(wait_walk.load(std::memory_order_relaxed) + (__rdtsc() - tw), std::memory_order_relaxed);
    uint32_t* rec = recs.data() + (size_t)(it % ring_items) * rec_words;
    select_item(rec);
#ifdef VV_SAMPLER_LAB
    if (g_plab & 16) swap<false>(rec, buf_row.data(), nullptr, 0, nullptr); else
#endif
    swap<true>(rec, buf_row.data(), events.data(), ev_mask, &ev_head);
    ++it;
    if (it - published >= chunk || it % B == 0) { walked.store(it, std::memory_order_release); published = it; }
  }
}

void vv_sampler::run_negs(bool also_frames) {
  const int B = p.batch_size, C = p.context_size, CN = C + p.num_negative_samples;
  RingHdr* h = ring->hdr;
  int64_t it = 0, published = 0, avail = 0;
  uint64_t ev_done = 0;
  int32_t* brow = buf_row_negs.data();
  while (!stop.load(std::memory_order_relaxed)) {
    unsigned spins = 0;
    uint64_t tw = 0;
    while (it >= avail) {
      avail = walked.load(std::memory_order_acquire);
      if (it < avail) break;
      if (published < it) { ev_tail.store(ev_done, std::memory_order_release); ring->hdr->done_negs.store(it, std::memory_order_release); if (also_frames) ring->hdr->done_frames.store(it, std::memory_order_release); published = it; }
      if (stop.load(std::memory_order_relaxed)) return;
      if (!tw) tw = __rdtsc();
      backoff(spins);
    }
    const int64_t k = it / B;
    if (it % B == 0) {
      spins = 0;
      while (k - ring->min_released() >= h->depth) { if (stop.load(std::memory_order_relaxed)) return; if (!tw) tw = __rdtsc(); backoff(spins); }
    }
    if (tw) wait_negs.store(wait_negs.load(std::memory_order_relaxed) + (__rdtsc() - tw), std::memory_order_relaxed);
    const uint32_t* rec = recs.data() + (size_t)(it % ring_items) * rec_words;
    int32_t* out = ring->idx_of(k) + (size_t)(it % B) * CN;
#ifdef VV_SAMPLER_LAB
    static uint32_t fake[512];
    negs((g_plab & 2) ? fake : rec, out + C, brow);
#else
    negs(rec, out + C, brow);
#endif
    if (also_frames) frames_item(rec, out, ring->label_of(k) + (it % B));
    const uint64_t e0 = (uint64_t)rec[2] | ((uint64_t)rec[3] << 32);
#ifdef VV_SAMPLER_LAB
    const uint32_t nev = (g_plab & 1) ? 0 : rec[1];
    if (g_plab & 1) ev_done = e0 + rec[1];
#else
    const uint32_t nev = rec[1];
#endif
    for (uint32_t e = 0; e < nev; ++e) { const Event ev = events[(e0 + e) & ev_mask]; brow[ev.pos] = ev.row; }
    if (nev) ev_done = e0 + nev;
    ++it;
    if (it - published >= 32 || it % B == 0) {
      ev_tail.store(ev_done, std::memory_order_release);        // (with the item counter, not per item: the walk reads this line)
      ring->hdr->done_negs.store(it, std::memory_order_release);
      if (also_frames) ring->hdr->done_frames.store(it, std::memory_order_release);
      published = it;
    }
  }
}

void vv_sampler::run_frames() {
  const int B = p.batch_size, CN = p.context_size + p.num_negative_samples;
  RingHdr* h = ring->hdr;
  int64_t it = 0, published = 0, avail = 0;
  while (!stop.load(std::memory_order_relaxed)) {
    unsigned spins = 0;
    uint64_t tw = 0;
    while (it >= avail) {
      avail = walked.load(std::memory_order_acquire);
      if (it < avail) break;
      if (published < it) { ring->hdr->done_frames.store(it, std::memory_order_release); published = it; }
      if (stop.load(std::memory_order_relaxed)) return;
      if (!tw) tw = __rdtsc();
      backoff(spins);
    }
    const int64_t k = it / B;
    if (it % B == 0) {
      spins = 0;
      while (k - ring->min_released() >= h->depth) { if (stop.load(std::memory_order_relaxed)) return; if (!tw) tw = __rdtsc(); backoff(spins); }
    }
    if (tw) wait_frames.store(wait_frames.load(std::memory_order_relaxed) + (__rdtsc() - tw), std::memory_order_relaxed);
    const uint32_t* rec = recs.data() + (size_t)(it % ring_items) * rec_words;
#ifdef VV_SAMPLER_LAB
    if (!(g_plab & 4))
#endif
    frames_item(rec, ring->idx_of(k) + (size_t)(it % B) * CN, ring->label_of(k) + (it % B));
    ++it;
    if (it - published >= 32 || it % B == 0) { ring->hdr->done_frames.store(it, std::memory_order_release); published = it; }
  }
}

int vv_sampler::sample_batch(int32_t* idx, int32_t* last_src, int32_t* label) {
  if (!fast) return next_general(idx, last_src, label);
  const int B = p.batch_size, C = p.context_size, CN = C + p.num_negative_samples;
  std::vector<int32_t> tmp;
  int32_t* out = idx;
  if (!out) { tmp.resize((size_t)B * CN); out = tmp.data(); }
  uint32_t* rec = rec1.data();
  for (int it = 0; it < B; ++it) {
    int32_t* o = out + (size_t)it * CN;
    select_item(rec);
    negs(rec, o + C, buf_row.data());       // the buffer as it stands BEFORE this item's swap-in (:855-875 precede :888-906)
    frames_item(rec, o, label ? label + it : nullptr);
    swap<false>(rec, buf_row.data(), nullptr, 0, nullptr);
  }
  if (last_src) memcpy(last_src, out, (size_t)B * CN * 4);     // no same-video negatives on this path: last == row
  return VV_OK;
}

extern "C" {

void vv_sampler_param_default(vv_sampler_param* p) {
  memset(p, 0, sizeof(*p));
  p->batch_size = 128; p->context_size = 5; p->num_negative_samples = 10;   // shipped prototxt :13-23
  p->max_buffer_size = 5000; p->negative_swap_percentage = 50; p->max_same_video_negs = 0;
  p->max_tries_for_negs = 100;
  p->max_shot_distance = 5.f;                                                      // caffe.proto:674
}

int vv_sampler_create(const vv_sampler_param* p, int32_t n_videos, const int32_t* video_id,
                      const int32_t* n_shots, const int64_t* row_base, const int32_t* shot_ids,
                      vv_sampler** out) {
  return vv_sampler_create_neg(p, n_videos, video_id, n_shots, row_base, shot_ids, 0, nullptr, nullptr, nullptr, nullptr, out);
}

int vv_sampler_create_neg(const vv_sampler_param* p_in, int32_t n_videos, const int32_t* video_id,
                          const int32_t* n_shots, const int64_t* row_base, const int32_t* shot_ids,
                          int32_t neg_videos, const int32_t* neg_video_id, const int32_t* neg_n_shots,
                          const int64_t* neg_row_base, const int32_t* neg_shot_ids, vv_sampler** out) {
  if (!p_in || !video_id || !n_shots || !row_base || !out || n_videos < 1) return VV_ERR_ARG;
  if (neg_videos < 0 || (neg_videos > 0 && (!neg_video_id || !neg_n_shots || !neg_row_base))) return VV_ERR_ARG;
  vv_sampler_param pp = *p_in;
  if (pp.context_type == VV_CONTEXT_PAIRWISE) pp.context_size = 2;                // :200-201
  const vv_sampler_param* p = &pp;
  if (p->batch_size < 1 || p->context_size < 2) return VV_ERR_ARG;                // :207,:209
  if (p->context_type < VV_CONTEXT_WINDOW || p->context_type > VV_CONTEXT_PAIRWISE) return VV_ERR_ARG;   // :760
  if (p->context_type == VV_CONTEXT_WINDOW && p->context_size % 2 != 1) return VV_ERR_ARG;   // :434
  const int Nn = p->num_negative_samples;
  if (Nn < 0) return VV_ERR_ARG;
  if (Nn > 0 && (p->negative_swap_percentage < 0 || p->negative_swap_percentage > 99 ||
                 p->max_buffer_size < Nn)) return VV_ERR_ARG;                     // :79-80
  // The reference writes same-video negatives at top_data offset C + added without ever comparing added with
  // num_negative_samples (:484-502): more same-video negatives than negative slots runs into the next item's slots
  // (undefined behaviour there).  Rejected here.
  if (p->max_same_video_negs > Nn) return VV_ERR_ARG;
  vv_sampler* s = new (std::nothrow) vv_sampler();
  if (!s) return VV_ERR_STATE;
  s->p = *p;
  s->video_id.assign(video_id, video_id + n_videos);
  s->n_shots.assign(n_shots, n_shots + n_videos);
  s->row_base.assign(row_base, row_base + n_videos);
  int max_n = 1; int64_t total = 0;
  bool any_ok = false;
  s->shot_off.resize(n_videos);
  for (int v = 0; v < n_videos; ++v) {
    if (n_shots[v] < 1) { delete s; return VV_ERR_ARG; }                         // :808
    s->shot_off[v] = total; total += n_shots[v]; max_n = std::max(max_n, n_shots[v]);
    any_ok = any_ok || (n_shots[v] >= 2 && n_shots[v] >= p->context_size);
  }
  if (!any_ok) { delete s; return VV_ERR_ARG; }                                   // the reference would spin forever (:796-848)
  s->max_n = max_n;
  if (shot_ids) { s->has_ids = true; s->shot_ids.assign(shot_ids, shot_ids + total); }
  s->perm.reserve(max_n);
  s->fm.init(std::max(std::max(max_n, p->max_buffer_size), 100) + 1);
  {  // keys are in bijection with rows iff video ids are distinct, shot ids distinct within a video and
     // the records' row ranges do not overlap
    std::unordered_set<int32_t> vids(video_id, video_id + n_videos);
    bool ok = (int)vids.size() == n_videos;
    int64_t lo = row_base[0], hi = row_base[0];
    std::vector<std::pair<int64_t, int64_t>> ranges;
    for (int v = 0; v < n_videos && ok; ++v) {
      lo = std::min(lo, row_base[v]); hi = std::max(hi, row_base[v] + n_shots[v]);
      ranges.emplace_back(row_base[v], row_base[v] + n_shots[v]);
      if (shot_ids) {
        std::unordered_set<int32_t> sids(shot_ids + s->shot_off[v], shot_ids + s->shot_off[v] + n_shots[v]);
        ok = (int)sids.size() == n_shots[v];
      }
    }
    if (ok) {
      std::sort(ranges.begin(), ranges.end());
      for (size_t i = 1; i < ranges.size() && ok; ++i) ok = ranges[i].first >= ranges[i - 1].second;
    }
    if (ok && neg_videos == 0 && lo >= 0 && hi < (1ll << 31) - 1) {     // a negative dataset brings keys that are not rows of this one
      s->dense_keys = true; s->row_min = lo;
      s->row_in_buf.assign((size_t)(hi - lo) + 64, 0);                            // 64 bytes of slack: the swap-in reads whole vectors
    }
  }
  if (p->initial_cursor < 0 || p->rand_seed < 0 || p->rand_seed == 2147483647) { delete s; return VV_ERR_ARG; }
  s->cursor = p->initial_cursor % n_videos;                                       // rand_skip, :156-180
  const int C = p->context_size, CN = C + Nn;
  s->fast = s->dense_keys && p->max_same_video_negs <= 0;
  s->CA = (p->context_type == VV_CONTEXT_WINDOW || p->context_type == VV_CONTEXT_PAST || p->context_type == VV_CONTEXT_PAIRWISE) ? C
          : (p->context_type == VV_CONTEXT_PAST_CONTINUOUS ? 2 : 0);
  s->rec_words = 4 + s->CA + Nn;
  {
    const char* e = getenv("VV_SAMPLER_AVX512");
    s->wide = !(e && atoi(e) == 0) && __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512dq") &&
              __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512vbmi2") && __builtin_cpu_supports("bmi2");
    s->inv_mb = p->max_buffer_size > 0 ? 1.0 / (double)p->max_buffer_size : 0.0;
  }
  s->rec1.assign((size_t)s->rec_words, 0u);
  {
    const int64_t per_item = (int64_t)C + Nn + 3ll * max_n + 32 + 160;
    int64_t want = 16384;
    if (const char* e = getenv("VV_SAMPLER_BLOCK")) want = std::max<int64_t>(1024, atoll(e));      // (tuning: words per generated block)
    const int64_t block = (std::max<int64_t>(want, 4 * per_item) + 15) / 16 * 16;
    if (block > (1ll << 28)) { delete s; return VV_ERR_ARG; }
    s->rng.init((int)block, Nn > 0 ? p->negative_swap_percentage : 0, (uint32_t)p->rand_seed, s->wide, Nn > 0 ? p->max_buffer_size : 0);
  }
  if (!s->fast) s->slots.assign((size_t)p->batch_size * CN, Slot());
  const int mb = Nn > 0 ? p->max_buffer_size : 0;
  s->buffer_ids.resize(mb);
  for (int i = 0; i < mb; ++i) s->buffer_ids[i] = i;
  s->buf_row.reserve(mb); s->buf_key.reserve(mb);
  // fill the negative buffer: one random shot of each visited record until full (:240-344)
  if (mb > 0 && neg_videos > 0) {
    // negative_dataset (:105-151 opens it, :253-286 its own cursor from the first record, :325-341 EVERY shot whose
    // key is new; no rand(), the main cursor stays).  The reference tests for a full buffer only after a whole record
    // (:343) and writes past negatives_ when the last record overshoots: only an exact fit passes its CHECK_EQ (:348).
    const int64_t tries = (int64_t)p->max_tries_for_negs * mb;
    int64_t off = 0; int cur = 0;
    std::vector<int64_t> neg_off(neg_videos);
    for (int v = 0; v < neg_videos; ++v) { if (neg_n_shots[v] < 1) { delete s; return VV_ERR_ARG; } neg_off[v] = off; off += neg_n_shots[v]; }
    for (int64_t t = 0; t < tries && (int)s->buf_row.size() < mb; ++t) {
      const int v = cur;
      cur = (cur + 1) % neg_videos;
      for (int j = 0; j < neg_n_shots[v]; ++j) {
        const uint64_t k = vv_sampler::key(neg_video_id[v], neg_shot_ids ? neg_shot_ids[neg_off[v] + j] : j);
        if (s->keys.count(k)) continue;
        if ((int)s->buf_row.size() >= mb) { delete s; return VV_ERR_ARG; }
        s->keys.insert(k);
        s->buf_row.push_back((int32_t)(neg_row_base[v] + j));
        s->buf_key.push_back(k);
      }
    }
    if ((int)s->buf_row.size() != mb) { delete s; return VV_ERR_ARG; }
  } else if (mb > 0) {
    const int64_t tries = (int64_t)p->max_tries_for_negs * mb;
    for (int64_t t = 0; t < tries && (int)s->buf_row.size() < mb; ++t) {
      const int v = s->cursor;
      s->cursor = (s->cursor + 1) % n_videos;
      const int j = s->rmod(s->n_shots[v]);
      if (!s->contains(v, j)) {
        s->insert_key(v, j);
        s->buf_row.push_back((int32_t)(s->row_base[v] + j));
        s->buf_key.push_back(vv_sampler::key(s->video_id[v], s->shot_id(v, j)));
      }
    }
    if ((int)s->buf_row.size() != mb) { delete s; return VV_ERR_ARG; }           // :344
  }
  *out = s;
  return VV_OK;
}

static int pop_batch(vv_batch_ring* r, int consumer, int32_t item_begin, int32_t item_count, int32_t* idx,
                     int32_t* last_src, int32_t* label, double timeout_s) {
  RingHdr* h = r->hdr;
  if (consumer < 0 || consumer >= h->consumers) return VV_ERR_ARG;
  if (item_begin < 0 || item_count < 0 || item_begin + item_count > h->batch_size) return VV_ERR_ARG;
  const int64_t k = r->next_k[consumer];
  unsigned spins = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (!r->ready(k)) {
    if (h->closed.load(std::memory_order_acquire)) return VV_ERR_STATE;
    backoff(spins);
    if (timeout_s > 0 && (spins & 1023) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return VV_ERR_STATE;
  }
  const size_t cn = (size_t)h->cn;
  if (idx) memcpy(idx, r->idx_of(k) + (size_t)item_begin * cn, (size_t)item_count * cn * 4);
  if (last_src) memcpy(last_src, (h->has_last ? r->last_of(k) : r->idx_of(k)) + (size_t)item_begin * cn, (size_t)item_count * cn * 4);
  if (label) memcpy(label, r->label_of(k) + item_begin, (size_t)item_count * 4);
  r->next_k[consumer] = k + 1;
  h->released[consumer].v.store(k + 1, std::memory_order_release);
  return VV_OK;
}

int vv_sampler_next(vv_sampler* s, int32_t* idx, int32_t* last_src, int32_t* label) {
  if (!s) return VV_ERR_ARG;
  if (s->ring) return pop_batch(s->ring, 0, 0, s->p.batch_size, idx, last_src, label, 0.0);
  return s->sample_batch(idx, last_src, label);
}

int vv_sampler_prefetch_start(vv_sampler* s, int32_t depth, int32_t threads, const char* shm_name, int32_t consumers) {
  if (!s || s->ring || depth < 1 || depth > 1024 || threads < 1 || consumers < 1 || consumers > VV_RING_MAX_CONSUMERS) return VV_ERR_ARG;
  const int B = s->p.batch_size, CN = s->p.context_size + s->p.num_negative_samples;
  s->ring = ring_create(shm_name, depth, B, CN, consumers, !s->fast);
  if (!s->ring) return VV_ERR_STATE;
  s->stop.store(0);
#ifdef VV_SAMPLER_LAB
  g_plab = getenv("VV_SAMPLER_LAB") ? atoi(getenv("VV_SAMPLER_LAB")) : 0;
#endif
  if (!s->fast || threads == 1) {
    s->n_stage_threads = 1;
    cpu_set_t set; const bool pin = stage_cpu_set(&set);
    s->threads.emplace_back([s, set, pin]() { pin_self(pin ? &set : nullptr); s->run_batches(); });
    return VV_OK;
  }
  // staged pipeline
  s->ring_items = (int64_t)B * std::min<int64_t>(depth, 4);
  if (s->ring_items < 256) s->ring_items = 256;
  if (const char* e = getenv("VV_SAMPLER_RING_ITEMS")) s->ring_items = std::max<int64_t>(64, atoll(e));      // (tuning: item records between the walk and the later stages)
  s->recs.assign((size_t)s->ring_items * s->rec_words, 0u);
  uint64_t cap = 1; while (cap < (uint64_t)s->ring_items * 32 + 4ull * s->max_n) cap <<= 1;
  s->events.assign(cap, vv_sampler::Event{0, 0}); s->ev_mask = cap - 1;
  s->buf_row_negs = s->buf_row;
  s->walked.store(0); s->ev_tail.store(0);
  s->wait_walk.store(0); s->wait_negs.store(0); s->wait_frames.store(0); s->tsc_start = __rdtsc();
  const bool three = threads >= 3;
  s->n_stage_threads = threads >= 4 ? 4 : (three ? 3 : 2);
  cpu_set_t set, own[8]; int n_own = 0, first_cpu[8];
  const bool pin = stage_cpu_set(&set, own, &n_own, first_cpu);
  if (pin && n_own >= 4 && !(getenv("VV_SAMPLER_SPREAD") && atoi(getenv("VV_SAMPLER_SPREAD")) == 0)) {
    n_own = claim_cores(own, first_cpu, n_own, 4, s->core_fds);
    s->n_core_fds = n_own;
  } else n_own = 0;
  // VV_SAMPLER_PLACE="walk,stream,negs,frames": one CPU number per stage thread (-1 / missing: the common set) -- which stages share
  // a core (SMT siblings share its L1 / L2: the walk reads what the stream thread writes) is a property of the host, measured, not guessed
  cpu_set_t one[4]; bool has[4] = {false, false, false, false};
  if (const char* e = getenv("VV_SAMPLER_PLACE")) {
    int k = 0;
    for (const char* q = e; *q && k < 4; ++k) {
      const int c = atoi(q);
      if (c >= 0 && c < CPU_SETSIZE) { CPU_ZERO(&one[k]); CPU_SET(c, &one[k]); has[k] = true; }
      while (*q && *q != ',') ++q;
      if (*q) ++q;
    }
  }
  const bool spread = n_own >= 4;
  const cpu_set_t sw = has[0] ? one[0] : (spread ? own[0] : set), ss = has[1] ? one[1] : (spread ? own[1] : set);
  const cpu_set_t sn = has[2] ? one[2] : (spread ? own[2] : set), sf = has[3] ? one[3] : (spread ? own[3] : set);
  const bool pw = pin || has[0], ps = pin || has[1], pn = pin || has[2], pf = pin || has[3];
  if (threads >= 4) {            // the fourth thread generates the rand() stream a block ahead of the walk (LibcRand::start_helper)
    s->rng.pin_helper_like_caller(ps ? &ss : nullptr);
    s->rng.start_helper();
  }
  s->threads.emplace_back([s, sw, pw]() { pin_self(pw ? &sw : nullptr); s->run_walk(); });
  s->threads.emplace_back([s, three, sn, pn]() { pin_self(pn ? &sn : nullptr); s->run_negs(!three); });
  if (three) s->threads.emplace_back([s, sf, pf]() { pin_self(pf ? &sf : nullptr); s->run_frames(); });
  return VV_OK;
}

int vv_sampler_prefetch_stop(vv_sampler* s) {
  if (!s) return VV_ERR_ARG;
  if (!s->ring) return VV_OK;
  s->stop.store(1);
  for (auto& t : s->threads) if (t.joinable()) t.join();
  s->threads.clear();
  s->rng.stop_helper();
  for (int i = 0; i < s->n_core_fds; ++i) close(s->core_fds[i]);
  s->n_core_fds = 0;
  ring_free(s->ring);
  s->ring = nullptr;
  s->n_stage_threads = 0;
  return VV_OK;
}

int vv_sampler_ring(vv_sampler* s, vv_batch_ring** out) {
  if (!s || !out || !s->ring) return VV_ERR_ARG;
  *out = s->ring;
  return VV_OK;
}

int64_t vv_sampler_stat(vv_sampler* s, int32_t which) {
  if (!s) return -1;
  switch (which) {
    case 0: return s->stat_restarts;             // swap-in walks restarted because a later shot of the video was evicted
    case 1: return s->fast ? 1 : 0;              // staged fast path in use
    case 2: return s->n_stage_threads;           // producer threads of the running prefetch (0 = none)
    case 3: return (int64_t)(__rdtsc() - s->tsc_start);                  // ticks since the prefetch pipeline started
    case 4: return (int64_t)s->wait_walk.load(std::memory_order_relaxed);     // ... of which the walk stage spent waiting
    case 5: return (int64_t)s->wait_negs.load(std::memory_order_relaxed);     // ... the negative-slot stage
    case 6: return (int64_t)s->wait_frames.load(std::memory_order_relaxed);   // ... the frame stage
    case 7: return s->wide ? 1 : 0;              // 512-bit forms in use
    case 9: return s->n_core_fds;                // cores claimed for the stage threads (4 = one each; 0 = the common set)
    case 8: return (int64_t)s->rng.wait_ticks(); // ticks the walk stage waited for the stream-generating thread's next block (read when the pipeline is quiet or approximately)
    default: return -1;
  }
}

int vv_sampler_destroy(vv_sampler* s) {
  if (s) vv_sampler_prefetch_stop(s);
  delete s;
  return VV_OK;
}

int vv_batch_ring_attach(const char* shm_name, double timeout_s, vv_batch_ring** out) {
  if (!shm_name || !*shm_name || !out) return VV_ERR_ARG;
  const std::string name = shm_name[0] == '/' ? shm_name : std::string("/") + shm_name;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const int fd = shm_open(name.c_str(), O_RDWR, 0600);
    if (fd >= 0) {
      struct stat st;
      if (fstat(fd, &st) == 0 && (size_t)st.st_size >= ring_hdr_bytes()) {
        void* mem = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (mem != MAP_FAILED) {
          RingHdr* h = (RingHdr*)mem;
          if (h->magic == kRingMagic) {
            std::atomic_thread_fence(std::memory_order_acquire);
            vv_batch_ring* r = new (std::nothrow) vv_batch_ring();
            if (!r) { munmap(mem, (size_t)st.st_size); return VV_ERR_STATE; }
            r->hdr = h; r->data = (unsigned char*)mem + ring_hdr_bytes(); r->map_bytes = (size_t)st.st_size;
            r->shm = true; r->owner = false; r->name = name;
            for (int i = 0; i < VV_RING_MAX_CONSUMERS; ++i) r->next_k[i] = h->released[i].v.load();
            *out = r;
            return VV_OK;
          }
          munmap(mem, (size_t)st.st_size);
        }
      } else close(fd);
    }
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return VV_ERR_STATE;
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
  }
}

int vv_batch_ring_info(vv_batch_ring* r, int32_t* batch_size, int32_t* slots_per_item, int32_t* consumers, int32_t* depth) {
  if (!r) return VV_ERR_ARG;
  if (batch_size) *batch_size = r->hdr->batch_size;
  if (slots_per_item) *slots_per_item = r->hdr->cn;
  if (consumers) *consumers = r->hdr->consumers;
  if (depth) *depth = r->hdr->depth;
  return VV_OK;
}

int vv_batch_ring_next(vv_batch_ring* r, int32_t consumer, int32_t item_begin, int32_t item_count, int32_t* idx,
                       int32_t* label, double timeout_s) {
  if (!r) return VV_ERR_ARG;
  return pop_batch(r, consumer, item_begin, item_count, idx, nullptr, label, timeout_s);
}

int vv_batch_ring_detach(vv_batch_ring* r) {
  if (!r) return VV_OK;
  if (r->owner) return VV_ERR_ARG;            // the sampler owns its ring (vv_sampler_prefetch_stop)
  ring_free(r);
  return VV_OK;
}

}  // extern "C"
