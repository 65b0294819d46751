// sampler.cc -- host-side triplet sampler of the MI355X videovec path (product code; the test
// oracle under oracle/ is a separate, independent restatement and is never linked here).
//
// Reproduces, with table-row indices in place of feature copies, the reference's
// VideoSampledShotsDataLayer in CONTEXT_WINDOW mode:
//   setup   src/caffe/layers/video_sampled_shots_data_layer.cpp:64-369
//   batch   ...:768-909 (InternalThreadEntry), :371-393,425-507 (AddSamplesToTop), :24-44
// including the exact consumption order of the C library's rand() stream.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <unordered_set>
#include <vector>

#include "../../include/videovec.h"

namespace {

// glibc rand() == random() with the default TYPE_3 state: 31 words seeded from seed 1 by the
// 16807 Lehmer recurrence, then state[f] += state[f-3] walking f cyclically; output is the new
// word >> 1; the first 310 outputs are discarded by srandom.  (glibc 2.35 stdlib/random_r.c)
class LibcRand {
 public:
  LibcRand() {
    uint32_t st[31];
    int64_t w = 1;
    st[0] = 1;
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
    // Linear history: the word at st[f] is the oldest, x[k-31].  From here on the stream is
    // x[k] = x[k-31] + x[k-3]; it does not depend on how the outputs are consumed, so it is produced a
    // block at a time (three independent dependency chains) instead of one value per call.
    for (int i = 0; i < 31; ++i) h_[i] = st[(f + i) % 31];
    pos_ = kBlock;
  }
  int32_t next() {
    if (pos_ == kBlock) refill();
    return (int32_t)(h_[31 + pos_++] >> 1);
  }
  // consume k values whose results nobody looks at (a shuffle of frames that are not used afterwards)
  void discard(int k) {
    while (k > 0) {
      if (pos_ == kBlock) refill();
      const int step = std::min(k, kBlock - pos_);
      pos_ += step; k -= step;
    }
  }

 private:
  static constexpr int kBlock = 1024;
  void refill() {
    if (pos_ == kBlock && filled_) memcpy(h_, h_ + kBlock, 31 * sizeof(uint32_t));
    for (int i = 31; i < 31 + kBlock; ++i) h_[i] = h_[i - 31] + h_[i - 3];
    pos_ = 0; filled_ = true;
  }
  uint32_t h_[31 + kBlock];
  int pos_;
  bool filled_ = false;
};

struct Slot { int32_t row = -1, last = -1; };

// a % d for 0 <= a < 2^32 and the divisors this sampler meets (1 .. max(max_buffer_size, longest video, 100)),
// by two multiplications with a precomputed reciprocal instead of a hardware division (Lemire, Kaser, Kurz:
// "Faster remainder by direct computation", 2019): M = floor((2^64 - 1) / d) + 1, a % d = floor(((M * a) mod 2^64) * d / 2^64).
// The sampler is a chain of ~110 rand() % k per batch item; on the host this chain, not memory, sets its speed.
class FastMod {
 public:
  void init(int dmax) {
    m_.resize((size_t)dmax + 1);
    for (int d = 1; d <= dmax; ++d) m_[d] = UINT64_C(0xFFFFFFFFFFFFFFFF) / (uint64_t)d + 1;
  }
  int32_t mod(int32_t a, int32_t d) const {
    const uint64_t low = m_[d] * (uint64_t)(uint32_t)a;
    return (int32_t)(((unsigned __int128)low * (uint64_t)d) >> 64);
  }
 private:
  std::vector<uint64_t> m_;
};

}  // namespace

struct vv_sampler {
  vv_sampler_param p;
  std::vector<int32_t> video_id, n_shots, shot_ids;
  std::vector<int64_t> row_base, shot_off;
  bool has_ids = false;
  LibcRand rng;
  FastMod fm;
  int32_t rmod(int32_t d) { return fm.mod(rng.next(), d); }     // rand() % d
  int32_t cursor = 0;
  std::vector<int32_t> buffer_ids;            // persistent permutation (…data_layer.cpp:81-83)
  std::vector<int32_t> buf_row;               // slot -> table row
  std::vector<uint64_t> buf_key;              // slot -> (video_id, shot_id)
  std::unordered_set<uint64_t> keys;          // negative_keys_set_ (general case)
  // fast path: when every (video_id, shot_id) key names exactly one table row, membership in the key
  // set is a bitmap over table rows
  bool dense_keys = false;
  std::vector<uint8_t> row_in_buf;
  bool contains(int v, int j) const {
    return dense_keys ? row_in_buf[(size_t)(row_base[v] - row_min + j)] != 0 : keys.count(key(video_id[v], shot_id(v, j))) != 0;
  }
  void insert_key(int v, int j) {
    if (dense_keys) row_in_buf[(size_t)(row_base[v] - row_min + j)] = 1; else keys.insert(key(video_id[v], shot_id(v, j)));
  }
  int64_t row_min = 0;
  std::vector<Slot> slots;                    // persistent prefetch_data_ contents [B][C+Nn]
  std::vector<int32_t> perm;

  static uint64_t key(int32_t vid, int32_t shot) { return ((uint64_t)(uint32_t)vid << 32) | (uint32_t)shot; }
  int32_t shot_id(int v, int j) const { return has_ids ? shot_ids[shot_off[v] + j] : j; }

  // include/caffe/util/rng.hpp:43-54
  void random_unique(std::vector<int32_t>& a, int n) {
    int left = (int)a.size();
    for (int first = 0; first < n; ++first, --left) std::swap(a[first], a[first + rmod(left)]);
  }
};

extern "C" {

void vv_sampler_param_default(vv_sampler_param* p) {
  memset(p, 0, sizeof(*p));
  p->batch_size = 128; p->context_size = 5; p->num_negative_samples = 10;   // shipped prototxt :13-23
  p->max_buffer_size = 5000; p->negative_swap_percentage = 50; p->max_same_video_negs = 0;
  p->max_tries_for_negs = 100;
}

int vv_sampler_create(const vv_sampler_param* p, int32_t n_videos, const int32_t* video_id,
                      const int32_t* n_shots, const int64_t* row_base, const int32_t* shot_ids,
                      vv_sampler** out) {
  if (!p || !video_id || !n_shots || !row_base || !out || n_videos < 1) return VV_ERR_ARG;
  if (p->batch_size < 1 || p->context_size < 2) return VV_ERR_ARG;                // :207,:209
  if (p->context_type < VV_CONTEXT_WINDOW || p->context_type > VV_CONTEXT_PAST_CONTINUOUS_FIXED) return VV_ERR_ARG;   // :760
  if (p->context_type == VV_CONTEXT_WINDOW && p->context_size % 2 != 1) return VV_ERR_ARG;   // :434
  const int Nn = p->num_negative_samples;
  if (Nn < 0) return VV_ERR_ARG;
  if (Nn > 0 && (p->negative_swap_percentage < 0 || p->negative_swap_percentage > 99 ||
                 p->max_buffer_size < Nn)) return VV_ERR_ARG;                     // :79-80
  vv_sampler* s = new (std::nothrow) vv_sampler();
  if (!s) return VV_ERR_STATE;
  s->p = *p;
  s->video_id.assign(video_id, video_id + n_videos);
  s->n_shots.assign(n_shots, n_shots + n_videos);
  s->row_base.assign(row_base, row_base + n_videos);
  int max_n = 1; int64_t total = 0;
  s->shot_off.resize(n_videos);
  for (int v = 0; v < n_videos; ++v) {
    if (n_shots[v] < 1) { delete s; return VV_ERR_ARG; }                         // :808
    s->shot_off[v] = total; total += n_shots[v]; max_n = std::max(max_n, n_shots[v]);
  }
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
    if (ok && hi - lo < (1ll << 31)) { s->dense_keys = true; s->row_min = lo; s->row_in_buf.assign((size_t)(hi - lo), 0); }
  }
  if (p->initial_cursor < 0) { delete s; return VV_ERR_ARG; }
  s->cursor = p->initial_cursor % n_videos;                                       // rand_skip, :156-180
  const int CN = p->context_size + Nn;
  s->slots.assign((size_t)p->batch_size * CN, Slot());
  const int mb = Nn > 0 ? p->max_buffer_size : 0;
  s->buffer_ids.resize(mb);
  for (int i = 0; i < mb; ++i) s->buffer_ids[i] = i;
  s->buf_row.reserve(mb); s->buf_key.reserve(mb);
  // fill the negative buffer: one random shot of each visited record until full (:240-344)
  if (mb > 0) {
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

int vv_sampler_next(vv_sampler* s, int32_t* idx, int32_t* last_src, int32_t* label) {
  if (!s) return VV_ERR_ARG;
  const vv_sampler_param& p = s->p;
  const int C = p.context_size, Nn = p.num_negative_samples, CN = C + Nn, half = C / 2;
  const int V = (int)s->video_id.size();
  for (int item = 0; item < p.batch_size;) {
    const int v = s->cursor;
    const int n = s->n_shots[v];
    const int64_t base = s->row_base[v];
    Slot* sl = &s->slots[(size_t)item * CN];
    int added = 0;
    const bool ok = n >= 2 && n >= C;                                              // :387,:427,:512,:601,:679
    if (ok && p.context_type != VV_CONTEXT_WINDOW) {
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
        if (Nn > 0 && n > C && p.max_same_video_negs <= 0) {
          s->rng.discard(n - C - 1);
        } else if (Nn > 0 && n > C) {                                              // :563-583
          for (int i = C + 1; i < n; ++i) {
            const int j = C + s->rmod(i - C + 1);
            if (i != j) std::swap(perm[i], perm[j]);
          }
          for (int nid = C; nid < n && added < p.max_same_video_negs; ++nid)
            if (perm[nid] < perm[1]) sl[C + added++].row = (int32_t)(base + perm[nid]);   // :570-577, F-1 values
        }
      } else if (Nn > 0 && begin > 0) {                                            // :652-670, :730-748
        for (int nid = begin - 1; nid >= 0 && added < p.max_same_video_negs; --nid)
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
      if (Nn > 0 && n > C && p.max_same_video_negs <= 0) {
        s->rng.discard(n - C - 1);                // the shuffle below draws n-C-1 values; its result is unused here
      } else if (Nn > 0 && n > C) {                                                // :479-503
        for (int i = C + 1; i < n; ++i) {         // std::random_shuffle(perm + C, perm + n)
          const int j = C + s->rmod(i - C + 1);
          if (i != j) std::swap(perm[i], perm[j]);
        }
        for (int nid = C; nid < n && added < p.max_same_video_negs; ++nid)
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
    if (label) label[item] = s->video_id[v];                                       // :879
    ++item;
    if (Nn > 0 && p.negative_swap_percentage > 0 && s->dense_keys) {               // :888-906, bitmap key set
      uint8_t* inb = s->row_in_buf.data() - s->row_min;      // indexed by table row
      int32_t* brow = s->buf_row.data();
      const int swap = p.negative_swap_percentage, mb = p.max_buffer_size;
      const int32_t vid = s->video_id[v];
      for (int j = 0; j < n; ++j) {
        const int64_t r = base + j;
        if (inb[r]) continue;
        if (s->rng.next() % 100 < swap) {                                          // :27
          const int pos = s->rmod(mb);                                             // :29
          inb[brow[pos]] = 0;
          inb[r] = 1;
          s->buf_key[pos] = vv_sampler::key(vid, s->shot_id(v, j));
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

int vv_sampler_destroy(vv_sampler* s) { delete s; return VV_OK; }

}  // extern "C"
