/*
 * vv_oracle.c -- CPU restatement of the reference's videovec_embedding training path.
 * TEST INFRASTRUCTURE ONLY (see vv_oracle.h).  Plain C11 + OpenMP; every function cites the
 * reference file:line (relative to /root/reference) it restates.  Written from the reference's
 * behaviour; no reference source text is reproduced here.
 */
#define _POSIX_C_SOURCE 200809L
#include "vv_oracle.h"

#include <dlfcn.h>
#include <time.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ============================================================ glibc random() TYPE_3 ========= */
/* glibc 2.35 stdlib/random_r.c: __srandom_r / __random_r with rand_type TYPE_3 (deg 31, sep 3),
 * which is what rand() uses (stdlib/rand.c -> __random()).  Third-party dependency of the
 * reference, pinned in tests against the real libc of this image. */
void orc_srand(orc_rng* g, unsigned seed) {
  int32_t word = (int32_t)(seed ? seed : 1u);
  g->r[0] = word;
  for (int i = 1; i < 31; ++i) {
    long hi = word / 127773, lo = word % 127773;
    word = (int32_t)(16807 * lo - 2836 * hi);
    if (word < 0) word += 2147483647;
    g->r[i] = word;
  }
  g->f = 3;
  g->b = 0;
  for (int i = 0; i < 310; ++i) (void)orc_rand(g);
}

int32_t orc_rand(orc_rng* g) {
  uint32_t val = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
  g->r[g->f] = (int32_t)val;
  if (++g->f >= 31) { g->f = 0; ++g->b; }
  else if (++g->b >= 31) g->b = 0;
  return (int32_t)(val >> 1);
}

/* include/caffe/util/rng.hpp:43-54 */
void orc_random_unique(orc_rng* g, int32_t* a, int len, int n) {
  int left = len, first = 0;
  while (n--) {
    int r = first + orc_rand(g) % left;
    int32_t t = a[first]; a[first] = a[r]; a[r] = t;
    ++first; --left;
  }
}

/* libstdc++ std::random_shuffle(first, last): for i in [1, len): j = rand() % (i + 1); swap if
 * different.  Call site: video_sampled_shots_data_layer.cpp:482 */
void orc_random_shuffle(orc_rng* g, int32_t* a, int len) {
  for (int i = 1; i < len; ++i) {
    int j = orc_rand(g) % (i + 1);
    if (i != j) { int32_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
}

/* ============================================================ sampler ======================= */
struct orc_sampler {
  orc_dataset ds;
  orc_sampler_param p;
  orc_rng rng;
  int64_t rand_calls;
  int32_t cursor;
  /* negative ring buffer (negatives_, negative_id_to_key_, negative_keys_set_, buffer_ids_) */
  int32_t* buffer_ids;     /* persistent permutation, …data_layer.cpp:81-83 */
  int32_t* buf_row;        /* slot -> table row (stands for the copied feature vector) */
  uint64_t* buf_key;       /* slot -> "vid:shot" key */
  int32_t* chain_next;     /* hash chain over slots */
  int32_t* bucket;         /* bucket heads */
  int32_t n_bucket;
  /* persistent prefetch_data_ contents (needed for quirk Q1) */
  int32_t* slot_row;       /* [B][C+Nn] */
  int32_t* slot_last;      /* [B][C+Nn] */
  int32_t* perm;           /* scratch, max n_shots */
};

static int32_t s_rand(orc_sampler* s) { ++s->rand_calls; return orc_rand(&s->rng); }

static uint64_t make_key(int32_t vid, int32_t shot) {
  return ((uint64_t)(uint32_t)vid << 32) | (uint32_t)shot;
}
static uint32_t hash_key(uint64_t k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return (uint32_t)k;
}
static int key_find(const orc_sampler* s, uint64_t k) {
  for (int32_t i = s->bucket[hash_key(k) & (s->n_bucket - 1)]; i >= 0; i = s->chain_next[i])
    if (s->buf_key[i] == k) return 1;
  return 0;
}
static void key_insert(orc_sampler* s, int32_t slot, uint64_t k) {
  uint32_t h = hash_key(k) & (s->n_bucket - 1);
  s->buf_key[slot] = k;
  s->chain_next[slot] = s->bucket[h];
  s->bucket[h] = slot;
}
static void key_erase(orc_sampler* s, int32_t slot) {
  uint32_t h = hash_key(s->buf_key[slot]) & (s->n_bucket - 1);
  int32_t* pp = &s->bucket[h];
  while (*pp != slot) pp = &s->chain_next[*pp];
  *pp = s->chain_next[slot];
}
static int32_t shot_id_of(const orc_dataset* ds, int v, int j) {
  return ds->shot_ids ? ds->shot_ids[ds->shot_off[v] + j] : j;
}

static void s_random_unique(orc_sampler* s, int32_t* a, int len, int n) {
  int left = len, first = 0;
  while (n--) {
    int r = first + s_rand(s) % left;
    int32_t t = a[first]; a[first] = a[r]; a[r] = t;
    ++first; --left;
  }
}
static void s_random_shuffle(orc_sampler* s, int32_t* a, int len) {
  for (int i = 1; i < len; ++i) {
    int j = s_rand(s) % (i + 1);
    if (i != j) { int32_t t = a[i]; a[i] = a[j]; a[j] = t; }
  }
}

void orc_sampler_destroy(orc_sampler* s) {
  if (!s) return;
  free(s->buffer_ids); free(s->buf_row); free(s->buf_key); free(s->chain_next); free(s->bucket);
  free(s->slot_row); free(s->slot_last); free(s->perm); free(s);
}

/* video_sampled_shots_data_layer.cpp:64-369 (DataLayerSetUp) */
orc_sampler* orc_sampler_create(const orc_dataset* ds, const orc_sampler_param* p, unsigned seed) {
  return orc_sampler_create_neg(ds, NULL, p, seed);
}
orc_sampler* orc_sampler_create_neg(const orc_dataset* ds, const orc_dataset* neg, const orc_sampler_param* p_in,
                                    unsigned seed) {
  orc_sampler_param pp = *p_in;
  if (pp.context_type == ORC_CONTEXT_PAIRWISE) pp.context_size = 2;                  /* :200-201 */
  const orc_sampler_param* p = &pp;
  if (p->context_size < 2 || p->batch_size < 1 || ds->n_videos < 1) return NULL;   /* :207,:209 */
  if (p->context_type < ORC_CONTEXT_WINDOW || p->context_type > ORC_CONTEXT_PAIRWISE) return NULL;  /* :760 */
  if (neg && neg->n_videos < 1) return NULL;
  if (p->num_negative_samples > 0 &&
      (p->negative_swap_percentage < 0 || p->negative_swap_percentage > 99)) return NULL; /* :79-80 */
  /* :484-502 write same-video negatives at channel C + added without comparing added with num_negative_samples: more of
   * them than negative slots runs into the next item's channels (undefined behaviour in the reference); refused here
   * exactly as the product refuses it. */
  if (p->max_same_video_negs > p->num_negative_samples) return NULL;
  orc_sampler* s = (orc_sampler*)calloc(1, sizeof(*s));
  s->ds = *ds; s->p = *p;
  orc_srand(&s->rng, seed);
  s->cursor = p->initial_cursor % ds->n_videos;                                        /* :156-180 */
  const int CN = p->context_size + p->num_negative_samples;
  const int mb = p->num_negative_samples > 0 ? p->max_buffer_size : 0;
  int max_n = 1;
  for (int v = 0; v < ds->n_videos; ++v) if (ds->n_shots[v] > max_n) max_n = ds->n_shots[v];
  s->perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)max_n);
  s->slot_row = (int32_t*)malloc(sizeof(int32_t) * (size_t)p->batch_size * CN);
  s->slot_last = (int32_t*)malloc(sizeof(int32_t) * (size_t)p->batch_size * CN);
  for (int i = 0; i < p->batch_size * CN; ++i) s->slot_row[i] = s->slot_last[i] = -1;
  s->n_bucket = 16;
  while (s->n_bucket < 2 * mb) s->n_bucket <<= 1;
  s->bucket = (int32_t*)malloc(sizeof(int32_t) * (size_t)s->n_bucket);
  for (int i = 0; i < s->n_bucket; ++i) s->bucket[i] = -1;
  s->buffer_ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mb + 1));
  s->buf_row = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mb + 1));
  s->buf_key = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(mb + 1));
  s->chain_next = (int32_t*)malloc(sizeof(int32_t) * (size_t)(mb + 1));
  for (int i = 0; i < mb; ++i) s->buffer_ids[i] = i;                                   /* :81-83 */

  /* :240-344 -- one rand()%num_shots per visited record until the buffer holds max_buffer_size
   * unique keys; the cursor advances on every visit and is NOT rewound afterwards. */
  if (mb > 0 && neg) {
    /* :253-286 the negative dataset's own cursor, :325-341 every shot whose key is new.  The reference tests for a
     * full buffer only after a whole record (:343) and writes past negatives_ when a record overshoots; only the
     * exact fit passes its CHECK_EQ (:348), anything else is refused here. */
    int added = 0, cur = 0;
    const int64_t tries = (int64_t)p->max_tries_for_negs * mb;
    for (int64_t nid = 0; nid < tries && added < mb; ++nid) {
      const int v = cur;
      cur = (cur + 1) % neg->n_videos;
      for (int j = 0; j < neg->n_shots[v]; ++j) {
        const uint64_t k = make_key(neg->video_id[v], shot_id_of(neg, v, j));
        if (key_find(s, k)) continue;
        if (added >= mb) { orc_sampler_destroy(s); return NULL; }
        s->buf_row[added] = (int32_t)(neg->row_base[v] + j);
        key_insert(s, added, k);
        ++added;
      }
    }
    if (added != mb) { orc_sampler_destroy(s); return NULL; }
  } else if (mb > 0) {
    int added = 0;
    const int64_t tries = (int64_t)p->max_tries_for_negs * mb;
    for (int64_t nid = 0; nid < tries; ++nid) {
      const int v = s->cursor;
      s->cursor = (s->cursor + 1) % ds->n_videos;
      const int n = ds->n_shots[v];
      const int j = s_rand(s) % n;                                                     /* :306 */
      const uint64_t k = make_key(ds->video_id[v], shot_id_of(ds, v, j));
      if (!key_find(s, k)) {
        s->buf_row[added] = (int32_t)(ds->row_base[v] + j);
        key_insert(s, added, k);
        ++added;
      }
      if (added >= mb) break;                                                          /* :338 */
    }
    if (added != mb) { orc_sampler_destroy(s); return NULL; }                          /* :344 */
  }
  return s;
}

/* video_sampled_shots_data_layer.cpp:371-393,425-507 (AddSamplesToTop, CONTEXT_WINDOW) */
static int add_samples_window(orc_sampler* s, int v, int item, int* added_negs) {
  const orc_sampler_param* p = &s->p;
  const int C = p->context_size, CN = C + p->num_negative_samples;
  const int n = s->ds.n_shots[v];
  const int64_t base = s->ds.row_base[v];
  *added_negs = 0;
  if (n < 2) return 0;                                                                 /* :387 */
  if (n < C) return 0;                                                                 /* :427 */
  int32_t* perm = s->perm;
  for (int i = 0; i < n; ++i) perm[i] = i;                                             /* :391 */
  s_random_unique(s, perm, n, C);                                                      /* :432 */
  /* :437 std::sort of the first C ids (distinct ints) */
  for (int i = 1; i < C; ++i) {
    int32_t x = perm[i]; int j = i - 1;
    while (j >= 0 && perm[j] > x) { perm[j + 1] = perm[j]; --j; }
    perm[j + 1] = x;
  }
  const int half = C / 2;
  int ctx = 0;
  int32_t* row = s->slot_row + (size_t)item * CN;
  int32_t* last = s->slot_last + (size_t)item * CN;
  for (int i = 0; i < C; ++i) {                                                        /* :439-453 */
    const int32_t r = (int32_t)(base + perm[i]);
    if (i == half) { row[0] = r; last[0] = r; }
    else { row[ctx + 1] = r; last[ctx + 1] = r; ++ctx; }
  }
  if (p->num_negative_samples > 0 && n > C) {                                          /* :479-503 */
    s_random_shuffle(s, perm + C, n - C);                                              /* :482 */
    for (int nid = C; nid < n && *added_negs < p->max_same_video_negs; ++nid) {
      if (perm[nid] < perm[half - 1] || perm[nid] > perm[half + 1]) {                  /* :489-490 */
        /* :492 copies only datum_height_-1 values: the last feature keeps the slot's old one */
        row[C + *added_negs] = (int32_t)(base + perm[nid]);
        ++*added_negs;
      }
    }
  }
  return 1;
}

/* video_sampled_shots_data_layer.cpp:510-596 (CONTEXT_PAST), 599-674 (PAST_CONTINUOUS), 677-757 (PAST_CONTINUOUS_FIXED):
 * the target is the LAST of the C chosen frames, the context the C-1 before it, in time order. */
static int add_samples_past(orc_sampler* s, int v, int item, int* added_negs) {
  const orc_sampler_param* p = &s->p;
  const int C = p->context_size, CN = C + p->num_negative_samples;
  const int n = s->ds.n_shots[v];
  const int64_t base = s->ds.row_base[v];
  *added_negs = 0;
  if (n < 2) return 0;                                                                 /* :387 */
  if (n < C) return 0;                                                                 /* :512, :601, :679 */
  int32_t* perm = s->perm;
  for (int i = 0; i < n; ++i) perm[i] = i;                                             /* :391 */
  int32_t* row = s->slot_row + (size_t)item * CN;
  int32_t* last = s->slot_last + (size_t)item * CN;
  int begin_frame = 0, sample_length = 0;
  if (p->context_type == ORC_CONTEXT_PAST) {
    s_random_unique(s, perm, n, C);                                                    /* :517 */
    for (int i = 1; i < C; ++i) {                                                      /* :522 std::sort */
      int32_t x = perm[i]; int j = i - 1;
      while (j >= 0 && perm[j] > x) { perm[j + 1] = perm[j]; --j; }
      perm[j + 1] = x;
    }
  } else {
    const int max_sample_length = (n - C) / (C - 1);                                   /* :609, :687 */
    if (p->context_type == ORC_CONTEXT_PAST_CONTINUOUS) {
      sample_length = s_rand(s) % (max_sample_length + 1);                             /* :610 */
      begin_frame = s_rand(s) % (n - (C - 1) * sample_length - C + 1);                 /* :612-613 */
    } else {
      sample_length = max_sample_length >= 1 ? max_sample_length - 1 : 0;              /* :688 */
      begin_frame = n - (C - 1) * sample_length - C;                                   /* :690-691 */
    }
  }
  int ctx = 0;
  for (int i = 0; i < C; ++i) {                                                        /* :525-541, :617-632, :695-710 */
    const int frame = p->context_type == ORC_CONTEXT_PAST ? perm[i] : begin_frame + i * (sample_length + 1);
    const int32_t r = (int32_t)(base + frame);
    if (i == C - 1) { row[0] = r; last[0] = r; }
    else { row[ctx + 1] = r; last[ctx + 1] = r; ++ctx; }
  }
  if (p->context_type == ORC_CONTEXT_PAST) {
    if (p->num_negative_samples > 0 && n > C) {                                        /* :563-583 */
      s_random_shuffle(s, perm + C, n - C);                                            /* :566 */
      for (int nid = C; nid < n && *added_negs < p->max_same_video_negs; ++nid)
        if (perm[nid] < perm[1]) {                                                     /* :570 */
          row[C + *added_negs] = (int32_t)(base + perm[nid]);                          /* :572-577: F-1 values */
          ++*added_negs;
        }
    }
  } else if (p->num_negative_samples > 0 && begin_frame > 0) {                         /* :652-670, :730-748 */
    for (int nid = begin_frame - 1; nid >= 0 && *added_negs < p->max_same_video_negs; --nid) {
      row[C + *added_negs] = (int32_t)(base + nid);
      ++*added_negs;
    }
  }
  return 1;
}

/* video_sampled_shots_data_layer.cpp:396-422 (CONTEXT_PAIRWISE): two distinct random frames in draw order; the label
 * is the video id, or with output_shot_distance their distance clamped to max_shot_distance (the reference
 * holds it in an int: the float bound is truncated, :412-415). */
static int add_samples_pairwise(orc_sampler* s, int v, int item, int32_t* label) {
  const orc_sampler_param* p = &s->p;
  const int CN = 2 + p->num_negative_samples;
  const int n = s->ds.n_shots[v];
  const int64_t base = s->ds.row_base[v];
  if (n < 2) return 0;                                                                 /* :387 */
  int32_t* perm = s->perm;
  for (int i = 0; i < n; ++i) perm[i] = i;                                             /* :391 */
  s_random_unique(s, perm, n, 2);                                                      /* :397 */
  int32_t* row = s->slot_row + (size_t)item * CN;
  int32_t* last = s->slot_last + (size_t)item * CN;
  row[0] = last[0] = (int32_t)(base + perm[0]);                                        /* :400-405 */
  row[1] = last[1] = (int32_t)(base + perm[1]);
  if (p->output_shot_distance) {
    const int d = abs(perm[0] - perm[1]);
    *label = (float)d >= p->max_shot_distance ? (int32_t)p->max_shot_distance : d;
  } else {
    *label = s->ds.video_id[v];
  }
  return 1;
}

/* video_sampled_shots_data_layer.cpp:768-909 (InternalThreadEntry) */
void orc_sampler_next(orc_sampler* s, int32_t* idx, int32_t* last_src, int32_t* label) {
  const orc_sampler_param* p = &s->p;
  const int C = p->context_size, Nn = p->num_negative_samples, CN = C + Nn;
  int item = 0;
  while (item < p->batch_size) {
    const int v = s->cursor;
    int added = 0;
    int32_t lab = s->ds.video_id[v];
    const int ok = p->context_type == ORC_CONTEXT_WINDOW ? add_samples_window(s, v, item, &added)   /* :820 */
                 : p->context_type == ORC_CONTEXT_PAIRWISE ? add_samples_pairwise(s, v, item, &lab)
                                                           : add_samples_past(s, v, item, &added);
    s->cursor = (s->cursor + 1) % s->ds.n_videos;                                      /* :826-846 */
    if (!ok) continue;                                                                 /* :848 */
    if (Nn > 0) {
      s_random_unique(s, s->buffer_ids, p->max_buffer_size, Nn - added);               /* :855 */
      for (int c = C + added; c < CN; ++c) {                                           /* :856-875 */
        const int32_t neg = s->buffer_ids[c - C - added];
        s->slot_row[(size_t)item * CN + c] = s->buf_row[neg];
        s->slot_last[(size_t)item * CN + c] = s->buf_row[neg];
      }
    }
    if (label) label[item] = lab;                                                      /* :879 */
    ++item;
    if (Nn > 0 && p->negative_swap_percentage > 0) {                                   /* :888-906 */
      const int n = s->ds.n_shots[v];
      for (int j = 0; j < n; ++j) {
        const uint64_t k = make_key(s->ds.video_id[v], shot_id_of(&s->ds, v, j));
        if (key_find(s, k)) continue;
        if (s_rand(s) % 100 < p->negative_swap_percentage) {                           /* :27 */
          const int pos = s_rand(s) % p->max_buffer_size;                              /* :29 */
          key_erase(s, pos);
          s->buf_row[pos] = (int32_t)(s->ds.row_base[v] + j);
          key_insert(s, pos, k);
        }
      }
    }
  }
  const size_t nb = sizeof(int32_t) * (size_t)p->batch_size * CN;
  if (idx) memcpy(idx, s->slot_row, nb);
  if (last_src) memcpy(last_src, s->slot_last, nb);
}

const int32_t* orc_sampler_buffer_rows(const orc_sampler* s) { return s->buf_row; }
const int32_t* orc_sampler_buffer_ids(const orc_sampler* s) { return s->buffer_ids; }
int32_t orc_sampler_cursor(const orc_sampler* s) { return s->cursor; }
int64_t orc_sampler_rand_calls(const orc_sampler* s) { return s->rand_calls; }

/* ============================================================ sgemm ========================= */
static int g_threads = 0;
void orc_set_threads(int n) {
#ifdef _OPENMP
  static int default_threads = 0;
  if (!default_threads) default_threads = omp_get_max_threads();
  omp_set_num_threads(n > 0 ? n : default_threads);       /* an external BLAS on the GNU OpenMP runtime follows this */
#endif
  g_threads = n;
}
int orc_get_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

/* Blocked, packed sgemm in the GotoBLAS / BLIS arrangement -- what an OpenBLAS or MKL build of the reference would
 * run for caffe_cpu_gemm (math_functions.cpp:12-21), self-contained so that the CPU baseline does not depend on what
 * BLAS the box happens to ship:
 *   for jc (NC columns) / pc (KC deep): pack B[pc.., jc..] once into NR-wide column panels (all threads);
 *     tasks = (MC-row block of A) x (chunk of B panels): pack the A block into MR-tall row panels (thread-private),
 *     then MR x NR register-blocked micro-kernels, C accumulated in place.
 * Micro-kernels: AVX-512 (8 x 32 in 16 zmm accumulators) when the CPU has it, else AVX2 + FMA (6 x 16 in 12 ymm),
 * chosen at run time. */
#include <immintrin.h>

/* Threads worth waking for `work` units (multiply-adds, or floats moved): a parallel region over hundreds of threads costs
 * far more than a small problem -- the fixture-sized cases of the tests ran 10^4 times slower on a 256-thread host than on
 * one thread.  One thread per 64K units, at most what orc_set_threads allows. */
static int threads_for(double work) {
  const int nt = orc_get_threads();
  const double want = work / 65536.0;
  if (want <= 1.0) return 1;
  return want >= nt ? nt : (int)want;
}

#define GK_KC 384
#define GK_MC 96             /* multiple of both micro-kernels' MR (8 and 6) */
#define GK_NC 4096

typedef void (*ukr_fn)(int kc, const float* ap, const float* bp, float* c, int ldc, int first);

/* 8 x 32: ap = kc steps of 8 values, bp = kc steps of 32 values; c += (first ? 0 : c) + ap^T bp */
__attribute__((target("avx512f"))) static void ukr_avx512(int kc, const float* ap, const float* bp, float* c, int ldc, int first) {
  __m512 acc[8][2];
  for (int i = 0; i < 8; ++i) { acc[i][0] = _mm512_setzero_ps(); acc[i][1] = _mm512_setzero_ps(); }
  for (int k = 0; k < kc; ++k) {
    const __m512 b0 = _mm512_loadu_ps(bp), b1 = _mm512_loadu_ps(bp + 16);
#pragma GCC unroll 8
    for (int i = 0; i < 8; ++i) {
      const __m512 a = _mm512_set1_ps(ap[i]);
      acc[i][0] = _mm512_fmadd_ps(a, b0, acc[i][0]);
      acc[i][1] = _mm512_fmadd_ps(a, b1, acc[i][1]);
    }
    ap += 8; bp += 32;
  }
  for (int i = 0; i < 8; ++i) {
    float* ci = c + (size_t)i * ldc;
    if (first) { _mm512_storeu_ps(ci, acc[i][0]); _mm512_storeu_ps(ci + 16, acc[i][1]); }
    else {
      _mm512_storeu_ps(ci, _mm512_add_ps(_mm512_loadu_ps(ci), acc[i][0]));
      _mm512_storeu_ps(ci + 16, _mm512_add_ps(_mm512_loadu_ps(ci + 16), acc[i][1]));
    }
  }
}
/* 6 x 16 */
__attribute__((target("avx2,fma"))) static void ukr_avx2(int kc, const float* ap, const float* bp, float* c, int ldc, int first) {
  __m256 acc[6][2];
  for (int i = 0; i < 6; ++i) { acc[i][0] = _mm256_setzero_ps(); acc[i][1] = _mm256_setzero_ps(); }
  for (int k = 0; k < kc; ++k) {
    const __m256 b0 = _mm256_loadu_ps(bp), b1 = _mm256_loadu_ps(bp + 8);
#pragma GCC unroll 6
    for (int i = 0; i < 6; ++i) {
      const __m256 a = _mm256_broadcast_ss(ap + i);
      acc[i][0] = _mm256_fmadd_ps(a, b0, acc[i][0]);
      acc[i][1] = _mm256_fmadd_ps(a, b1, acc[i][1]);
    }
    ap += 6; bp += 16;
  }
  for (int i = 0; i < 6; ++i) {
    float* ci = c + (size_t)i * ldc;
    if (first) { _mm256_storeu_ps(ci, acc[i][0]); _mm256_storeu_ps(ci + 8, acc[i][1]); }
    else {
      _mm256_storeu_ps(ci, _mm256_add_ps(_mm256_loadu_ps(ci), acc[i][0]));
      _mm256_storeu_ps(ci + 8, _mm256_add_ps(_mm256_loadu_ps(ci + 8), acc[i][1]));
    }
  }
}

static int g_force_isa = 0;                                      /* tests: 1 = AVX2 kernel even on an AVX-512 machine */
void orc_sgemm_force_isa(int v) { g_force_isa = v; }
const char* orc_sgemm_isa(void) {
  return (!g_force_isa && __builtin_cpu_supports("avx512f")) ? "avx512f 8x32" : "avx2+fma 6x16";
}

/* op(A)[M][K] (a_trans: stored [K][M]) times op(B)[K][N] (b_trans: stored [N][K]) -> C[M][N] row-major, ldc = N.
 * C = alpha * op(A) op(B) + beta * C. */
static void gemm_blocked(int M, int N, int K, float alpha, const float* A, int a_trans, const float* B, int b_trans,
                         float beta, float* C) {
  const int nt = threads_for((double)M * N * K / 16.0);
  const int use512 = !g_force_isa && __builtin_cpu_supports("avx512f");
  const int MR = use512 ? 8 : 6, NR = use512 ? 32 : 16;
  const ukr_fn ukr = use512 ? ukr_avx512 : ukr_avx2;
  /* beta first, then plain accumulation (alpha is folded into the packed A) */
  if (beta == 0.f) {
#pragma omp parallel for num_threads(nt)
    for (int i = 0; i < M; ++i) memset(C + (size_t)i * N, 0, sizeof(float) * (size_t)N);
  } else if (beta != 1.f) {
#pragma omp parallel for num_threads(nt)
    for (int i = 0; i < M; ++i) for (int j = 0; j < N; ++j) C[(size_t)i * N + j] *= beta;
  }
  const int mblocks = (M + GK_MC - 1) / GK_MC;
  if (mblocks * 2 <= nt) {
    /* Few row blocks (the weight gradient: M = D): the packed slab of A (all M rows x KC) is packed once per K block
     * and shared; every thread packs column panels of B for itself and runs them against all of A.  Same products,
     * same K order per element of C as the general arrangement below. */
    const int mpan = (M + MR - 1) / MR, npan_all = (N + NR - 1) / NR;
    float* Ash = (float*)aligned_alloc(64, sizeof(float) * (size_t)GK_KC * mpan * MR + 64);
#pragma omp parallel num_threads(nt)
    {
      float* Bq = (float*)aligned_alloc(64, sizeof(float) * (size_t)GK_KC * NR + 64);
      float tile[8 * 32];
      for (int pc = 0; pc < K; pc += GK_KC) {
        const int kc = K - pc < GK_KC ? K - pc : GK_KC;
#pragma omp for schedule(static)
        for (int ip = 0; ip < mpan; ++ip) {
          float* dst = Ash + (size_t)ip * GK_KC * MR;
          const int r0 = ip * MR, mr = M - r0 < MR ? M - r0 : MR;
          if (!a_trans) {
            for (int i = 0; i < mr; ++i) {
              const float* src = A + (size_t)(r0 + i) * K + pc;
              for (int k = 0; k < kc; ++k) dst[(size_t)k * MR + i] = alpha * src[k];
            }
            for (int i = mr; i < MR; ++i) for (int k = 0; k < kc; ++k) dst[(size_t)k * MR + i] = 0.f;
          } else {
            for (int k = 0; k < kc; ++k) {
              const float* src = A + (size_t)(pc + k) * M + r0;
              for (int i = 0; i < mr; ++i) dst[(size_t)k * MR + i] = alpha * src[i];
              for (int i = mr; i < MR; ++i) dst[(size_t)k * MR + i] = 0.f;
            }
          }
        }                                                         /* implicit barrier */
#pragma omp for schedule(dynamic, 1)
        for (int jp = 0; jp < npan_all; ++jp) {
          const int j0 = jp * NR, nr = N - j0 < NR ? N - j0 : NR;
          if (!b_trans) {
            for (int k = 0; k < kc; ++k) {
              const float* src = B + (size_t)(pc + k) * N + j0;
              for (int j = 0; j < nr; ++j) Bq[(size_t)k * NR + j] = src[j];
              for (int j = nr; j < NR; ++j) Bq[(size_t)k * NR + j] = 0.f;
            }
          } else {
            for (int j = 0; j < nr; ++j) {
              const float* src = B + (size_t)(j0 + j) * K + pc;
              for (int k = 0; k < kc; ++k) Bq[(size_t)k * NR + j] = src[k];
            }
            for (int j = nr; j < NR; ++j) for (int k = 0; k < kc; ++k) Bq[(size_t)k * NR + j] = 0.f;
          }
          for (int ip = 0; ip < mpan; ++ip) {
            const int r0 = ip * MR, mr = M - r0 < MR ? M - r0 : MR;
            const float* ap = Ash + (size_t)ip * GK_KC * MR;
            if (mr == MR && nr == NR) ukr(kc, ap, Bq, C + (size_t)r0 * N + j0, N, 0);
            else {
              ukr(kc, ap, Bq, tile, NR, 1);
              for (int i = 0; i < mr; ++i) for (int j = 0; j < nr; ++j) C[(size_t)(r0 + i) * N + j0 + j] += tile[i * NR + j];
            }
          }
        }                                                         /* implicit barrier: Ash is repacked next */
      }
      free(Bq);
    }
    free(Ash);
    return;
  }
  const int nc_max = N < GK_NC ? N : GK_NC;
  const int npan_max = (nc_max + NR - 1) / NR;
  float* Bp = (float*)aligned_alloc(64, sizeof(float) * (size_t)GK_KC * npan_max * NR + 64);
#pragma omp parallel num_threads(nt)
  {
    float* Ap = (float*)aligned_alloc(64, sizeof(float) * (size_t)GK_KC * (GK_MC + 8) + 64);
    float tile[8 * 32];
    for (int jc = 0; jc < N; jc += GK_NC) {
      const int nc = N - jc < GK_NC ? N - jc : GK_NC;
      const int npan = (nc + NR - 1) / NR;
      /* enough tasks for every thread even when M is one or two blocks (the weight gradient: M = D) */
      int chunks = (4 * nt + mblocks - 1) / mblocks;
      if (chunks > npan) chunks = npan;
      if (chunks < 1) chunks = 1;
      const int pan_per = (npan + chunks - 1) / chunks;
      for (int pc = 0; pc < K; pc += GK_KC) {
        const int kc = K - pc < GK_KC ? K - pc : GK_KC;
#pragma omp for schedule(static)
        for (int jp = 0; jp < npan; ++jp) {                       /* pack B: panel jp = kc x NR, k-major */
          float* dst = Bp + (size_t)jp * GK_KC * NR;
          const int j0 = jc + jp * NR, nr = N - j0 < NR ? N - j0 : NR;
          if (!b_trans) {
            for (int k = 0; k < kc; ++k) {
              const float* src = B + (size_t)(pc + k) * N + j0;
              for (int j = 0; j < nr; ++j) dst[(size_t)k * NR + j] = src[j];
              for (int j = nr; j < NR; ++j) dst[(size_t)k * NR + j] = 0.f;
            }
          } else {
            for (int j = 0; j < nr; ++j) {
              const float* src = B + (size_t)(j0 + j) * K + pc;
              for (int k = 0; k < kc; ++k) dst[(size_t)k * NR + j] = src[k];
            }
            for (int j = nr; j < NR; ++j) for (int k = 0; k < kc; ++k) dst[(size_t)k * NR + j] = 0.f;
          }
        }                                                         /* implicit barrier */
#pragma omp for schedule(dynamic, 1)
        for (int task = 0; task < mblocks * chunks; ++task) {
          const int ib = task / chunks, ch = task % chunks;
          const int i0 = ib * GK_MC, mc = M - i0 < GK_MC ? M - i0 : GK_MC;
          const int mpan = (mc + MR - 1) / MR;
          for (int ip = 0; ip < mpan; ++ip) {                     /* pack A: panel ip = kc x MR, k-major, times alpha */
            float* dst = Ap + (size_t)ip * GK_KC * MR;
            const int r0 = i0 + ip * MR, mr = M - r0 < MR ? M - r0 : MR;
            if (!a_trans) {
              for (int i = 0; i < mr; ++i) {
                const float* src = A + (size_t)(r0 + i) * K + pc;
                for (int k = 0; k < kc; ++k) dst[(size_t)k * MR + i] = alpha * src[k];
              }
              for (int i = mr; i < MR; ++i) for (int k = 0; k < kc; ++k) dst[(size_t)k * MR + i] = 0.f;
            } else {
              for (int k = 0; k < kc; ++k) {
                const float* src = A + (size_t)(pc + k) * M + r0;
                for (int i = 0; i < mr; ++i) dst[(size_t)k * MR + i] = alpha * src[i];
                for (int i = mr; i < MR; ++i) dst[(size_t)k * MR + i] = 0.f;
              }
            }
          }
          const int jp0 = ch * pan_per, jp1 = jp0 + pan_per < npan ? jp0 + pan_per : npan;
          for (int jp = jp0; jp < jp1; ++jp) {
            const int j0 = jc + jp * NR, nr = N - j0 < NR ? N - j0 : NR;
            const float* bp = Bp + (size_t)jp * GK_KC * NR;
            for (int ip = 0; ip < mpan; ++ip) {
              const int r0 = i0 + ip * MR, mr = M - r0 < MR ? M - r0 : MR;
              const float* ap = Ap + (size_t)ip * GK_KC * MR;
              if (mr == MR && nr == NR) ukr(kc, ap, bp, C + (size_t)r0 * N + j0, N, 0);
              else {                                              /* edge: through a full-size scratch tile */
                ukr(kc, ap, bp, tile, NR, 1);
                for (int i = 0; i < mr; ++i) for (int j = 0; j < nr; ++j) C[(size_t)(r0 + i) * N + j0 + j] += tile[i * NR + j];
              }
            }
          }
        }                                                         /* implicit barrier: Bp is repacked next */
      }
    }
    free(Ap);
  }
  free(Bp);
}

/* The reference delegates its GEMM to an external BLAS (cblas_sgemm, math_functions.cpp:12-21; which library is
 * a build choice of the user, Makefile.config:34).  orc_set_blas loads one at run time (a shared object exporting the
 * standard cblas_sgemm, e.g. MKL's libmkl_rt.so or OpenBLAS) so that the CPU baseline can be timed the way a reference
 * build would run; without it the self-contained OpenMP kernel below is used. */
typedef void (*cblas_sgemm_fn)(int order, int transA, int transB, int M, int N, int K, float alpha, const float* A, int lda,
                               const float* B, int ldb, float beta, float* C, int ldc);
static cblas_sgemm_fn g_cblas_sgemm = NULL;
int orc_set_blas(const char* path) {
  g_cblas_sgemm = NULL;
  if (!path || !*path) return 0;
  void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
  if (!h) return -1;
  g_cblas_sgemm = (cblas_sgemm_fn)dlsym(h, "cblas_sgemm");
  return g_cblas_sgemm ? 0 : -2;
}
int orc_has_blas(void) { return g_cblas_sgemm != NULL; }

/* caffe_cpu_gemm (math_functions.cpp:12-21): row-major, lda/ldb derived from the trans flags. */
void orc_sgemm(int transA, int transB, int M, int N, int K, float alpha, const float* A,
               const float* B, float beta, float* C) {
  if (g_cblas_sgemm) {          /* CblasRowMajor = 101, CblasNoTrans = 111, CblasTrans = 112 (math_functions.cpp:16-20) */
    g_cblas_sgemm(101, transA ? 112 : 111, transB ? 112 : 111, M, N, K, alpha, A, transA ? M : K, B, transB ? K : N, beta, C, N);
    return;
  }
  gemm_blocked(M, N, K, alpha, A, transA, B, transB, beta, C);
}

/* ============================================================ layers ======================== */
/* normalization_layer.cpp:29-61.  caffe_powx -> powf (mkl_alternate.hpp:55). */
void orc_normalize_fwd(int num, int dim, const float* x, float* y) {
  const float eps = 1e-10f;
#pragma omp parallel for num_threads(threads_for((double)num * dim))
  for (int i = 0; i < num; ++i) {
    const float* xi = x + (size_t)i * dim;
    float s = 0.f;
    for (int j = 0; j < dim; ++j) s += powf(xi[j], 2.f);                              /* :39-44 */
    const float nrm = powf(s, 0.5f) + eps;                                            /* :47-51 */
    for (int j = 0; j < dim; ++j) y[(size_t)i * dim + j] = xi[j] / nrm;               /* :54-59 */
  }
}

/* normalization_layer.cpp:63-112: dx = (s*dy - x*(x.dy)) / (s^1.5 + eps), s = sum x^2 */
void orc_normalize_bwd(int num, int dim, const float* x, const float* dy, float* dx) {
  const float eps = 1e-10f;
#pragma omp parallel for num_threads(threads_for((double)num * dim))
  for (int i = 0; i < num; ++i) {
    const float* xi = x + (size_t)i * dim;
    const float* gi = dy + (size_t)i * dim;
    float dot = 0.f, s = 0.f;
    for (int j = 0; j < dim; ++j) dot += xi[j] * gi[j];                               /* :77-79 */
    for (int j = 0; j < dim; ++j) s += powf(xi[j], 2.f);                              /* :88-93 */
    const float den = powf(s, 1.5f) + eps;                                            /* :103-107 */
    for (int j = 0; j < dim; ++j)
      dx[(size_t)i * dim + j] = (s * gi[j] - xi[j] * dot) / den;                      /* :85,97-110 */
  }
}

/* max_margin_loss_layer.cpp:53-127 */
void orc_max_margin_fwd(int count, const float* s_true, const float* s_bogus, const float* weight,
                        float margin, int norm, float* loss, float* violations) {
  float nv = 0.f;
  double acc = 0.0;
  for (int i = 0; i < count; ++i) {
    const float d = s_true[i] - s_bogus[i];                                           /* :69 */
    if (d < 0) nv += 1.f;                                                             /* :78-80 */
    float h = fmaxf(0.f, margin - d);                                                 /* :99 */
    if (weight) h *= (norm == 2) ? sqrtf(weight[i]) : weight[i];                      /* :84-90 */
    acc += (norm == 2) ? (double)h * h : fabs((double)h);                             /* :112-118 */
  }
  if (loss) *loss = (float)(acc / count);
  if (violations) *violations = nv;                                                   /* :123-126 */
}

/* max_margin_loss_layer.cpp:129-214 */
void orc_max_margin_bwd(int count, const float* s_true, const float* s_bogus, const float* weight,
                        float margin, int norm, float loss_weight, float* d_true, float* d_bogus) {
  for (int i = 0; i < count; ++i) {
    float h = fmaxf(0.f, margin - (s_true[i] - s_bogus[i]));                          /* :149-161 */
    if (weight) h *= weight[i];                                                       /* :154 */
    float g;
    if (norm == 1) {                                                                  /* :173-189 */
      g = h > 0.f ? (weight ? weight[i] : 1.f) : h;
      g *= loss_weight / count;
    } else {
      g = h * (loss_weight * 2 / count);                                              /* :191 */
    }
    d_bogus[i] = g;
    d_true[i] = -g;                                                                   /* :211 */
  }
}

/* sum_layer.cpp:31-54 */
void orc_sum_fwd(int num, int dim, int num_output, const float* x, float* y) {
#pragma omp parallel for num_threads(threads_for((double)num * dim))
  for (int i = 0; i < num; ++i) {
    float s = 0.f;
    for (int j = 0; j < dim; ++j) s += x[(size_t)i * dim + j];
    for (int o = 0; o < num_output; ++o) y[(size_t)i * num_output + o] = s;
  }
}
/* sum_layer.cpp:56-82 */
void orc_sum_bwd(int num, int dim, int num_output, const float* dy, float* dx) {
#pragma omp parallel for num_threads(threads_for((double)num * dim))
  for (int i = 0; i < num; ++i) {
    float s = 0.f;
    for (int o = 0; o < num_output; ++o) s += dy[(size_t)i * num_output + o];
    for (int j = 0; j < dim; ++j) dx[(size_t)i * dim + j] = s;
  }
}

/* ============================================================ the plain layers of the graph === */
/* relu_layer.cpp:10-20 */
void orc_relu_fwd(int64_t n, const float* x, float slope, float* y) {
  for (int64_t i = 0; i < n; ++i) y[i] = fmaxf(x[i], 0.f) + slope * fminf(x[i], 0.f);
}
/* relu_layer.cpp:23-37: the gate is read from the BOTTOM data */
void orc_relu_bwd(int64_t n, const float* x, const float* dy, float slope, float* dx) {
  for (int64_t i = 0; i < n; ++i) dx[i] = dy[i] * ((x[i] > 0.f) + slope * (x[i] <= 0.f));
}
/* dropout_layer.cpp:34-50 (TRAIN: x * mask * 1/(1-ratio); TEST: copy).  The mask is an argument: the
 * reference draws it from its own Bernoulli stream (caffe_rng_bernoulli), which is not part of the path's contract. */
void orc_dropout_fwd(int64_t n, const float* x, const uint8_t* mask, float ratio, int train, float* y) {
  const float scale = 1.f / (1.f - ratio);
  if (train) for (int64_t i = 0; i < n; ++i) y[i] = x[i] * (float)mask[i] * scale;
  else for (int64_t i = 0; i < n; ++i) y[i] = x[i];
}
/* dropout_layer.cpp:52-68 */
void orc_dropout_bwd(int64_t n, const float* dy, const uint8_t* mask, float ratio, int train, float* dx) {
  const float scale = 1.f / (1.f - ratio);
  if (train) for (int64_t i = 0; i < n; ++i) dx[i] = dy[i] * (float)mask[i] * scale;
  else for (int64_t i = 0; i < n; ++i) dx[i] = dy[i];
}
/* eltwise_layer.cpp:53-105.  op: 0 PROD, 1 SUM (coeff may be NULL = all ones), 2 MAX */
void orc_eltwise_fwd(int op, int64_t n, int nb, const float* const* bottom, const float* coeff, float* top) {
  if (op == 0) {
    for (int64_t i = 0; i < n; ++i) top[i] = bottom[0][i] * bottom[1][i];
    for (int k = 2; k < nb; ++k) for (int64_t i = 0; i < n; ++i) top[i] = top[i] * bottom[k][i];
  } else if (op == 1) {
    for (int64_t i = 0; i < n; ++i) top[i] = 0.f;
    for (int k = 0; k < nb; ++k) {
      const float c = coeff ? coeff[k] : 1.f;
      for (int64_t i = 0; i < n; ++i) top[i] += c * bottom[k][i];                     /* caffe_axpy */
    }
  } else {
    for (int64_t i = 0; i < n; ++i) top[i] = bottom[0][i] > bottom[1][i] ? bottom[0][i] : bottom[1][i];
    for (int k = 2; k < nb; ++k) for (int64_t i = 0; i < n; ++i) if (bottom[k][i] > top[i]) top[i] = bottom[k][i];
  }
}
/* eltwise_layer.cpp:108-159: the diff of bottom `which`.  PROD: stable = product of the other bottoms
 * (stable_prod_grad, the default), otherwise top / bottom; MAX routes the diff to the arg-max bottom
 * (first maximum wins between bottoms 0 and 1 only when strictly greater, as the forward's mask does). */
void orc_eltwise_bwd(int op, int64_t n, int nb, const float* const* bottom, const float* coeff,
                     const float* top, const float* dtop, int which, int stable, float* dbottom) {
  if (op == 0) {
    if (stable) {
      int started = 0;
      for (int k = 0; k < nb; ++k) {
        if (k == which) continue;
        if (!started) { for (int64_t i = 0; i < n; ++i) dbottom[i] = bottom[k][i]; started = 1; }
        else for (int64_t i = 0; i < n; ++i) dbottom[i] = bottom[k][i] * dbottom[i];
      }
    } else {
      for (int64_t i = 0; i < n; ++i) dbottom[i] = top[i] / bottom[which][i];
    }
    for (int64_t i = 0; i < n; ++i) dbottom[i] = dbottom[i] * dtop[i];
  } else if (op == 1) {
    const float c = coeff ? coeff[which] : 1.f;
    if (c == 1.f) for (int64_t i = 0; i < n; ++i) dbottom[i] = dtop[i];
    else for (int64_t i = 0; i < n; ++i) dbottom[i] = c * dtop[i];
  } else {
    for (int64_t i = 0; i < n; ++i) {
      int arg = bottom[0][i] > bottom[1][i] ? 0 : 1;
      float best = bottom[arg][i];
      for (int k = 2; k < nb; ++k) if (bottom[k][i] > best) { best = bottom[k][i]; arg = k; }
      dbottom[i] = arg == which ? dtop[i] : 0.f;
    }
  }
}
/* SLICE forward / CONCAT backward (slice_layer.cpp:79-105, concat_layer.cpp:86-117): a blob cut along num
 * (outer = 1, inner = channels*height*width, width_k = num_k) or along channels (outer = num,
 * inner = height*width, width_k = channels_k) into pieces stored one after the other per outer index. */
void orc_split_pieces(int outer, int64_t inner, int npieces, const int32_t* width, const float* whole,
                      float* const* piece) {
  int64_t total = 0;
  for (int k = 0; k < npieces; ++k) total += width[k];
  int64_t at = 0;
  for (int k = 0; k < npieces; ++k) {
    const int64_t len = width[k] * inner;
    for (int o = 0; o < outer; ++o)
      memcpy(piece[k] + (size_t)o * len, whole + ((size_t)o * total + at) * inner, sizeof(float) * (size_t)len);
    at += width[k];
  }
}
/* CONCAT forward / SLICE backward (concat_layer.cpp:45-83, slice_layer.cpp:107-135) */
void orc_join_pieces(int outer, int64_t inner, int npieces, const int32_t* width, const float* const* piece,
                     float* whole) {
  int64_t total = 0;
  for (int k = 0; k < npieces; ++k) total += width[k];
  int64_t at = 0;
  for (int k = 0; k < npieces; ++k) {
    const int64_t len = width[k] * inner;
    for (int o = 0; o < outer; ++o)
      memcpy(whole + ((size_t)o * total + at) * inner, piece[k] + (size_t)o * len, sizeof(float) * (size_t)len);
    at += width[k];
  }
}
/* split_layer.cpp:36-51: the bottom diff is the sum of the top diffs (the forward shares the data, :28-34) */
void orc_split_bwd(int64_t n, int ntop, const float* const* dtop, float* dbottom) {
  for (int64_t i = 0; i < n; ++i) dbottom[i] = dtop[0][i];
  for (int k = 1; k < ntop; ++k) for (int64_t i = 0; i < n; ++i) dbottom[i] += dtop[k][i];
}
/* inner_product_layer.cpp:61-74: Y (M x N) = X (M x K) W^T (N x K) + 1 b^T */
void orc_inner_product_fwd(int M, int N, int K, const float* X, const float* W, const float* b, float* Y) {
  orc_sgemm(0, 1, M, N, K, 1.f, X, W, 0.f, Y);
  if (b) {
    const int nt = threads_for((double)M * N);
#pragma omp parallel for num_threads(nt)
    for (int r = 0; r < M; ++r) for (int d = 0; d < N; ++d) Y[(size_t)r * N + d] += b[d];
  }
}
/* inner_product_layer.cpp:76-106: dW = dY^T X, db = dY^T 1, dX = dY W (each optional) */
void orc_inner_product_bwd(int M, int N, int K, const float* X, const float* W, const float* dY,
                           float* dW, float* db, float* dX) {
  if (dW) orc_sgemm(1, 0, N, K, M, 1.f, dY, X, 0.f, dW);
  if (db) {
    for (int d = 0; d < N; ++d) db[d] = 0.f;
    for (int r = 0; r < M; ++r) for (int d = 0; d < N; ++d) db[d] += dY[(size_t)r * N + d];
  }
  if (dX) orc_sgemm(0, 0, M, K, N, 1.f, dY, W, 0.f, dX);
}

/* solver.cpp:440-460 */
float orc_learning_rate(const char* policy, float base_lr, float gamma, float power, int stepsize,
                        int iter) {
  if (!strcmp(policy, "fixed")) return base_lr;
  if (!strcmp(policy, "step")) return base_lr * powf(gamma, (float)(iter / stepsize));
  if (!strcmp(policy, "exp")) return base_lr * powf(gamma, (float)iter);
  if (!strcmp(policy, "inv")) return base_lr * powf(1.f + gamma * iter, -power);
  return NAN;
}

/* solver.cpp:502-531 (ComputeUpdateValue, CPU branch) then blob.cpp:112-136 (Blob::Update) */
void orc_sgd_update(int64_t n, float* w, float* grad, float* hist, float rate, float lr_mult,
                    float momentum, float weight_decay, float decay_mult, int reg) {
  const float local_rate = rate * lr_mult, local_decay = weight_decay * decay_mult;
  if (local_decay != 0.f) {
    if (reg == 2) for (int64_t i = 0; i < n; ++i) grad[i] += local_decay * w[i];
    else for (int64_t i = 0; i < n; ++i)
      grad[i] += local_decay * (float)((w[i] > 0.f) - (w[i] < 0.f));
  }
  for (int64_t i = 0; i < n; ++i) hist[i] = local_rate * grad[i] + momentum * hist[i];
  for (int64_t i = 0; i < n; ++i) grad[i] = hist[i];
  for (int64_t i = 0; i < n; ++i) w[i] -= grad[i];
}

/* NesterovSolver / AdaGradSolver::ComputeUpdateValue, CPU branches (solver.cpp:599-655, 714-781), then
 * Blob::Update.  solver_type: 0 SGD, 1 NESTEROV, 2 ADAGRAD (caffe.proto SolverParameter.SolverType). */
void orc_solver_update(int64_t n, float* w, float* grad, float* hist, float rate, float lr_mult,
                       float momentum, float weight_decay, float decay_mult, int reg, int solver_type,
                       float delta) {
  if (solver_type == 0) { orc_sgd_update(n, w, grad, hist, rate, lr_mult, momentum, weight_decay, decay_mult, reg); return; }
  const float local_rate = rate * lr_mult, local_decay = weight_decay * decay_mult;
  if (local_decay != 0.f) {
    if (reg == 2) for (int64_t i = 0; i < n; ++i) grad[i] += local_decay * w[i];
    else for (int64_t i = 0; i < n; ++i)
      grad[i] += local_decay * (float)((w[i] > 0.f) - (w[i] < 0.f));
  }
  float* update = (float*)malloc((size_t)(n ? n : 1) * sizeof(float));
  if (solver_type == 1) {
    for (int64_t i = 0; i < n; ++i) update[i] = hist[i];                               /* save history momentum */
    for (int64_t i = 0; i < n; ++i) hist[i] = local_rate * grad[i] + momentum * hist[i];
    for (int64_t i = 0; i < n; ++i) update[i] = (1.f + momentum) * hist[i] + -momentum * update[i];   /* step back then over step */
    for (int64_t i = 0; i < n; ++i) grad[i] = update[i];
  } else {
    for (int64_t i = 0; i < n; ++i) update[i] = powf(grad[i], 2.f);                    /* caffe_powx(diff, 2) */
    for (int64_t i = 0; i < n; ++i) hist[i] = update[i] + hist[i];
    for (int64_t i = 0; i < n; ++i) update[i] = powf(hist[i], 0.5f);
    for (int64_t i = 0; i < n; ++i) update[i] += delta;
    for (int64_t i = 0; i < n; ++i) update[i] = grad[i] / update[i];
    for (int64_t i = 0; i < n; ++i) grad[i] = local_rate * update[i];
  }
  free(update);
  for (int64_t i = 0; i < n; ++i) w[i] -= grad[i];
}

/* ============================================================ whole step ==================== */
static float* falloc(size_t n) {
  float* p = (float*)calloc(n ? n : 1, sizeof(float));
  if (!p) { fprintf(stderr, "vv_oracle: out of memory (%zu floats)\n", n); abort(); }
  return p;
}
/* The big per-iteration arrays (the reference's blobs live for the whole run: Blob::Reshape allocates once,
 * blob.cpp:14-27) come from a small pool of blocks that survive between calls, so that a timed iteration does not pay
 * for page faults on a gigabyte of fresh memory.  Contents are NOT zeroed.  Calls are serialised (one step at a time). */
#define ORC_POOL 32
static struct { float* p; size_t n; int used; } g_pool[ORC_POOL];
static float* palloc(size_t n) {
  int best = -1;
  for (int i = 0; i < ORC_POOL; ++i)
    if (!g_pool[i].used && g_pool[i].p && g_pool[i].n >= n && (best < 0 || g_pool[i].n < g_pool[best].n)) best = i;
  if (best >= 0 && g_pool[best].n <= 2 * n + 1024) { g_pool[best].used = 1; return g_pool[best].p; }
  for (int i = 0; i < ORC_POOL; ++i)
    if (!g_pool[i].used && !g_pool[i].p) {
      g_pool[i].p = (float*)aligned_alloc(64, ((n ? n : 1) * sizeof(float) + 63) / 64 * 64);
      if (!g_pool[i].p) { fprintf(stderr, "vv_oracle: out of memory (%zu floats)\n", n); abort(); }
      g_pool[i].n = n; g_pool[i].used = 1;
      return g_pool[i].p;
    }
  for (int i = 0; i < ORC_POOL; ++i)                     /* every slot holds a block of another size: replace a free one */
    if (!g_pool[i].used) {
      free(g_pool[i].p);
      g_pool[i].p = (float*)aligned_alloc(64, ((n ? n : 1) * sizeof(float) + 63) / 64 * 64);
      if (!g_pool[i].p) { fprintf(stderr, "vv_oracle: out of memory (%zu floats)\n", n); abort(); }
      g_pool[i].n = n; g_pool[i].used = 1;
      return g_pool[i].p;
    }
  fprintf(stderr, "vv_oracle: buffer pool exhausted\n"); abort();
}
static void pfree(float* p) {
  for (int i = 0; i < ORC_POOL; ++i) if (g_pool[i].p == p) { g_pool[i].used = 0; return; }
  free(p);
}

/* ORC_TIMING=1: wall time of the phases of orc_forward_backward on stderr (where a CPU run of the step spends its time) */
static double orc_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
#define ORC_PHASE(name) do { if (timing) { const double t_ = orc_now(); fprintf(stderr, "[orc] %-28s %8.1f ms\n", name, (t_ - tph) * 1e3); tph = t_; } } while (0)

/* Net::ForwardBackward over mednet_embedding_train.prototxt (TRAIN phase), layer by layer. */
void orc_forward_backward(const orc_step_cfg* cfg, const float* table, const int32_t* idx,
                          const int32_t* last_src, const float* W, const float* b,
                          orc_step_out* out) {
  const int B = cfg->B, C = cfg->C, Nn = cfg->Nn, F = cfg->F, D = cfg->D;
  const int CN = C + Nn, R = CN * B, Q = 1 + Nn;
  const int nt = threads_for((double)R * (F > D ? F : D));
  const int timing = getenv("ORC_TIMING") != NULL;
  double tph = orc_now();

  /* --- data layer copy (…data_layer.cpp:439-452,856-875) then SLICE dim 1 + CONCAT dim 0
   * (prototxt :48-131; slice_layer.cpp:79-105, concat_layer.cpp:45-83) + FLATTEN (a reshape): X row = ch*B+b */
  float* data = palloc((size_t)B * CN * F);
#pragma omp parallel for num_threads(nt)
  for (int i = 0; i < B * CN; ++i) {
    float* dst = data + (size_t)i * F;
    if (idx[i] >= 0) memcpy(dst, table + (size_t)idx[i] * F, sizeof(float) * F);
    else memset(dst, 0, sizeof(float) * F);
    if (last_src && last_src[i] != idx[i])
      dst[F - 1] = last_src[i] >= 0 ? table[(size_t)last_src[i] * F + F - 1] : 0.f;
  }
  float* X = palloc((size_t)R * F);
  int32_t* ones = (int32_t*)malloc(sizeof(int32_t) * (size_t)(CN > Q ? CN : Q));
  for (int k = 0; k < (CN > Q ? CN : Q); ++k) ones[k] = 1;
#pragma omp parallel for num_threads(nt)
  for (int p = 0; p < nt; ++p) {                      /* the slice of items [b0, b1): one piece per channel */
    const int b0 = (int)((int64_t)B * p / nt), b1 = (int)((int64_t)B * (p + 1) / nt);
    if (b1 == b0) continue;
    float** piece = (float**)malloc(sizeof(float*) * (size_t)CN);
    for (int ch = 0; ch < CN; ++ch) piece[ch] = X + ((size_t)ch * B + b0) * F;
    orc_split_pieces(b1 - b0, F, CN, ones, data + (size_t)b0 * CN * F, piece);
    free(piece);
  }
  pfree(data);

  ORC_PHASE("data + slice/concat");
  /* --- fc7 INNER_PRODUCT (inner_product_layer.cpp:61-74): Y = X W^T + 1 b^T */
  float* Y = palloc((size_t)R * D);
  orc_inner_product_fwd(R, D, F, X, W, b, Y);
  ORC_PHASE("fc7 forward");
  /* --- RELU (relu_layer.cpp:10-20), DROPOUT in place on ip2 (dropout_layer.cpp:34-50) */
  float* H = palloc((size_t)R * D);
  const float slope = cfg->relu_negative_slope;
  const int64_t RD = (int64_t)R * D, BD = (int64_t)B * D;
#pragma omp parallel for num_threads(nt)
  for (int p = 0; p < nt; ++p) {
    const int64_t o = RD * p / nt, n = RD * (p + 1) / nt - o;
    orc_relu_fwd(n, Y + o, slope, H + o);
    if (cfg->dropout_ratio > 0.f) orc_dropout_fwd(n, H + o, cfg->dropout_mask + o, cfg->dropout_ratio, 1, H + o);
  }
  /* --- SLICE dim 0 (prototxt :232-257): E_ch = H[ch*B .. (ch+1)*B), no copy */
  /* --- ELTWISE SUM with coeffs (eltwise_layer.cpp:65-71): A = sum_j c_j E_cj */
  float* A = falloc((size_t)B * D);
  if (C > 1) {
#pragma omp parallel for num_threads(nt)
    for (int p = 0; p < nt; ++p) {
      const int64_t o = BD * p / nt, n = BD * (p + 1) / nt - o;
      const float** E = (const float**)malloc(sizeof(float*) * (size_t)C);
      for (int j = 1; j < C; ++j) E[j - 1] = H + (size_t)j * BD + o;
      orc_eltwise_fwd(1, n, C - 1, E, cfg->ctx_coeff, A + o);
      free(E);
    }
  }
  ORC_PHASE("relu/dropout + context sum");
  /* --- NORMALIZATION of the context mean (prototxt :281-288) */
  float* Ahat = falloc((size_t)B * D);
  orc_normalize_fwd(B, D, A, Ahat);
  /* --- CONCAT dim 0 of target + negatives, NORMALIZATION, SLICE (prototxt :290-342) */
  float* PN = palloc((size_t)Q * B * D);
  {
    const float** piece = (const float**)malloc(sizeof(float*) * (size_t)Q);
    piece[0] = H;
    for (int k = 0; k < Nn; ++k) piece[1 + k] = H + (size_t)(C + k) * BD;
    for (int k = 0; k < Q; ++k) ones[k] = B;                                          /* concat_layer.cpp:48-55 */
    orc_join_pieces(1, D, Q, ones, piece, PN);
    free(piece);
  }
  float* Phat = palloc((size_t)Q * B * D);
  orc_normalize_fwd(Q * B, D, PN, Phat);
  ORC_PHASE("normalisations + concat");
  /* --- ELTWISE PROD + SUM (prototxt :354-629): s+ replicated Nn times, s-[b][k] */
  /* the per-q layers work on B x D values each: teams of the size the stand-alone layer functions pick for that much
   * work (alternating between team sizes from one parallel region to the next costs more than the regions do) */
  const int ntq = threads_for((double)BD);
  float* s_true = falloc((size_t)B * (Nn > 0 ? Nn : 1));
  float* s_bogus = falloc((size_t)B * (Nn > 0 ? Nn : 1));
  float* prod = falloc((size_t)B * D);
  float* col = falloc((size_t)B);
  for (int q = 0; q < Q; ++q) {
    const float* P = Phat + (size_t)q * B * D;
#pragma omp parallel for num_threads(ntq)
    for (int p = 0; p < ntq; ++p) {                                                   /* eltwise :61-65 */
      const int64_t o = BD * p / ntq, n = BD * (p + 1) / ntq - o;
      const float* two[2] = { Ahat + o, P + o };
      orc_eltwise_fwd(0, n, 2, two, NULL, prod + o);
    }
    if (q == 0) orc_sum_fwd(B, D, Nn, prod, s_true);                                  /* sum :43-47 */
    else {
      orc_sum_fwd(B, D, 1, prod, col);
      for (int bb = 0; bb < B; ++bb) s_bogus[(size_t)bb * Nn + (q - 1)] = col[bb];    /* concat dim 1 */
    }
  }
  ORC_PHASE("prod + sum forward");
  /* --- MAX_MARGIN_LOSS (prototxt :655-671) */
  const int count = B * Nn;
  float* wrep = NULL;                 /* the 3rd bottom: item weight replicated over the Nn terms (a SUM layer) */
  if (cfg->item_weight) {
    wrep = falloc((size_t)count);
    for (int bb = 0; bb < B; ++bb) for (int k = 0; k < Nn; ++k) wrep[(size_t)bb * Nn + k] = cfg->item_weight[bb];
  }
  orc_max_margin_fwd(count, s_true, s_bogus, wrep, cfg->margin, cfg->norm, &out->loss,
                     &out->violations);
  out->loss *= cfg->loss_weight;                                                      /* layer.hpp:416-422 */

  /* ================= backward (Net::BackwardFromTo, net.cpp:567-578) ================= */
  float* d_true = falloc((size_t)count + 1);
  float* d_bogus = falloc((size_t)count + 1);
  orc_max_margin_bwd(count, s_true, s_bogus, wrep, cfg->margin, cfg->norm, cfg->loss_weight,
                     d_true, d_bogus);
  free(wrep);
  if (cfg->global_count > 0 && cfg->global_count != count) {
    /* data-parallel shard: the loss normaliser is the GLOBAL B*Nn */
    const float f = (float)count / (float)cfg->global_count;
    for (int i = 0; i < count; ++i) { d_true[i] *= f; d_bogus[i] *= f; }
  }
  ORC_PHASE("loss fwd + bwd");
  /* SUM / PROD backward per q; the Q copies of context_feature are tops of a SPLIT (split_layer.cpp:36-51) */
  float* dAhat = falloc((size_t)B * D);
  float* dPhat = palloc((size_t)Q * B * D);
  float* dAq = palloc((size_t)Q * B * D);
  float* dprod = falloc((size_t)B * D);
  for (int q = 0; q < Q; ++q) {
    if (q == 0) orc_sum_bwd(B, D, Nn, d_true, dprod);                                 /* sum :56-82 */
    else {
      for (int bb = 0; bb < B; ++bb) col[bb] = d_bogus[(size_t)bb * Nn + (q - 1)];
      orc_sum_bwd(B, D, 1, col, dprod);
    }
    const float* P = Phat + (size_t)q * B * D;
#pragma omp parallel for num_threads(ntq)
    for (int p = 0; p < ntq; ++p) {                                                   /* eltwise :116-131 */
      const int64_t o = BD * p / ntq, n = BD * (p + 1) / ntq - o;
      const float* two[2] = { Ahat + o, P + o };
      orc_eltwise_bwd(0, n, 2, two, NULL, NULL, dprod + o, 1, 1, dPhat + (size_t)q * BD + o);
      orc_eltwise_bwd(0, n, 2, two, NULL, NULL, dprod + o, 0, 1, dAq + (size_t)q * BD + o);
    }
  }
#pragma omp parallel for num_threads(nt)
  for (int p = 0; p < nt; ++p) {
    const int64_t o = BD * p / nt, n = BD * (p + 1) / nt - o;
    const float** tops = (const float**)malloc(sizeof(float*) * (size_t)Q);
    for (int q = 0; q < Q; ++q) tops[q] = dAq + (size_t)q * BD + o;
    orc_split_bwd(n, Q, tops, dAhat + o);
    free(tops);
  }
  pfree(dAq);
  ORC_PHASE("sum/prod/split backward");
  float* dPN = palloc((size_t)Q * B * D);
  orc_normalize_bwd(Q * B, D, PN, dPhat, dPN);
  float* dA = falloc((size_t)B * D);
  orc_normalize_bwd(B, D, A, dAhat, dA);
  /* diffs back to the rows of dH: CONCAT backward (concat_layer.cpp:86-100) hands the target and negative rows
   * their pieces of dPN, the ELTWISE SUM backward (:132-138) writes c_j dA into the context rows; the SLICE dim 0
   * backward (slice_layer.cpp:107-120) is then the identity on this layout. */
  float* dH = palloc((size_t)R * D);
  {
    float** piece = (float**)malloc(sizeof(float*) * (size_t)Q);
    piece[0] = dH;
    for (int k = 0; k < Nn; ++k) piece[1 + k] = dH + (size_t)(C + k) * BD;
    orc_split_pieces(1, D, Q, ones, dPN, piece);
    free(piece);
  }
  if (C > 1) {
#pragma omp parallel for num_threads(nt)
    for (int p = 0; p < nt; ++p) {
      const int64_t o = BD * p / nt, n = BD * (p + 1) / nt - o;
      for (int j = 1; j < C; ++j)
        orc_eltwise_bwd(1, n, C - 1, NULL, cfg->ctx_coeff, NULL, dA + o, j - 1, 1, dH + (size_t)j * BD + o);
    }
  }
  ORC_PHASE("normalisation/concat/eltwise bwd");
  /* DROPOUT backward (dropout_layer.cpp:52-68), RELU backward (relu_layer.cpp:23-37) */
  float* dY = palloc((size_t)R * D);
#pragma omp parallel for num_threads(nt)
  for (int p = 0; p < nt; ++p) {
    const int64_t o = RD * p / nt, n = RD * (p + 1) / nt - o;
    if (cfg->dropout_ratio > 0.f) orc_dropout_bwd(n, dH + o, cfg->dropout_mask + o, cfg->dropout_ratio, 1, dH + o);
    orc_relu_bwd(n, Y + o, dH + o, slope, dY + o);
  }
  ORC_PHASE("dropout/relu backward");
  /* INNER_PRODUCT backward (inner_product_layer.cpp:76-106): dW = dY^T X, db = dY^T 1 */
  orc_inner_product_bwd(R, D, F, X, W, dY, out->dW, out->db, NULL);
  if (out->dW && cfg->ip_regularization > 0.f) {
    const float f = (float)(1.0 + cfg->ip_regularization / 2);
    for (size_t i = 0; i < (size_t)D * F; ++i) out->dW[i] *= f;
  }

  ORC_PHASE("fc7 backward");
  if (out->Y) memcpy(out->Y, Y, sizeof(float) * (size_t)R * D);
  if (out->H) memcpy(out->H, H, sizeof(float) * (size_t)R * D);
  if (out->ctx) memcpy(out->ctx, Ahat, sizeof(float) * (size_t)B * D);
  if (out->posneg) memcpy(out->posneg, Phat, sizeof(float) * (size_t)Q * B * D);
  if (out->s_true) memcpy(out->s_true, s_true, sizeof(float) * (size_t)B * Nn);
  if (out->s_bogus) memcpy(out->s_bogus, s_bogus, sizeof(float) * (size_t)B * Nn);
  if (out->dY) memcpy(out->dY, dY, sizeof(float) * (size_t)R * D);

  pfree(X); pfree(Y); pfree(H); free(A); free(Ahat); pfree(PN); pfree(Phat); free(s_true);
  free(s_bogus); free(prod); free(col); free(d_true); free(d_bogus); free(dAhat); pfree(dPhat);
  free(dprod); pfree(dPN); free(dA); pfree(dH); pfree(dY); free(ones);
}

/* videovec_extraction.prototxt:179-205 (fc7 INNER_PRODUCT + RELU), optional NORMALIZATION as in
 * the TEST branch of mednet_embedding_train.prototxt:344-352 */
void orc_embed(int n, int F, int D, const float* table, const int32_t* rows, const float* W,
               const float* b, int relu, int l2norm, float* out) {
  float* X = falloc((size_t)n * F);
  for (int i = 0; i < n; ++i)
    memcpy(X + (size_t)i * F, table + (size_t)(rows ? rows[i] : i) * F, sizeof(float) * F);
  orc_sgemm(0, 1, n, D, F, 1.f, X, W, 0.f, out);
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < D; ++d) {
      float y = out[(size_t)i * D + d] + (b ? b[d] : 0.f);
      out[(size_t)i * D + d] = relu ? fmaxf(y, 0.f) : y;
    }
  if (l2norm) {
    float* t = falloc((size_t)n * D);
    orc_normalize_fwd(n, D, out, t);
    memcpy(out, t, sizeof(float) * (size_t)n * D);
    free(t);
  }
  free(X);
}

/* ============================================================ retrieval statistics ========== */
static int cls_of(int id, const int32_t* map_ids, const int32_t* map_cls, int n_map) {
  for (int i = 0; i < n_map; ++i) if (map_ids[i] == id) return map_cls[i];
  return 0;
}
typedef struct { float d; int i; } orc_di;
static int cmp_di(const void* a, const void* b) {
  const orc_di* x = (const orc_di*)a; const orc_di* y = (const orc_di*)b;
  if (x->d < y->d) return -1;
  if (x->d > y->d) return 1;
  return x->i - y->i;
}
/* retrieval_stats_layer.cpp:104-141, 143-355 */
void orc_retrieval_stats(int n, int dim, const float* feat, const int32_t* video_ids,
                         const int32_t* map_ids, const int32_t* map_cls, int n_map,
                         int exclude_same_video, float* mean_ap, float* hit1, float* hit5) {
  float* dist = falloc((size_t)n * n);
  orc_sgemm(0, 1, n, n, dim, -2.f, feat, feat, 0.f, dist);                           /* :208-209 */
  orc_di* row = (orc_di*)malloc(sizeof(orc_di) * (size_t)n);
  double s_ap = 0, s_1 = 0, s_5 = 0, npos = 0;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < n; ++j) { row[j].d = dist[(size_t)i * n + j]; row[j].i = j; }
    row[i].d = -1e15f;                                                               /* :228-229 */
    qsort(row, (size_t)n, sizeof(orc_di), cmp_di);                                    /* :233 */
    const int label = cls_of(video_ids[i], map_ids, map_cls, n_map);
    if (label < 0) continue;                                                          /* :246-248 */
    double ap = 0, a1 = 0, a5 = 0, val = 0, ret = 0;                                  /* :104-141 */
    for (int k = 1; k < n; ++k) {
      const int j = row[k].i;
      if (video_ids[j] != video_ids[i] || !exclude_same_video) {
        val += 1;
        if (cls_of(video_ids[j], map_ids, map_cls, n_map) == label) {
          if (val <= 1) a1 += 1;
          if (val <= 5) a5 += 1;
          ret += 1;
          ap += ret / val;
        }
      }
    }
    if (ret > 0) ap /= ret;
    a5 /= 5;
    s_ap += ap; s_1 += a1; s_5 += a5; npos += 1;
  }
  *mean_ap = (float)(s_ap / npos); *hit1 = (float)(s_1 / npos); *hit5 = (float)(s_5 / npos);   /* :347-349 */
  free(row); free(dist);
}
