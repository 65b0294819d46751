// api.hip -- implementation of the C ABI in include/videovec.h (gfx950 / HIP only; there is no CPU
// fallback: every entry point fails with VV_ERR_NOGPU / VV_ERR_HIP when no device is usable).
#include "../../include/videovec.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <map>
#include <unordered_map>
#include <string>
#include <thread>
#include <vector>

#include "vv_ctx.h"

namespace vv { int gemm_variant(); bool ablate_on(); thread_local const KernelOpts* g_ko = nullptr; }
// every entry point that launches kernels: the device of the context, and ITS kernel options for the launchers on this thread
#define VV_ENTER(c) do { HIPCHK(hipSetDevice((c)->device)); vv::g_ko = &(c)->ko; \
    if ((c)->comm && vv::comm_failed((c)->comm)) return fail(VV_ERR_HIP, "the data-parallel exchange failed: %s", vv::comm_error((c)->comm)); } while (0)
// environment of a PRODUCT option (read once per context) / of a lab switch (ignored unless the library was built with -DVV_LAB)
static inline const char* opt_env(const char* name) { return getenv(name); }
#ifdef VV_LAB
static inline const char* lab_env(const char* name) { return getenv(name); }
#else
static inline const char* lab_env(const char*) { return nullptr; }
#endif
using namespace vv;

static int upd_pending_guard(vv_ctx* c, const char* who);      // (vv_update_hint, below)
thread_local char vv_g_err[512] = "";
int vv_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(vv_g_err, sizeof(vv_g_err), fmt, ap);
  va_end(ap);
  return code;
}

static void dfree(void* p) { if (p) (void)hipFree(p); }

namespace vv { thread_local ProfPair g_prof; }

// Arms a pair of events for the launcher called next: its (first / last) kernel is dispatched with
// hipExtLaunchKernelGGL, which stamps them from the dispatch packet itself (vv_internal.h: VV_LAUNCH).
static void prof_begin(vv_ctx* c, const char* name, hipEvent_t* e0, hipEvent_t* e1) {
  *e0 = *e1 = nullptr;
  if (!c->prof || (c->prof_calls % (uint64_t)c->prof_every) != (uint64_t)c->prof_every - 1) return;   // the N-th, 2N-th, ... step
  if (!c->prof_only.empty() && c->prof_only.find(std::string(",") + name + ",") == std::string::npos) return;
  auto& pe = c->prof_map[name];
  if (pe.ev.size() >= 8192) return;
  while (c->ev_pool.size() < c->ev_used + 2) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    c->ev_pool.push_back(e);
  }
  *e0 = c->ev_pool[c->ev_used++]; *e1 = c->ev_pool[c->ev_used++];
  vv::g_prof.start = *e0; vv::g_prof.stop = *e1;
}
static void prof_end(vv_ctx* c, const char* name, hipEvent_t e0, hipEvent_t e1) {
  if (!e0) return;
  // a launcher that did not take the events (a kernel variant launched the plain way): fall back to stream records
  if (vv::g_prof.start) { (void)hipEventRecord(e0, c->stream); vv::g_prof.start = nullptr; }
  if (vv::g_prof.stop) { (void)hipEventRecord(e1, c->stream); vv::g_prof.stop = nullptr; }
  c->prof_map[name].ev.emplace_back(e0, e1);
}
// VV_TRACE_HOST=<ms>: report every launcher call that keeps the HOST longer than that (first use of a kernel variant,
// a runtime pool growing, ...) -- the GPU queue runs dry behind such a call.
static inline double host_now_ms() {
  timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
#define PROFILED(c, name, call)                 \
  do {                                          \
    hipEvent_t e0_, e1_;                        \
    prof_begin(c, name, &e0_, &e1_);            \
    const double th_ = (c)->trace_host_ms >= 0 ? host_now_ms() : 0.0; \
    call;                                       \
    if ((c)->trace_host_ms >= 0) { const double d_ = host_now_ms() - th_; if (d_ > (c)->trace_host_ms) fprintf(stderr, "[vv host] %s: %.3f ms (call %llu)\n", name, d_, (unsigned long long)c->iter); } \
    prof_end(c, name, e0_, e1_);                \
  } while (0)

extern "C" {

const char* vv_last_error(void) { return vv_g_err; }
const char* vv_version(void) { return "videovec-mi355x 0.1 (gfx950)"; }

void vv_step_cfg_default(vv_step_cfg* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->margin = 2.0f; cfg->norm = VV_NORM_L2; cfg->loss_weight = 1.0f;
  cfg->momentum = 0.9f; cfg->weight_decay = 5e-4f; cfg->lr = 1e-3f;
  cfg->lr_mult[0] = 1.f; cfg->lr_mult[1] = 2.f;
  cfg->decay_mult[0] = 1.f; cfg->decay_mult[1] = 0.f;
  cfg->reg = VV_REG_L2;
  cfg->solver_type = VV_SOLVER_SGD; cfg->delta = 1e-8f; cfg->ip_regularization = 0.f;
}

static int create_init(vv_ctx* c);
static inline int comm_join(vv_ctx* c) { return vv_comm_join(c); }

int vv_create(int device, int prec, vv_ctx** out) {
  if (!out) return fail(VV_ERR_ARG, "vv_create: out is NULL");
  if (prec != VV_PREC_F16 && prec != VV_PREC_BF16) return fail(VV_ERR_ARG, "vv_create: bad prec %d", prec);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(VV_ERR_NOGPU, "vv_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= n) return fail(VV_ERR_ARG, "vv_create: device %d out of range (%d)", device, n);
  HIPCHK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(VV_ERR_NOGPU, "vv_create: device %d is %s; this build targets gfx950 only", device,
                prop.gcnArchName);
  vv_ctx* c = new vv_ctx();
  c->device = device; c->prec = prec;
  const int rc = create_init(c);
  if (rc != VV_OK) { vv_destroy(c); return rc; }        // the message of the failing call stays in vv_last_error()
  *out = c;
  return VV_OK;
}

// everything vv_create allocates; a failure half way leaves a context vv_destroy can release
static int create_init(vv_ctx* c) {
  HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;

  HIPCHK(hipMalloc(&c->scales, sizeof(Scales)));
  Scales h = {1.f, 1.f, 1.f, 0u};
  HIPCHK(hipMemcpy(c->scales, &h, sizeof(h), hipMemcpyHostToDevice));
  HIPCHK(hipMalloc(&c->wmax_blocks, 2 * WMAX_SLOTS * sizeof(float)));
  HIPCHK(hipMemset(c->wmax_blocks, 0, 2 * WMAX_SLOTS * sizeof(float)));
  HIPCHK(hipMalloc(&c->loss2, 2 * sizeof(float)));
  HIPCHK(hipMemset(c->loss2, 0, 2 * sizeof(float)));
  // product options: initial values from the environment, per context (vv_set_option changes them afterwards)
  if (const char* v = opt_env("VV_WGRAD_TR")) c->ko.wgrad_tr = atoi(v) != 0;
  if (const char* v = opt_env("VV_FWD_LEAD")) c->ko.fwd_lead = atoi(v);
  if (const char* v = opt_env("VV_FWD_MERGE")) c->ko.fwd_merge = atoi(v);
  if (const char* v = opt_env("VV_SCORE_STREAM")) c->ko.score_stream = atoi(v);
  if (const char* v = opt_env("VV_SCORE_PF")) c->ko.score_pf = atoi(v);
  if (const char* v = opt_env("VV_WGRAD_LEAN")) c->ko.wgrad_lean = atoi(v);
  if (const char* v = opt_env("VV_SEG_BWD")) c->seg_bwd = atoi(v) != 0;
  if (const char* v = opt_env("VV_H16")) c->h16 = atoi(v) != 0;
  if (const char* v = opt_env("VV_SLAB16")) c->slab16 = atoi(v) != 0;
  if (const char* v = opt_env("VV_V16")) c->v16 = atoi(v) != 0;
  if (const char* v = opt_env("VV_DEDUP")) c->dedup = atoi(v) != 0;
  if (const char* v = opt_env("VV_FUSE_UPDATE")) c->fuse_update = atoi(v) != 0;
  if (const char* v = opt_env("VV_DROP_DEDUP")) c->drop_dedup = atoi(v) != 0;
  if (const char* v = opt_env("VV_COMM_GATE")) c->comm_gate = atoi(v) != 0;
  if (const char* v = opt_env("VV_COMM_INLINE")) c->comm_inline = atoi(v) != 0;
  if (const char* v = opt_env("VV_COMM_TEST_DELAY_US")) c->comm_test_delay_us = atoi(v);
  // lab switches (-DVV_LAB builds only; several of them produce WRONG results by design)
  if (const char* v = lab_env("VV_GEMM_VARIANT")) c->ko.gemm_variant = atoi(v);
  if (const char* v = lab_env("VV_ABLATE")) c->ko.ablate = atoi(v);
  if (const char* v = lab_env("VV_LAB_FWD_ABL")) c->ko.lab_fwd_abl = atoi(v);
  if (const char* v = lab_env("VV_LAB_WG_ABL")) c->ko.lab_wg_abl = atoi(v);
  if (const char* v = lab_env("VV_FWD_RING10")) c->ko.fwd_ring10 = atoi(v) != 0;
  if (const char* v = lab_env("VV_PH_MQ")) c->ko.ph_mq = atoi(v);
  if (const char* v = lab_env("VV_SCORE_REG")) c->ko.score_reg = atoi(v);
  if (const char* v = lab_env("VV_SCORE_WAVES")) c->ko.score_waves = atoi(v);
  if (const char* v = lab_env("VV_SCORE_RR")) c->ko.score_rr = atoi(v);
  if (const char* v = lab_env("VV_LAB_SCORE_PIPE")) c->ko.lab_score_pipe = atoi(v);
  if (const char* v = lab_env("VV_GUARD_PROACTIVE")) c->guard_proactive = atoi(v) != 0;
  if (const char* v = lab_env("VV_FUSE_KEEP_GRADS")) c->fuse_keep_grads = atoi(v) != 0;
  if (const char* v = lab_env("VV_COMM_SKIP_AR1")) c->comm_skip_ar1 = atoi(v) != 0;
  if (const char* v = lab_env("VV_DEDUP_GATE")) c->dd_gate_word = atoi(v) != 0;
  if (const char* v = lab_env("VV_DEDUP_LDS_KB")) c->dd_lds_kb = atoi(v);
  if (const char* v = lab_env("VV_TRACE_HOST")) c->trace_host_ms = atof(v);
  c->trace_waits = lab_env("VV_TRACE_WAITS") != nullptr;
  vv::g_ko = &c->ko;
  HIPCHK(hipMalloc(&c->dd_info_all, vv_ctx::kDdSets * 4 * sizeof(int32_t)));
  HIPCHK(hipMemset(c->dd_info_all, 0, vv_ctx::kDdSets * 4 * sizeof(int32_t)));
  HIPCHK(hipStreamCreateWithFlags(&c->dd_stream, hipStreamNonBlocking));
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && ncu > 0) c->n_cu = ncu; }
  for (int i = 0; i < vv_ctx::kDdSets; ++i) {
    c->dd_set[i].info = c->dd_info_all + 4 * i;
    HIPCHK(hipEventCreateWithFlags(&c->dd_set[i].done, hipEventDisableTiming));
  }
  c->dd_info = c->dd_set[0].info;
  if (const char* v = lab_env("VV_DEDUP_ASYNC")) c->dd_async = atoi(v) != 0;
  if (const char* v = lab_env("VV_DEDUP_SPIN_US")) c->dd_spin_us = atof(v);
  HIPCHK(hipHostMalloc((void**)&c->U_host, 2 * sizeof(int32_t), hipHostMallocMapped));   // {U, saturated f16 gradient sums}
  c->U_host[0] = c->U_host[1] = 0;
  HIPCHK(hipHostGetDevicePointer((void**)&c->U_host_dev, c->U_host, 0));
  HIPCHK(hipMalloc(&c->w_gate, (W_CHUNKS_MAX + 2) * W_GATE_STRIDE * sizeof(int32_t)));
  HIPCHK(hipMemset(c->w_gate, 0, (W_CHUNKS_MAX + 2) * W_GATE_STRIDE * sizeof(int32_t)));
  c->pub_count = c->w_gate + W_CHUNKS_MAX * W_GATE_STRIDE;      // the arrival counter of the publishing kernels: a line of its own
  c->pub_count0 = c->pub_count + W_GATE_STRIDE;                 // ... and the one of the FIRST chunk's kernel, which runs on the compute stream beside them
  if (const char* v = opt_env("VV_COMM_FIRST_INLINE")) c->overlap_first_inline = atoi(v) != 0;
  if (const char* v = opt_env("VV_WGRAD_UPDATE")) c->wgrad_update = atoi(v) != 0;
  { const char* nc = opt_env("VV_COMM_CHUNKS"); if (nc) c->n_chunks = std::max(1, std::min(W_CHUNKS_MAX, atoi(nc))); }
  HIPCHK(hipEventCreateWithFlags(&c->ev_chunk0, hipEventDisableTiming));
  HIPCHK(hipHostMalloc((void**)&c->gate_err, sizeof(int32_t), hipHostMallocMapped));
  *c->gate_err = 0;
  HIPCHK(hipHostGetDevicePointer((void**)&c->gate_err_dev, c->gate_err, 0));
  HIPCHK(hipMalloc(&c->gg_bound, GG_BOUND_SLOTS * GG_BOUND_STRIDE * sizeof(unsigned long long)));
  HIPCHK(hipMemset(c->gg_bound, 0, GG_BOUND_SLOTS * GG_BOUND_STRIDE * sizeof(unsigned long long)));
  HIPCHK(hipMalloc(&c->gg, sizeof(GradGuard)));
  { GradGuard g0; memset(&g0, 0, sizeof(g0)); g0.mul = 1.f; HIPCHK(hipMemcpy(c->gg, &g0, sizeof(g0), hipMemcpyHostToDevice)); }
  HIPCHK(hipHostMalloc((void**)&c->gmax_host, 32 * sizeof(unsigned long long), hipHostMallocMapped));
  memset(c->gmax_host, 0, 32 * sizeof(unsigned long long));
  HIPCHK(hipHostGetDevicePointer((void**)&c->gmax_host_dev, c->gmax_host, 0));
  HIPCHK(hipHostMalloc((void**)&c->seq_host, 2 * sizeof(int32_t), hipHostMallocMapped));
  c->seq_host[0] = c->seq_host[1] = 0;
  HIPCHK(hipHostGetDevicePointer((void**)&c->seq_host_dev, c->seq_host, 0));
  return VV_OK;
}

int vv_device_query(int device, char* buf, size_t n) {
  if (!buf || n == 0) return fail(VV_ERR_ARG, "vv_device_query: no buffer");
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) return fail(VV_ERR_NOGPU, "vv_device_query: no HIP device visible");
  if (device < 0 || device >= cnt) return fail(VV_ERR_ARG, "vv_device_query: device %d out of range (%d)", device, cnt);
  hipDeviceProp_t p;
  HIPCHK(hipGetDeviceProperties(&p, device));
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipSetDevice(device));
  (void)hipMemGetInfo(&free_b, &total_b);
  snprintf(buf, n,
           "Device id:                     %d\nName:                          %s\nArchitecture:                  %s\n"
           "Compute units:                 %d\nClock rate:                    %d kHz\nMemory clock rate:             %d kHz\n"
           "Total global memory:           %zu\nFree global memory:            %zu\nShared memory (LDS) per block: %zu\n"
           "Registers per block:           %d\nWavefront size:                %d\nMax threads per block:         %d\n"
           "L2 cache size:                 %d\nConcurrent kernels:            %s\n",
           device, p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, (size_t)p.totalGlobalMem, free_b,
           (size_t)p.sharedMemPerBlock, p.regsPerBlock, p.warpSize, p.maxThreadsPerBlock, p.l2CacheSize, p.concurrentKernels ? "Yes" : "No");
  return VV_OK;
}

static void free_batch(vv_ctx* c) {
  dfree(c->idx_dev); dfree(c->rows); dfree(c->H); dfree(c->dYh); dfree(c->dbp); dfree(c->slab_sc); c->slab_sc = nullptr;
  dfree(c->loss_part); dfree(c->viol_part); dfree(c->s_true); dfree(c->s_bogus);
  dfree(c->coeff); dfree(c->slabs); dfree(c->item_w); c->item_w = nullptr;
  if (c->dd_stream) (void)hipStreamSynchronize(c->dd_stream);
  dfree(c->dd_agg); dfree(c->dd_pos); dfree(c->dYu);
  for (int i = 0; i < vv_ctx::kDdSets; ++i) {
    vv_ctx::DdSet& d = c->dd_set[i];
    dfree(d.rows); dfree(d.slot_of); dfree(d.uniq); dfree(d.map); dfree(d.ord); dfree(d.cnt); dfree(d.seg);
    d.rows = d.slot_of = d.uniq = d.map = d.ord = d.cnt = d.seg = nullptr; d.used_seq = 0;
  }
  c->dd_rows = nullptr;
  dfree(c->segV); dfree(c->seg_rec); dfree(c->seg_dbp); dfree(c->gg_slots);
  c->segV = nullptr; c->seg_rec = nullptr; c->seg_dbp = nullptr; c->gg_slots = nullptr; c->gg_nslot = 0;
  c->sg_adj = 0; c->gg_seq0 = 0;
  c->dd_agg = nullptr; c->dd_slot_of = c->dd_uniq = c->dd_map = c->dd_ord = c->dd_cnt = c->dd_seg = c->dd_pos = nullptr;
  c->dYu = nullptr;
  c->idx_dev = c->rows = nullptr; c->H = nullptr; c->dYh = nullptr; c->dbp = nullptr;
  c->loss_part = c->viol_part = c->s_true = c->s_bogus = c->coeff = nullptr; c->slabs = nullptr;
  c->slab_bytes = 0; c->B = c->C = c->Nn = c->R = c->Rp = 0; c->coeff_host.clear();
  c->have_fwd = false;
  c->red_lazy = c->grads_stale = false;       // a reduction that was still due named the buffers just freed (red_args)
}

int vv_destroy(vv_ctx* c) {
  if (!c) return VV_OK;
  (void)hipSetDevice(c->device);
  if (vv::g_ko == &c->ko) vv::g_ko = nullptr;        // (this thread's launcher options pointed into the context)
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) { vv::comm_destroy(c->comm); c->comm = nullptr; }    // drains the communication stream: before any buffer it uses goes
  vv_ops_release(c);
  free_batch(c);
  dfree(c->table); dfree(c->patch_desc); dfree(c->W); dfree(c->b); dfree(c->hW); dfree(c->hb); dfree(c->Wh);
  dfree(c->scales); dfree(c->wmax_blocks); dfree(c->grads_own); dfree(c->mask); dfree(c->loss2);
  dfree(c->dd_key); dfree(c->dd_info_all); dfree(c->gg); dfree(c->gg_bound); dfree(c->w_gate);
  if (c->gate_err) (void)hipHostFree(c->gate_err);
  if (c->gmax_host) (void)hipHostFree(c->gmax_host);
  for (int i = 0; i < vv_ctx::kDdSets; ++i) {
    if (c->dd_set[i].done) (void)hipEventDestroy(c->dd_set[i].done);
  }
  if (c->dd_stream) (void)hipStreamDestroy(c->dd_stream);
  if (c->ev_chunk) (void)hipEventDestroy(c->ev_chunk);
  if (c->ev_idx) (void)hipEventDestroy(c->ev_idx);
  if (c->ev_chunk0) (void)hipEventDestroy(c->ev_chunk0);
  if (c->U_host) (void)hipHostFree(c->U_host);
  for (int i = 0; i < vv_ctx::kStage; ++i)
    if (c->stage_host[i]) (void)hipHostFree(c->stage_host[i]);
  if (c->seq_host) (void)hipHostFree(c->seq_host);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);

  delete c;
  return VV_OK;
}

// Per-context switches by name (include/videovec.h).  A name of a lab switch is refused unless the library was built with -DVV_LAB.
int vv_set_option(vv_ctx* c, const char* name, double value) {
  if (!c || !name) return fail(VV_ERR_ARG, "vv_set_option: ctx / name is NULL");
  const std::string n(name);
  const int iv = (int)value;
  if (n == "dedup") return vv_set_dedup(c, iv);
  if (n == "seg_bwd") { c->seg_bwd = iv != 0; return VV_OK; }
  if (n == "drop_dedup") { c->drop_dedup = iv != 0; return VV_OK; }
  if (n == "h16") { c->h16 = iv != 0; return VV_OK; }
  if (n == "slab16") { c->slab16 = iv != 0; return VV_OK; }
  if (n == "v16") { c->v16 = iv != 0; return VV_OK; }
  if (n == "fuse_update") { c->fuse_update = iv != 0; return VV_OK; }
  if (n == "fwd_lead") { c->ko.fwd_lead = iv; return VV_OK; }
  if (n == "fwd_merge") { c->ko.fwd_merge = iv; return VV_OK; }
  if (n == "wgrad_tr") { c->ko.wgrad_tr = iv != 0; return VV_OK; }
  if (n == "score_stream") { c->ko.score_stream = iv; return VV_OK; }
  if (n == "score_pf") { c->ko.score_pf = iv; return VV_OK; }
  if (n == "wgrad_lean") { c->ko.wgrad_lean = iv; return VV_OK; }
  if (n == "comm_gate") { c->comm_gate = iv != 0; return VV_OK; }
  if (n == "comm_inline") { c->comm_inline = iv != 0; return VV_OK; }
  if (n == "comm_first_inline") { c->overlap_first_inline = iv != 0; return VV_OK; }
  if (n == "wgrad_update") { c->wgrad_update = iv != 0; return VV_OK; }
  if (n == "comm_chunks") { c->n_chunks = std::max(1, std::min(W_CHUNKS_MAX, iv)); return VV_OK; }
  if (n == "comm_test_delay_us") { c->comm_test_delay_us = iv; return VV_OK; }
#ifdef VV_LAB
  if (n == "gemm_variant") { c->ko.gemm_variant = iv; return VV_OK; }
  if (n == "ablate") { c->ko.ablate = iv; return VV_OK; }
  if (n == "lab_fwd_abl") { c->ko.lab_fwd_abl = iv; return VV_OK; }
  if (n == "lab_wg_abl") { c->ko.lab_wg_abl = iv; return VV_OK; }
  if (n == "ph_mq") { c->ko.ph_mq = iv; return VV_OK; }
#endif
  return fail(VV_ERR_ARG, "vv_set_option: unknown option '%s'", name);
}

int vv_get_option(vv_ctx* c, const char* name, double* value) {
  if (!c || !name || !value) return fail(VV_ERR_ARG, "vv_get_option: NULL argument");
  const std::string n(name);
  if (n == "dedup") *value = c->dedup;
  else if (n == "seg_bwd") *value = c->seg_bwd;
  else if (n == "drop_dedup") *value = c->drop_dedup;
  else if (n == "h16") *value = c->h16;
  else if (n == "slab16") *value = c->slab16;
  else if (n == "v16") *value = c->v16;
  else if (n == "fuse_update") *value = c->fuse_update;
  else if (n == "fwd_lead") *value = c->ko.fwd_lead;
  else if (n == "fwd_merge") *value = c->ko.fwd_merge;
  else if (n == "wgrad_tr") *value = c->ko.wgrad_tr;
  else if (n == "score_stream") *value = c->ko.score_stream;
  else if (n == "score_pf") *value = c->ko.score_pf;
  else if (n == "wgrad_lean") *value = c->ko.wgrad_lean;
  else if (n == "comm_gate") *value = c->comm_gate;
  else if (n == "comm_inline") *value = c->comm_inline;
  else if (n == "comm_first_inline") *value = c->overlap_first_inline;
  else if (n == "wgrad_update") *value = c->wgrad_update;
  else if (n == "comm_chunks") *value = c->n_chunks;
  else if (n == "comm_test_delay_us") *value = c->comm_test_delay_us;
  else return fail(VV_ERR_ARG, "vv_get_option: unknown option '%s'", name);
  return VV_OK;
}

int vv_set_dedup(vv_ctx* c, int on) {
  if (!c) return fail(VV_ERR_ARG, "vv_set_dedup: ctx is NULL");
  c->dedup = on != 0;
  c->sg_adj = 0; c->gg_seq0 = 0;                                                  // a fresh start for the gradient scale
  return VV_OK;
}

int vv_dedup_stats(vv_ctx* c, int64_t* rows, int64_t* unique_rows) {
  if (!c) return fail(VV_ERR_ARG, "vv_dedup_stats: ctx is NULL");
  if (!c->have_fwd) return fail(VV_ERR_STATE, "vv_dedup_stats: no forward pass yet");
  VV_ENTER(c);
  HIPCHK(hipStreamSynchronize(c->stream));
  int32_t U = c->R;
  if (c->last_dedup) HIPCHK(hipMemcpy(&U, c->dd_info, sizeof(U), hipMemcpyDeviceToHost));
  if (rows) *rows = c->R;
  if (unique_rows) *unique_rows = U;
  return VV_OK;
}

int vv_grad_scale_stats(vv_ctx* c, int64_t* repeats, float* scale) {
  if (!c) return fail(VV_ERR_ARG, "vv_grad_scale_stats: ctx is NULL");
  if (repeats) *repeats = c->gg_repeats;
  if (scale) *scale = c->sg;
  return VV_OK;
}

int vv_set_stream(vv_ctx* c, void* s) {
  if (!c) return fail(VV_ERR_ARG, "vv_set_stream: ctx is NULL");
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return VV_OK;
}

int vv_synchronize(vv_ctx* c) {
  if (!c) return fail(VV_ERR_ARG, "vv_synchronize: ctx is NULL");
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  HIPCHK(hipStreamSynchronize(c->stream));
  return VV_OK;
}

// ------------------------------------------------------------------------------- table --------
static int table_alloc(vv_ctx* c, int64_t n_rows, int F) {
  if (n_rows <= 0 || F <= 0) return fail(VV_ERR_ARG, "table: n_rows=%lld F=%d", (long long)n_rows, F);
  if (n_rows >= (1ll << 31) - 1) return fail(VV_ERR_ARG, "table: too many rows for int32 indices");
  VV_ENTER(c);
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->D && F != c->F) return fail(VV_ERR_STATE, "table: F=%d differs from the parameters' F=%d", F, c->F);
  dfree(c->table); c->table = nullptr; c->patch_cap = 0;
  c->n_rows = n_rows; c->F = F; c->Fp = (int)round_up(F, F_ALIGN);
  const size_t bytes = (size_t)(n_rows + 1) * c->Fp * sizeof(uint16_t);   // +1: the all-zero row
  HIPCHK(hipMalloc(&c->table, bytes));
  HIPCHK(hipMemsetAsync(c->table, 0, bytes, c->stream));
  return VV_OK;
}

static int set_sx(vv_ctx* c, float sx) {
  c->sx = sx;
  HIPCHK(hipMemcpyAsync(&c->scales->sx, &c->sx, sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return VV_OK;
}

int vv_table_set(vv_ctx* c, const float* rows, int64_t n_rows, int32_t F) {
  if (!c || !rows) return fail(VV_ERR_ARG, "vv_table_set: NULL argument");
  int rc = table_alloc(c, n_rows, F);
  if (rc) return rc;
  // range guard for f16: keep max|x|*sx inside [2^-8, 2^14]; otherwise move it to ~2^8
  float mx = 0.f;
  {
    // one pass over the host table for max|x| and the first non-finite value; split over a few threads (the scan of a
    // 100 GB table on one core is the longest part of set-up)
    const int64_t total = n_rows * F;
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(16, std::thread::hardware_concurrency()), total >> 22));
    std::vector<float> part_mx(nt, 0.f);
    std::vector<int64_t> part_bad(nt, -1);
    auto scan = [&](int t) {
      const int64_t lo = total * t / nt, hi = total * (t + 1) / nt;
      float m = 0.f; int64_t bad = -1;
      for (int64_t i = lo; i < hi; ++i) {
        const float v = fabsf(rows[i]);
        if (!(v <= 3.0e38f)) { bad = i; break; }
        if (v > m) m = v;
      }
      part_mx[t] = m; part_bad[t] = bad;
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(scan, t);
    scan(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < nt; ++t) {
      if (part_bad[t] >= 0) return fail(VV_ERR_ARG, "vv_table_set: non-finite feature at %lld", (long long)part_bad[t]);
      mx = std::max(mx, part_mx[t]);
    }
  }
  float sx = 1.f;
  if (c->prec == VV_PREC_F16 && mx > 0.f && (mx > 16384.f || mx < 1.f / 256.f)) {
    int e; frexpf(mx, &e); sx = ldexpf(1.f, 8 - e);
  }
  if ((rc = set_sx(c, sx))) return rc;
  const int64_t chunk_rows = std::max<int64_t>(1, (64ll << 20) / ((int64_t)F * 4));
  DevTmp<float> stage;
  HIPCHK(stage.alloc((size_t)chunk_rows * F));
  for (int64_t r0 = 0; r0 < n_rows; r0 += chunk_rows) {
    const int64_t nr = std::min(chunk_rows, n_rows - r0);
    HIPCHK(hipMemcpyAsync(stage, rows + r0 * F, (size_t)nr * F * sizeof(float), hipMemcpyHostToDevice, c->stream));
    launch_table_convert(c->prec, stage, c->table + r0 * c->Fp, nr, F, c->Fp, sx, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  HIPCHK(hipGetLastError());
  return VV_OK;
}

int vv_table_synth(vv_ctx* c, uint64_t seed, int64_t n_rows, int32_t F) {
  if (!c) return fail(VV_ERR_ARG, "vv_table_synth: ctx is NULL");
  int rc = table_alloc(c, n_rows, F);
  if (rc) return rc;
  if ((rc = set_sx(c, 1.f))) return rc;
  launch_table_synth(c->prec, c->table, seed, n_rows, F, c->Fp, 1.f, c->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return VV_OK;
}

int vv_table_get(vv_ctx* c, const int32_t* rows, int64_t n, float* out) {
  if (!c || !out || n <= 0) return fail(VV_ERR_ARG, "vv_table_get: bad argument");
  if (!c->table) return fail(VV_ERR_STATE, "vv_table_get: no table");
  VV_ENTER(c);
  DevTmp<int32_t> drows; DevTmp<float> dout;
  if (rows) {
    for (int64_t i = 0; i < n; ++i)
      if (rows[i] < 0 || rows[i] >= c->n_rows) return fail(VV_ERR_ARG, "vv_table_get: row %d out of range", rows[i]);
    HIPCHK(drows.alloc((size_t)n));
    HIPCHK(hipMemcpy(drows, rows, n * sizeof(int32_t), hipMemcpyHostToDevice));
  } else if (n > c->n_rows) return fail(VV_ERR_ARG, "vv_table_get: n > n_rows");
  HIPCHK(dout.alloc((size_t)n * c->F));
  launch_table_read(c->prec, c->table, drows, n, c->F, c->Fp, 1.f / c->sx, dout, c->stream);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, dout, (size_t)n * c->F * sizeof(float), hipMemcpyDeviceToHost));
  return VV_OK;
}

// ------------------------------------------------------------------------------- params -------
int vv_params_set(vv_ctx* c, int32_t D, const float* W, const float* b, const float* hW,
                  const float* hb) {
  if (!c || !W || D <= 0) return fail(VV_ERR_ARG, "vv_params_set: bad argument");
  if (!c->table) return fail(VV_ERR_STATE, "vv_params_set: set the feature table first (defines F)");
  VV_ENTER(c);
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  HIPCHK(hipStreamSynchronize(c->stream));
  const int F = c->F;
  const size_t nW = (size_t)D * F;
  if (D != c->D && c->comm) return fail(VV_ERR_STATE, "vv_params_set: the communicator was sized for D = %d (vv_comm_destroy first)", c->D);
  if (D != c->D) {
    dfree(c->W); dfree(c->b); dfree(c->hW); dfree(c->hb); dfree(c->Wh); dfree(c->grads_own);
    c->W = c->b = c->hW = c->hb = nullptr; c->Wh = nullptr; c->grads = c->grads_own = nullptr;
    c->D = D; c->Dp = (int)round_up(D, D_ALIGN);
#ifdef VV_LAB
    // (lab: VV_LAB_PARAM_ARENA="skew_hW,skew_Wh" bytes -- W, its history and the 16-bit copy in ONE allocation, the second and third a
    // 2-MiB multiple + the given skew behind the first: does the update-in-the-epilogue kernel's speed depend on their relative placement?
    // The arena is never freed: lab processes exit.)
    if (const char* e = getenv("VV_LAB_PARAM_ARENA")) {
      const size_t S = round_up(nW * 4, (size_t)2 << 20);
      const size_t sk1 = (size_t)atoll(e), sk2 = strchr(e, ',') ? (size_t)atoll(strchr(e, ',') + 1) : 0;
      char* arena = nullptr;
      HIPCHK(hipMalloc(&arena, 3 * S + ((size_t)16 << 20)));
      c->W = (float*)arena; c->hW = (float*)(arena + S + sk1); c->Wh = (uint16_t*)(arena + 2 * S + ((size_t)4 << 20) + sk2);
      HIPCHK(hipMalloc(&c->b, D * 4)); HIPCHK(hipMalloc(&c->hb, D * 4));
      fprintf(stderr, "[vv lab] parameter arena: W %p hW %p Wh %p\n", (void*)c->W, (void*)c->hW, (void*)c->Wh);
    } else
#endif
    {
    HIPCHK(hipMalloc(&c->W, nW * 4)); HIPCHK(hipMalloc(&c->hW, nW * 4));
    HIPCHK(hipMalloc(&c->b, D * 4)); HIPCHK(hipMalloc(&c->hb, D * 4));
    HIPCHK(hipMalloc(&c->Wh, (size_t)c->Dp * c->Fp * 2));
    }
    HIPCHK(hipMemset(c->Wh, 0, (size_t)c->Dp * c->Fp * 2));
    HIPCHK(hipMalloc(&c->grads_own, (nW + D) * 4));
    HIPCHK(hipMemset(c->grads_own, 0, (nW + D) * 4));
    c->grads = c->grads_own;
#ifdef VV_LAB
    if (getenv("VV_DEBUG_PTRS")) fprintf(stderr, "[vv ptrs] W %p hW %p Wh %p grads %p (D %d F %d)\n", (void*)c->W, (void*)c->hW, (void*)c->Wh, (void*)c->grads_own, D, c->F);
#endif
    free_batch(c);
  }
  HIPCHK(hipMemcpy(c->W, W, nW * 4, hipMemcpyHostToDevice));
  if (b) HIPCHK(hipMemcpy(c->b, b, D * 4, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(c->b, 0, D * 4));
  if (hW) HIPCHK(hipMemcpy(c->hW, hW, nW * 4, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(c->hW, 0, nW * 4));
  if (hb) HIPCHK(hipMemcpy(c->hb, hb, D * 4, hipMemcpyHostToDevice)); else HIPCHK(hipMemset(c->hb, 0, D * 4));
  // scale for the half copy from max|W|, then convert (a scale update still pending from an earlier SGD step is void)
  c->scale_pending = false;
  c->red_lazy = false; c->grads_stale = false;   // (a gradient nobody asked for goes with the parameters it belonged to)
  // ... and so does a hinted step that was waiting for its vv_apply_update (its half-applied update is overwritten here) and a hint not yet
  // consumed (the context was stuck -- every later vv_forward_backward refused, vv_apply_update failing on !have_fwd)
  c->upd_in_wgrad = false; c->upd_hint = false; c->grads_lost = false;
  HIPCHK(hipMemsetAsync(&c->scales->wmax_bits, 0, sizeof(unsigned), c->stream));
  launch_absmax(c->W, (int64_t)nW, &c->scales->wmax_bits, c->stream);
  launch_scale_update(c->prec, c->scales, nullptr, 0, c->stream);
  launch_w_convert(c->prec, c->W, c->Wh, D, F, c->Dp, c->Fp, c->scales, c->stream);
  c->params_partial = false;                 // (every rank is given the whole matrix)
  // seed the running max for the first SGD step's scale update
  launch_absmax(c->W, (int64_t)nW, &c->scales->wmax_bits, c->stream);
  c->wmax_seed_live = true;
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  c->have_fwd = false;
  return VV_OK;
}

// Sharded update: every rank holds the fp32 master rows and the history of ITS shard only.  Gathers them -- a COLLECTIVE: every rank of
// the communicator makes the call that leads here (vv_params_get as the facade's Snapshot does, vv_comm_schedule, vv_comm_destroy): one
// grouped all-gather of W, hW and hb (b is complete already).
static int gather_params(vv_ctx* c) {
  if (!c->params_partial || !c->comm) { c->params_partial = false; return VV_OK; }
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  const int world = vv::comm_world(c->comm), rps = c->D / world;
  void* bufs[3] = {c->W, c->hW, c->hb};
  const size_t sbytes[3] = {(size_t)rps * c->F * 4, (size_t)rps * c->F * 4, (size_t)rps * 4};
  HIPCHK(hipEventRecord(c->ev_chunk, c->stream));
  HIPCHK(hipStreamWaitEvent(vv::comm_stream(c->comm), c->ev_chunk, 0));
  if (vv::comm_allgather(c->comm, bufs, sbytes, 3, 1)) return fail(VV_ERR_HIP, "all-gather of the parameters: %s", vv::comm_error(c->comm));
  if (vv::comm_record_done(c->comm)) return fail(VV_ERR_HIP, "all-gather of the parameters: %s", vv::comm_error(c->comm));
  HIPCHK(hipStreamWaitEvent(c->stream, vv::comm_done_event(c->comm), 0));
  c->params_partial = false;
  return VV_OK;
}

int vv_params_get(vv_ctx* c, float* W, float* b, float* hW, float* hb) {
  if (!c) return fail(VV_ERR_ARG, "vv_params_get: ctx is NULL");
  { const int rcg = upd_pending_guard(c, "vv_params_get"); if (rcg) return rcg; }
  if (!c->W) return fail(VV_ERR_STATE, "vv_params_get: no parameters");
  VV_ENTER(c);
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  { const int rcg = gather_params(c); if (rcg) return rcg; }
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->comm && vv::comm_failed(c->comm)) return fail(VV_ERR_HIP, "vv_params_get: the data-parallel exchange failed: %s", vv::comm_error(c->comm));   // (what the update in flight ran into)
  const size_t nW = (size_t)c->D * c->F;
  if (W) HIPCHK(hipMemcpy(W, c->W, nW * 4, hipMemcpyDeviceToHost));
  if (b) HIPCHK(hipMemcpy(b, c->b, c->D * 4, hipMemcpyDeviceToHost));
  if (hW) HIPCHK(hipMemcpy(hW, c->hW, nW * 4, hipMemcpyDeviceToHost));
  if (hb) HIPCHK(hipMemcpy(hb, c->hb, c->D * 4, hipMemcpyDeviceToHost));
  return VV_OK;
}

// ------------------------------------------------------------------------------- step ---------
static int ensure_batch(vv_ctx* c, int B, int C, int Nn) {
  if (B == c->B && C == c->C && Nn == c->Nn && c->H) return VV_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  free_batch(c);
  const int CN = C + Nn;
  const int64_t R64 = (int64_t)B * CN;
  if (R64 > (1ll << 30)) return fail(VV_ERR_ARG, "batch too large");
  c->B = B; c->C = C; c->Nn = Nn; c->R = (int)R64; c->Rp = (int)round_up(R64, R_ALIGN);
  const int D = c->D;
  HIPCHK(hipMalloc(&c->idx_dev, (size_t)c->R * 4));
  HIPCHK(hipMalloc(&c->rows, (size_t)c->Rp * 4));
  HIPCHK(hipMalloc(&c->H, (size_t)c->R * D * 4));
  // + BK rows of slack: the phase-staggered weight-gradient kernel may read one padded K-tile past the last row
  HIPCHK(hipMalloc(&c->dYh, (size_t)(c->Rp + BK) * c->Dp * 2));
  HIPCHK(hipMemset(c->dYh, 0, (size_t)(c->Rp + BK) * c->Dp * 2));     // padding rows / columns stay zero
  HIPCHK(hipMalloc(&c->dbp, (size_t)B * D * 4));
  HIPCHK(hipMalloc(&c->loss_part, (size_t)B * 4));
  HIPCHK(hipMalloc(&c->viol_part, (size_t)B * 4));
  HIPCHK(hipMalloc(&c->s_true, (size_t)B * 4));
  HIPCHK(hipMalloc(&c->s_bogus, (size_t)B * std::max(Nn, 1) * 4));
  HIPCHK(hipMalloc(&c->coeff, (size_t)std::max(C - 1, 1) * 4));
  HIPCHK(hipMalloc(&c->item_w, (size_t)B * 4));
  // split-K so that tiles * S is about one wave of workgroups over the 256 CUs
  const int tiles = (c->Dp / BM) * (c->Fp / BN);
  const int total_steps = c->Rp / BK;
  int S = (256 + tiles - 1) / tiles;
  if (S > total_steps) S = total_steps;
  if (S < 1) S = 1;
  c->S = S; c->kps = (total_steps + S - 1) / S;
  c->slab_bytes = (size_t)S * (size_t)slab_pitch(c->Dp, c->Fp) * 4;
  HIPCHK(hipMalloc(&c->slabs, c->slab_bytes));
  HIPCHK(hipMalloc(&c->slab_sc, (size_t)8 * tiles * sizeof(float)));
  HIPCHK(hipMemset(c->slab_sc, 0, (size_t)8 * tiles * sizeof(float)));
  // de-duplication work arrays
  c->dd_agg_stride = c->R / 1024 + 2;
  HIPCHK(hipMalloc(&c->dd_agg, (size_t)2 * c->dd_agg_stride * sizeof(unsigned long long)));
  HIPCHK(hipMemset(c->dd_agg, 0, (size_t)2 * c->dd_agg_stride * sizeof(unsigned long long)));
  for (int i = 0; i < vv_ctx::kDdSets; ++i) {
    vv_ctx::DdSet& d = c->dd_set[i];
    HIPCHK(hipMalloc(&d.rows, (size_t)c->Rp * 4));
    HIPCHK(hipMalloc(&d.slot_of, (size_t)c->Rp * 4));
    HIPCHK(hipMalloc(&d.uniq, (size_t)c->Rp * 4));
    HIPCHK(hipMalloc(&d.map, (size_t)c->Rp * 4));
    HIPCHK(hipMalloc(&d.ord, (size_t)c->Rp * 4));
    HIPCHK(hipMalloc(&d.cnt, (size_t)c->Rp * 4));
    HIPCHK(hipMalloc(&d.seg, (size_t)(c->Rp + 1) * 4));
  }
  HIPCHK(hipMalloc(&c->dd_pos, (size_t)c->Rp * 4));
  HIPCHK(hipMalloc(&c->dYu, (size_t)(c->Rp + BK) * c->Dp * 2));
  HIPCHK(hipMemset(c->dYu, 0, (size_t)(c->Rp + BK) * c->Dp * 2));
  HIPCHK(hipMalloc(&c->segV, (size_t)2 * B * D * 4));
  HIPCHK(hipMalloc(&c->seg_rec, (size_t)c->Rp * sizeof(SegRec)));
  HIPCHK(hipMalloc(&c->seg_dbp, (size_t)SEGB_BLOCKS * D * 4));
  c->gg_nslot = std::max(std::max(B, SEGB_BLOCKS), (c->Rp + 3) / 4);
  HIPCHK(hipMalloc(&c->gg_slots, (size_t)3 * 2 * c->gg_nslot * 4));
  HIPCHK(hipMemset(c->gg_slots, 0, (size_t)3 * 2 * c->gg_nslot * 4));
  *c->U_host = 0;
  return VV_OK;
}

static int check_cfg(vv_ctx* c, const vv_step_cfg* cfg) {
  if (!c || !cfg) return fail(VV_ERR_ARG, "NULL ctx / cfg");
  if (!c->table || !c->W) return fail(VV_ERR_STATE, "table and parameters must be set first");
  if (cfg->B < 1) return fail(VV_ERR_ARG, "batch_size must be >= 1 (video_sampled_shots_data_layer.cpp:209)");
  if (cfg->C < 2) return fail(VV_ERR_ARG, "context_size must be >= 2 (video_sampled_shots_data_layer.cpp:207)");
  if (cfg->Nn < 1) return fail(VV_ERR_ARG, "num_negative_samples must be >= 1 for the ranking loss");
  if (cfg->norm != VV_NORM_L1 && cfg->norm != VV_NORM_L2) return fail(VV_ERR_ARG, "Unknown Norm (max_margin_loss_layer.cpp:120)");
  if (cfg->dropout_ratio < 0.f || cfg->dropout_ratio >= 1.f) return fail(VV_ERR_ARG, "dropout_ratio must be in [0,1)");
  if (cfg->reg != VV_REG_L1 && cfg->reg != VV_REG_L2) return fail(VV_ERR_ARG, "Unknown regularization type (solver.cpp:523)");
  if (cfg->solver_type < VV_SOLVER_SGD || cfg->solver_type > VV_SOLVER_ADAGRAD) return fail(VV_ERR_ARG, "Unknown SolverType (solver.hpp:141)");
  if (cfg->solver_type == VV_SOLVER_ADAGRAD && cfg->momentum != 0.f) return fail(VV_ERR_ARG, "Momentum cannot be used with AdaGrad. (solver.hpp:121-122)");
  if (cfg->ip_regularization < 0.f) return fail(VV_ERR_ARG, "ip_regularization must be >= 0");
  return VV_OK;
}

// Index batches reach the GPU through kStage pinned, device-mapped buffers that the step's first kernel reads in place
// (see vv_forward_backward_ring).  Returns the slot to fill; waits until the step that last read it is past its index
// kernels (the forward GEMM stamps seq_host).
static int stage_acquire(vv_ctx* c, size_t bytes, int* slot) {
  if (bytes != c->stage_bytes) {
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < vv_ctx::kStage; ++i) {
      if (c->stage_host[i]) (void)hipHostFree(c->stage_host[i]);
      c->stage_host[i] = c->stage_dev[i] = nullptr; c->stage_seq[i] = 0;
      HIPCHK(hipHostMalloc((void**)&c->stage_host[i], bytes, hipHostMallocMapped));
      HIPCHK(hipHostGetDevicePointer((void**)&c->stage_dev[i], c->stage_host[i], 0));
    }
    c->stage_bytes = bytes;
  }
  const int sl = c->stage_next;
  c->stage_next = (c->stage_next + 1) % vv_ctx::kStage;
  const double tw0 = host_now_ms();
  for (unsigned spins = 0; (int32_t)(__atomic_load_n(c->seq_host, __ATOMIC_ACQUIRE) - c->stage_seq[sl]) < 0; ++spins) {
    if (spins > 4096) { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
    if (hipStreamQuery(c->stream) == hipSuccess) break;     // nothing queued any more: every earlier step is done
  }
  c->wait_ms[3] += host_now_ms() - tw0;
  *slot = sl;
  return VV_OK;
}

static void flush_scale_update(vv_ctx* c) {
  if (!c->scale_pending) return;
  launch_scale_update(c->prec, c->scales, c->wmax_blocks + c->wmax_cur * WMAX_SLOTS, c->wmax_n, c->stream);
  c->scale_pending = false;
  c->wmax_seed_live = false;               // (the scale update folds and clears Scales::wmax_bits)
}

// The reduction vv_forward_backward left undone (vv_ctx::red_lazy): run it now, as the plain k_reduce -- somebody reads the
// gradient or the loss before an update.
static int reduce_now(vv_ctx* c) {
  if (c->upd_in_wgrad) {
    // (vv_loss_get between the backward pass and vv_apply_update of a step announced by vv_update_hint: the loss scalars now, from the
    // partials; the bias update and the bookkeeping stay with vv_apply_update)
    ReduceArgs ra = c->red_args;
    ra.parts = 2; ra.scale_sc = nullptr; ra.gmax_host = nullptr;
    launch_reduce(ra, c->stream);
    HIPCHK(hipGetLastError());
    return VV_OK;
  }
  if (c->grads_stale) {
    // the fused update consumed the slabs without writing dW out; they are untouched since: reduce them now (dW only --
    // db, the loss and the guard's report were produced by the fused launch)
    c->grads_stale = false;
    ReduceArgs ra = c->red_args;
    ra.parts = 1; ra.scale_sc = nullptr; ra.gmax_host = nullptr;
    launch_reduce(ra, c->stream);
    HIPCHK(hipGetLastError());
  }
  if (!c->red_lazy) return VV_OK;
  c->red_lazy = false;
  ReduceArgs ra = c->red_args;
  if (c->scale_pending) {
    ra.scale_sc = c->scales; ra.scale_wmax = c->wmax_blocks + c->wmax_cur * WMAX_SLOTS; ra.scale_n = c->wmax_n; ra.scale_prec = c->prec;
    c->scale_pending = false; c->wmax_seed_live = false;
  }
  PROFILED(c, "reduce", launch_reduce(ra, c->stream));
  HIPCHK(hipGetLastError());
  return VV_OK;
}

// F-chunks of the overlapped update: n column blocks of W whose widths fall off geometrically (ratio 0.6: e.g. 51 / 31 / 18 %
// of F for three).  The exchange is slower than the forward GEMM consumes W, so the GEMM always ends up waiting for the
// LAST chunk and then still has that chunk's K-tiles to run: a small last chunk shortens that tail, a large first chunk
// keeps the number of collectives (each with its launch latency) down.  Boundaries are even K-tiles (the gate sits in
// front of phase 3 of an even K-tile), at least 4 apart; fills c->chunk_kt[0 .. n], returns n (1 when F is too short).
static int chunk_plan(vv_ctx* c) {
  const int nk = c->Fp / BK;
  int n = std::max(1, std::min(c->n_chunks, nk / 8));
  for (;;) {
    double den = 0, r = 1;
    for (int i = 0; i < n; ++i) { den += r; r *= 0.6; }
    c->chunk_kt[0] = 0;
    double acc = 0; r = 1;
    bool ok = true;
    for (int i = 1; i < n; ++i) {
      acc += r; r *= 0.6;
      int kt = (int)(nk * acc / den + 0.5) & ~1;
      if (kt < c->chunk_kt[i - 1] + 4 || kt > nk - 4) ok = false;
      c->chunk_kt[i] = kt;
    }
    c->chunk_kt[n] = nk;
    if (ok || n == 1) return n;
    --n;
  }
}

// Joins the communication stream: after this, everything the library queued there -- the F-chunks of an overlapped
// update: all-reduce, SGD, publish -- is ordered before whatever the compute stream is given next.  Every entry point
// that reads or writes the parameters, their half copy, the momentum or the gradient buffer goes through here first;
// the one reader that does not is the next step's forward GEMM, which waits chunk by chunk inside the kernel instead.
extern "C++" int vv_comm_join(vv_ctx* c) {
  // upd_unjoined: the gated forward GEMM has consumed the update chunk by chunk, but nothing on the compute stream waits for
  // the END of the update's kernels (their plain stores -- W, the history -- are released at that end, not at the gate)
  if (c->comm && (c->upd_inflight || c->upd_unjoined)) {
    HIPCHK(hipStreamWaitEvent(c->stream, vv::comm_done_event(c->comm), 0));
    c->upd_inflight = c->upd_unjoined = false;
  }
  if (c->gate_err && *(volatile int32_t*)c->gate_err) {
    *c->gate_err = 0;
    return fail(VV_ERR_STATE, "a forward pass gave up waiting for the overlapped parameter update (a rank failed, or the communication stream stalled)");
  }
  return VV_OK;
}

// The grouping of one batch (k_dd_claim .. k_dd_segstart) into the next of the rotating output sets, on the grouping
// stream; *set_out = the set.  The step that consumes the set binds it and orders its stream behind set.done (fb_impl).
static int dd_issue(vv_ctx* c, const int32_t* didx, int idx_on_device, int64_t row_limit, int32_t seq, int* set_out) {
  hipStream_t s = c->stream;
  const int D = c->D;
  {
    // The grouping kernels need only the indices: they run on the context's second stream, so that when the host is a
    // step ahead (the normal case: nothing in the loop waits for the GPU) the grouping of step k+1 executes beside the
    // kernels of step k and the step's own stream merely waits for an event that has long been signalled.  Three sets of
    // output arrays rotate; a set is rewritten only after the step that read it has issued its last reader.
    hipStream_t ds = c->dd_async ? c->dd_stream : s;
    if (c->dd_async && idx_on_device == 1) {
      // the caller's indices may still be in the making on the context's stream: the grouping stream waits for it
      if (!c->ev_idx) HIPCHK(hipEventCreateWithFlags(&c->ev_idx, hipEventDisableTiming));
      HIPCHK(hipEventRecord(c->ev_idx, s));
      HIPCHK(hipStreamWaitEvent(ds, c->ev_idx, 0));
    }
    const int64_t need = c->n_rows + 1 + c->patch_cap;
    if (need > c->dd_key_cap) {
      HIPCHK(hipStreamSynchronize(s));
      HIPCHK(hipStreamSynchronize(c->dd_stream));
      dfree(c->dd_key); c->dd_key = nullptr;
      HIPCHK(hipMalloc(&c->dd_key, (size_t)need * sizeof(unsigned long long)));
      HIPCHK(hipMemsetAsync(c->dd_key, 0, (size_t)need * sizeof(unsigned long long), ds));
      c->dd_key_cap = need;
    }
    if (++c->dd_epoch == 0) {        // epoch tags wrapped: start over with clean tag words
      HIPCHK(hipMemsetAsync(c->dd_key, 0, (size_t)c->dd_key_cap * sizeof(unsigned long long), ds));
      HIPCHK(hipMemsetAsync(c->dd_agg, 0, (size_t)2 * c->dd_agg_stride * sizeof(unsigned long long), ds));
      c->dd_epoch = 1;
    }
    const int si = (int)(c->dd_step++ % vv_ctx::kDdSets);
    vv_ctx::DdSet& set = c->dd_set[si];
    *set_out = si;
    if (c->dd_async && set.used_seq) {
      // The set was last read by the step with sequence number used_seq.  The step's stream runs its kernels in order, so
      // once a kernel of ANY later step has stamped its number, every kernel of that step has finished.  The host waits
      // for that stamp (normally long there: it bounds how far the host runs ahead to kDdSets - 1 steps) -- an event
      // recorded per step for the same purpose cost ~6 us of stream time each (a queue barrier packet between two kernels).
      // In steady state this wait is what paces the host, so the grouping it queues next starts right at the stamp and
      // shares the chip with the kernel that wrote it.  Two stamps exist: the forward GEMM's (word 0, written as it starts)
      // and the score kernel's (word 1, i.e. "the forward GEMM has finished").  A/B on one box, 4 x 4000 steps each
      // (profiles/r02_step_ablations.txt): released by the forward GEMM the grouping costs that GEMM 5-6 us (0.085 against
      // 0.080 ms alone: one persistent workgroup per CU, and a CU that also hosts grouping workgroups finishes late) and the
      // step takes 0.2305 ms; released by the score kernel it costs the score and segment kernels 4.5 + 2 us and the step
      // takes 0.2337 ms.  Released after the segment kernel or later it runs into the weight-gradient GEMM, is starved there
      // (80 us instead of 33) and the next step waits for it (0.239-0.253 ms).  Default: the forward GEMM's stamp -- the
      // faster step, at the price of a forward-GEMM duration (and roofline fraction) that includes the co-running kernels.
      // VV_DEDUP_GATE=1 selects the other.
      const int gate_word = c->dd_gate_word;
      const double tw0 = host_now_ms();
      for (unsigned spins = 0; (int32_t)(__atomic_load_n(c->seq_host + gate_word, __ATOMIC_ACQUIRE) - set.used_seq) <= 0; ++spins) {
        if (spins > 4096) { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
        if (hipStreamQuery(s) == hipSuccess) break;             // nothing queued any more: every earlier step is done
      }
      c->wait_ms[1] += host_now_ms() - tw0;
    }
    set.used_seq = seq;
    DedupArgs da;
    da.idx = didx; da.rows = set.rows; da.u_host = c->U_host_dev; da.key = c->dd_key; da.agg = c->dd_agg; da.agg_stride = c->dd_agg_stride;
    da.slot_of = set.slot_of; da.uniq_rows = set.uniq; da.map = set.map; da.ord = set.ord; da.cnt = set.cnt;
    da.seg_start = set.seg; da.pos = c->dd_pos; da.info = set.info; da.tickets = set.info + 2;
    {
      // placement of the grouping workgroups beside the forward GEMM (kernels_dedup.hip): only when that GEMM -- sized for
      // the previous step's distinct-row count, as launch_fwd_gemm will size it -- leaves at least 24 CUs idle
      const int lds_kb = c->dd_lds_kb;
      const int hint = *(volatile int32_t*)c->U_host;
      const long tiles = fwd_gemm_plan(c->R, hint, D, nullptr);
      da.lds_bytes = lds_kb >= 0 ? lds_kb * 1024 : (c->dd_async && gemm_variant() == 5 && hint > 0 && tiles <= c->n_cu - 16 ? 36 * 1024 : 0);
    }
    da.R = c->R; da.Rp = c->Rp; da.zero_row = (int32_t)c->n_rows; da.row_limit = (int32_t)row_limit; da.epoch = c->dd_epoch;
    PROFILED(c, "dedup", (launch_dedup(da, ds), launch_dedup_groups(da, ds)));
    if (c->dd_async) HIPCHK(hipEventRecord(set.done, ds));
  }
  return VV_OK;
}

// idx_on_device: 0 host indices; 1 device indices produced on the context's stream (ordered after everything queued
// there); 2 device indices that are complete already (no ordering needed: the ring's staging slots, static batches)
// vv_update_hint: the parameter matrix of this step was updated inside the weight-gradient GEMM; until vv_apply_update has finished the step
// (bias, loss, the scale bookkeeping) the parameters are half-way -- only vv_apply_update and vv_loss_get may come next
static int upd_pending_guard(vv_ctx* c, const char* who) {
  if (c && c->grads_lost && !strncmp(who, "vv_grads", 8))
    return fail(VV_ERR_STATE, "%s: the last step was announced by vv_update_hint and applied its update where the gradient was produced: "
                              "that gradient was never stored (run the step without the hint to read it)", who);
  if (c && c->upd_in_wgrad)
    return fail(VV_ERR_STATE, "%s: the step announced by vv_update_hint has applied its update where the gradient was produced -- call vv_apply_update "
                              "first (the gradient of such a step is not kept; without the hint every call is allowed as before)", who);
  return VV_OK;
}

// "One hint covers one step": the vv_forward_backward* entry point that follows vv_update_hint consumes it even when it returns early (bad
// arguments, no batch in the ring, a failed allocation) -- a hint left standing would apply an update, with a stale rate, in some later,
// unrelated backward pass.  fb_impl takes the hint first; this drops what is left when the wrapper never got there.
struct HintScope {
  vv_ctx* c;
  explicit HintScope(vv_ctx* c_) : c(c_) {}
  ~HintScope() { if (c) c->upd_hint = false; }
};

int vv_update_hint(vv_ctx* c, const vv_step_cfg* cfg) {
  int rc = check_cfg(c, cfg);
  if (rc) return rc;
  c->upd_hint = true;
  c->upd_cfg = *cfg;
  return VV_OK;
}

static int fb_impl(vv_ctx* c, const vv_step_cfg* cfg, const int32_t* idx, int idx_on_device, int64_t row_limit, int32_t seq = 0) {
  { const int rcg = upd_pending_guard(c, "vv_forward_backward"); if (rcg) return rcg; }
  const bool upd_hint = c && c->upd_hint;       // (consumed by this call whatever path it takes)
  if (c) c->upd_hint = false;
  int rc = check_cfg(c, cfg);
  if (rc) return rc;
  if (!idx) return fail(VV_ERR_ARG, "vv_forward_backward: idx is NULL");
  VV_ENTER(c);
  // (an overlapped update of the previous step may still be arriving: see the forward GEMM below)
  if (c->upd_inflight && (cfg->B != c->B || cfg->C != c->C || cfg->Nn != c->Nn || !c->H) && (rc = comm_join(c))) return rc;
  if ((rc = ensure_batch(c, cfg->B, cfg->C, cfg->Nn))) return rc;
  const int B = c->B, C = c->C, Nn = c->Nn, CN = C + Nn, D = c->D;
  hipStream_t s = c->stream;

  const int32_t* didx = idx;
  if (!idx_on_device) {
    // host indices: checked while they are copied into a pinned staging slot that the first kernel reads in place
    int sl = 0;
    if ((rc = stage_acquire(c, (size_t)c->R * 4, &sl))) return rc;
    int32_t* dst = c->stage_host[sl];
    int bad = -1;
    for (int i = 0; i < c->R; ++i) { const int32_t v = idx[i]; dst[i] = v; if (v < -1 || v >= row_limit) bad = i; }
    if (bad >= 0) return fail(VV_ERR_ARG, "idx[%d] = %d out of range [-1, %lld)", bad, idx[bad], (long long)c->n_rows);
    if (!seq) seq = ++c->step_seq;
    c->stage_seq[sl] = seq;
    didx = c->stage_dev[sl];
  }
  if (!seq) seq = ++c->step_seq;          // every step has a sequence number; its forward GEMM stamps it into host-visible memory

  // eltwise coefficients (cached on the device until they change)
  std::vector<float> coeff(C - 1);
  for (int j = 0; j < C - 1; ++j) coeff[j] = cfg->ctx_coeff ? cfg->ctx_coeff[j] : 1.0f / (C - 1);
  if (coeff != c->coeff_host) {
    HIPCHK(hipMemcpyAsync(c->coeff, coeff.data(), coeff.size() * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    c->coeff_host = coeff;
  }
  if (cfg->item_weight) {
    for (int i = 0; i < B; ++i)
      if (!(cfg->item_weight[i] >= 0.f)) return fail(VV_ERR_ARG, "item_weight[%d] = %g: All weights should be greater than 0 (max_margin_loss_layer.cpp:34)", i, (double)cfg->item_weight[i]);
    HIPCHK(hipMemcpyAsync(c->item_w, cfg->item_weight, (size_t)B * 4, hipMemcpyHostToDevice, s));
  }
  if (cfg->dropout_ratio > 0.f && cfg->dropout_mask) {
    const size_t nb = (size_t)c->R * D;
    if (nb > c->mask_bytes) { dfree(c->mask); c->mask = nullptr; HIPCHK(hipMalloc(&c->mask, nb)); c->mask_bytes = nb; }
    HIPCHK(hipMemcpyAsync(c->mask, cfg->dropout_mask, nb, hipMemcpyHostToDevice, s));
  }

  // De-duplicate the batch rows.  Dropout sits BEHIND the projection (fc7 -> ReLU -> drop2, mednet_embedding_train.prototxt:190-230): the
  // projection of equal rows is equal, only the mask differs per instance.  Where the segment-wise pair carries the masks (k_score_fwd +
  // k_seg_bwd at D = 512: every instance masks its row as it reads it, the backward sums m_i-weighted terms per distinct row) dropout
  // rides the de-duplicated path; on the other shapes it stays dense (the mask in the forward GEMM's epilogue).  The two executions
  // evaluate the same mask function (vv_internal.h: DropSpec).  Needs the default two-buffer GEMM kernels.
  const bool drop_on = cfg->dropout_ratio > 0.f;
  const bool drop_dd = drop_on && c->drop_dedup && c->seg_bwd && score_fwd_dropout_supported(D, C, Nn);
  const bool dd = c->dedup && (!drop_on || drop_dd) && (gemm_variant() == 0 || (gemm_variant() >= 5 && gemm_variant() <= 8)) && !ablate_on();
  DropSpec dsp;
  if (drop_on) {
    dsp.mode = cfg->dropout_mask ? 2 : 1;
    dsp.thr = (uint32_t)(cfg->dropout_ratio * 65536.f + 0.5f);
    dsp.s32 = (uint32_t)(mix64(cfg->dropout_seed * 0x9E3779B97F4A7C15ull + c->iter, 0x5eedull) >> 32);
    dsp.scale = 1.f / (1.f - cfg->dropout_ratio);
    dsp.mask = cfg->dropout_mask ? c->mask : nullptr;
    dsp.B = B; dsp.CN = CN; dsp.D = D;
  }
  c->last_drop = (dd && drop_on) ? dsp : DropSpec();
  c->last_dedup = dd;
  if (!dd) launch_map_rows(didx, c->rows, c->R, c->Rp, (int32_t)c->n_rows, (int32_t)row_limit, s);
  if (dd) {
    int si = 0;
    if ((rc = dd_issue(c, didx, idx_on_device, row_limit, seq, &si))) return rc;
    vv_ctx::DdSet& set = c->dd_set[si];
    set.used_seq = seq;
    c->dd_rows = set.rows; c->dd_slot_of = set.slot_of; c->dd_uniq = set.uniq; c->dd_map = set.map; c->dd_ord = set.ord;
    c->dd_cnt = set.cnt; c->dd_seg = set.seg; c->dd_info = set.info;
    if (c->dd_async) {
      // The grouping takes ~30 us on an idle second stream and the host is normally several steps ahead of the GPU: it
      // can afford to watch the event for a moment.  Once the event has fired nothing needs to be put into the step's
      // stream at all (a wait packet in front of the forward GEMM costs ~6 us of stream time even when already satisfied);
      // otherwise the stream waits as usual.
      bool fired = hipEventQuery(set.done) == hipSuccess;
      // (an idle step stream -- the first step after a synchronisation -- means a grouping just queued cannot have run yet:
      // watching it would hold the host for its whole 40 us while the GPU then waits for the step's launches; queue the wait)
      if (!fired && c->dd_spin_us > 0 && hipStreamQuery(s) != hipSuccess) {
        const double t0 = host_now_ms();
        do { fired = hipEventQuery(set.done) == hipSuccess; } while (!fired && (host_now_ms() - t0) * 1e3 < c->dd_spin_us);
        c->wait_ms[2] += host_now_ms() - t0;
      }
      if (!fired) HIPCHK(hipStreamWaitEvent(s, set.done, 0));
    }
  }

  FwdArgs fa;
  fa.table = c->table; fa.rows = dd ? c->dd_uniq : c->rows; fa.Wh = c->Wh; fa.bias = c->b; fa.scales = c->scales;
  fa.H = c->H; fa.R = c->R; fa.D = D; fa.Fp = c->Fp; fa.relu = 1; fa.zero_row = (int32_t)c->n_rows;
  fa.n_dev = dd ? c->dd_info : nullptr;
  fa.R_hint = dd ? *(volatile int32_t*)c->U_host : 0;
  fa.seq_host = c->seq_host_dev; fa.seq = seq;

  // ip2 as f16 (option "h16"): only where the segment-wise pair reads it -- de-duplicated batches of D = 512 / 1024 (k_score_fwd / k_score_stream /
  // k_seg_bwd carry the f16 row loads; the dense and the generic kernels keep fp32 rows), D % 8 == 0, the phase-staggered forward kernel
  const bool seg_path = dd && c->seg_bwd && (D == 512 || D == 1024);
  const bool h16 = c->h16 && seg_path && gemm_variant() == 5 && !ablate_on();
  fa.h16 = h16 ? 1 : 0;
  c->last_h16 = h16;
  fa.drop_ratio = (dd && drop_on) ? 0.f : cfg->dropout_ratio;      // (de-duplicated: H holds the shared pre-dropout rows, the instances mask them)
  fa.mask = (!(dd && drop_on) && cfg->dropout_ratio > 0.f && cfg->dropout_mask) ? c->mask : nullptr;
  fa.drop_seed = cfg->dropout_seed * 0x9E3779B97F4A7C15ull + c->iter;
  fa.B = B; fa.CN = CN;
  if (c->upd_inflight) {
    // Data-parallel overlap: the previous step's update is still arriving on the communication stream, F-chunk by
    // F-chunk.  The forward GEMM starts anyway and waits per chunk inside the kernel -- provided its persistent
    // workgroups leave compute units free for the update's own kernels (all-reduce, SGD: they must be able to run
    // BESIDE the waiting GEMM, or nothing would ever release it); otherwise the stream joins here, as without overlap.
    const long tiles = fwd_gemm_plan(c->R, dd ? fa.R_hint : 0, D, nullptr);
    const bool no_gate = !c->comm_gate;
    if (!no_gate && gemm_variant() == 5 && !ablate_on() && fwd_gemm_can_gate(fa) && (!dd || fa.R_hint > 0) && tiles <= c->n_cu - 16) {
      fa.gate = c->w_gate; fa.gate_seq = c->upd_seq; fa.gate_err = c->gate_err_dev;
      if (c->grads_sharded) { fa.gate_n = 1; fa.gate_kt[0] = 0; fa.gate_kt[1] = c->Fp / BK; }        // the sharded update publishes once
      else { fa.gate_n = chunk_plan(c); for (int i = 0; i <= fa.gate_n; ++i) fa.gate_kt[i] = c->chunk_kt[i]; }
      c->upd_inflight = false;              // whatever follows the forward GEMM on this stream follows every PUBLISHED store of the update
      c->upd_unjoined = true;               // ... its plain stores (W, the history) only behind the event: vv_comm_join for whoever reads those
    } else if ((rc = comm_join(c))) return rc;
  }
  PROFILED(c, "fwd_gemm", launch_fwd_gemm(c->prec, fa, s));

  const int64_t count = (int64_t)B * Nn;
  const int64_t gcount = cfg->global_count > 0 ? cfg->global_count : count;
  // half-precision gradient scale: a power of two near the loss count keeps dY*sg around 1
  int e; frexpf((float)gcount, &e);
  // The segment-wise backward then settles the scale on the device, from a bound on the step's own gradients, before it
  // rounds anything (GuardArgs::proactive).  On the other paths the host's scale follows the data: every step reports its
  // largest |dY| (k_reduce, a host-visible ring); when the max of step seq - 4 -- a FIXED lag, so that the scale sequence
  // does not depend on host timing and runs stay bit-reproducible -- lies outside [2^5, 2^13) in the current scaled units,
  // the scale moves so that it lies in [2^9, 2^10).  That only keeps the guard's repeats rare: a value past f16's range never
  // reaches the weight-gradient GEMM either way (GradGuard).
  if (c->prec == VV_PREC_F16) {
    constexpr int kLag = 4;
    if (c->gg_seq0 == 0) c->gg_seq0 = seq;
    if (seq - c->gg_seq0 >= kLag) {
      const int32_t want = seq - kLag;
      volatile unsigned long long* en = c->gmax_host + 2 * (want & 15);
      bool have = false;
      const double tw0 = host_now_ms();
      for (unsigned spins = 0; !(have = (int32_t)(uint32_t)__atomic_load_n(en, __ATOMIC_ACQUIRE) == want); ++spins) {
        if (spins > 4096) { timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }
        if (hipStreamQuery(s) == hipSuccess) { have = (int32_t)(uint32_t)__atomic_load_n(en, __ATOMIC_ACQUIRE) == want; break; }
      }
      c->wait_ms[0] += host_now_ms() - tw0;
      if (c->trace_waits && ++c->wait_calls % 100 == 0) {
        fprintf(stderr, "[vv waits] per step over the last 100: gradient-scale report %.3f, grouping-set gate %.3f, grouping event %.3f, staging slot %.3f ms\n",
                c->wait_ms[0] / 100, c->wait_ms[1] / 100, c->wait_ms[2] / 100, c->wait_ms[3] / 100);
        c->wait_ms[0] = c->wait_ms[1] = c->wait_ms[2] = c->wait_ms[3] = 0;
      }
      if (have) {
        const uint32_t bits = (uint32_t)(en[0] >> 32), gbits = (uint32_t)(en[1] >> 32);
        float gmax, gbound; memcpy(&gmax, &bits, 4); memcpy(&gbound, &gbits, 4);
        const uint32_t rep = (uint32_t)en[1];
        if (rep & (1u << 30)) return fail(VV_ERR_STATE, "internal error: a 16-bit gradient value passed f16's range in step %d although the "
                                                        "gradient-scale guard was active (GradGuard)", (int)want);
        if (rep & (1u << 16)) ++c->gg_repeats;
        if (gbound > 0.f) {
          // segment-wise path: nothing to steer -- k_seg_bwd settles its scale on the device from the step's own bound
        } else if (gmax > 0.f && std::isfinite(gmax)) {
          int eu; (void)frexpf(gmax, &eu);
          const int ex = eu + e + c->sg_adj;            // exponent of the max in the current scaled units
          if (ex > 13 || ex < 5) c->sg_adj += 10 - ex;
        }
      }
    }
    c->sg_adj = std::max(-100 - e, std::min(100 - e, c->sg_adj));
  }
  c->sg = c->prec == VV_PREC_F16 ? ldexpf(1.f, e + c->sg_adj) : 1.f;
  c->last_loss_weight = cfg->loss_weight;

  ScoreArgs sa;
  sa.H = c->H; sa.dYh = c->dYh; sa.dbp = c->dbp; sa.loss_part = c->loss_part; sa.viol_part = c->viol_part;
  sa.s_true = c->s_true; sa.s_bogus = c->s_bogus; sa.coeff = c->coeff;
  sa.B = B; sa.C = C; sa.Nn = Nn; sa.D = D; sa.Dp = c->Dp;
  sa.margin = cfg->margin; sa.norm = cfg->norm;
  sa.grad_scale = cfg->loss_weight / (float)gcount;
  sa.drop_scale = cfg->dropout_ratio > 0.f ? 1.f / (1.f - cfg->dropout_ratio) : 1.f;
  sa.sg = c->sg;
  sa.map = dd ? c->dd_map : nullptr; sa.seg_start = dd ? c->dd_seg : nullptr; sa.ord = dd ? c->dd_ord : nullptr;
  sa.item_w = cfg->item_weight ? c->item_w : nullptr;
  sa.gate_host = c->seq_host_dev + 1; sa.gate_seq = seq;
  if (dd && drop_on) sa.drop = dsp;

  // de-duplicated batches of the supported shape: the backward stays factored per instance and is summed per distinct
  // row (k_score_fwd + k_seg_bwd); otherwise per-instance 16-bit gradient rows (+ k_segsum when de-duplicated)
  const bool seg = dd && c->seg_bwd && score_fwd_supported(sa);
  if (seg != seg_path) return fail(VV_ERR_STATE, "internal error: the forward pass was planned for %s rows of ip2, the score kernels expect the other form", h16 ? "f16" : "fp32");
  sa.h16 = h16 ? 1 : 0;
  // the per-item vectors as f16: the one-sweep kernel's D = 1024 form only (840 k gathers of a 4 KB row per step at configs[4]'s per-GPU shape;
  // at D = 512 the 2 MB matrix lives in L2 and 16-bit vectors measured net zero, profiles/r04_h16_intermediate.txt)
  const bool v16 = h16 && c->v16 && D == 1024 && !drop_on;
  sa.v16 = v16 ? 1 : 0;
  c->last_seg_bwd = seg; c->last_score = sa;
  // f16 gradient-scale guard (vv_internal.h: GradGuard): the kernels that round gradients to 16 bits are launched once as
  // usual (round 0) and once more per guard round as conditional repeats -- near-empty launches unless the round before
  // reported a value past f16's range.  Rounds: 1 where one kernel rounds (per-instance rows, or the segment sums); 2 for
  // rows + their sums (k_segsum): sums of clipped rows say nothing about the true sums, which the second round sees.
  GuardArgs gd;
  if (c->prec == VV_PREC_F16) { gd.gg = c->gg; gd.slots = c->gg_slots; gd.nslot = c->gg_nslot; gd.seq = seq; }
  // (the segment-wise backward needs no repeat at all: its score kernel bounds every instance's gradient and k_seg_bwd
  // settles the scale before it rounds anything -- GuardArgs::proactive)
  const bool proactive_on = c->guard_proactive;   // (lab) false: the repeat form on this path too (A/B)
  const bool proactive = seg && proactive_on;
  const int n_rounds = c->prec != VV_PREC_F16 ? 0 : (proactive ? 0 : (dd && !seg ? 2 : 1));
  gd.n_of[0] = seg ? 0 : B;
  gd.n_of[1] = seg ? SEGB_BLOCKS : (dd ? (c->Rp + 3) / 4 : 0);
  SegBwdArgs ba;
  SegsumArgs ga;
  if (seg) {
    sa.V = c->segV; sa.rec = c->seg_rec;
#ifdef VV_LAB
    if (const char* v = lab_env("VV_LAB_SCORE_HACK")) sa.lab_hack = atoi(v);
    if (lab_env("VV_LAB_SCORE_TS")) {
      // (lab) k_score_fwd's phase stamps: one buffer, every step overwrites it; tools/lab/score_ts.py reads it through vv_lab_score_ts
      static thread_local uint32_t* ts_buf = nullptr;
      if (!ts_buf) { HIPCHK(hipMalloc(&ts_buf, (size_t)65536 * 16 * 4)); HIPCHK(hipMemset(ts_buf, 0, (size_t)65536 * 16 * 4)); }
      sa.lab_ts = ts_buf; c->lab_score_ts = ts_buf;
    }
#endif
    if (gd.gg && proactive) { sa.bound_out = c->gg_bound; sa.bound_seq = seq; }
    ba.H = c->H; ba.V = c->segV; ba.rec = c->seg_rec; ba.seg_start = c->dd_seg; ba.info = c->dd_info; ba.dYu = c->dYu;
    ba.dbp = c->seg_dbp; ba.Rp = c->Rp; ba.D = D; ba.Dp = c->Dp; ba.inv_sg = 1.f / c->sg; ba.h16 = h16 ? 1 : 0; ba.v16 = v16 ? 1 : 0;
    if (drop_on) ba.drop = dsp;
  } else if (dd) {
    ga.dYh = c->dYh; ga.seg_start = c->dd_seg; ga.info = c->dd_info; ga.dYu = c->dYu; ga.Rp = c->Rp; ga.Dp = c->Dp;
  }
  for (int round = 0; round <= n_rounds; ++round) {
    gd.round = round; gd.final_round = round == n_rounds;
    if (seg) {
      if (round == 0) PROFILED(c, "score_loss", launch_score_fwd(sa, s));
      ba.guard = gd; ba.guard.producer = 1; ba.guard.last = 1;
      if (gd.gg && proactive) { ba.guard.proactive = 1; ba.guard.bound = c->gg_bound; ba.guard.cnt_max = c->dd_info + 1; }
      if (round == 0) PROFILED(c, "segsum", launch_seg_bwd(c->prec, ba, s));
      else PROFILED(c, "guard", launch_seg_bwd(c->prec, ba, s));
    } else {
      sa.guard = gd; sa.guard.producer = 0; sa.guard.last = dd ? 0 : 1;
      if (round == 0) PROFILED(c, "score_loss", launch_score_loss(c->prec, sa, s));
      else if (!dd) PROFILED(c, "guard", launch_score_loss(c->prec, sa, s));
      else launch_score_loss(c->prec, sa, s);
      if (dd) {
        ga.guard = gd; ga.guard.producer = 1; ga.guard.last = 1;
        if (round == 0) PROFILED(c, "segsum", launch_segsum(c->prec, ga, s));
        else PROFILED(c, "guard", launch_segsum(c->prec, ga, s));
      }
    }
  }
  WgradArgs wa;
  wa.dYh = dd ? c->dYu : c->dYh; wa.table = c->table; wa.rows = dd ? c->dd_uniq : c->rows; wa.slabs = c->slabs;
  wa.Rp = c->Rp; wa.Dp = c->Dp; wa.Fp = c->Fp; wa.S = c->S; wa.ksteps_per_split = c->kps;
  wa.n_dev = dd ? c->dd_info : nullptr; wa.zero_row = (int32_t)c->n_rows;
  {
    const int64_t t_rows = c->n_rows + 1 + c->patch_cap;      // every row a K-tile can name: the table, its zero row, the scratch rows behind it
    wa.lean = c->ko.wgrad_lean && t_rows < (1ll << 24) && (int64_t)c->Fp * 2 < (1ll << 24) && t_rows * c->Fp * 2 + 8192 < (1ll << 32);
  }
  ReduceArgs ra;
  ra.slabs = c->slabs; ra.S = c->S; ra.Dp = c->Dp; ra.Fp = c->Fp; ra.dbp = seg ? c->seg_dbp : c->dbp; ra.B = B;
  ra.db_rows = seg ? SEGB_BLOCKS : 0;
  ra.scales = c->scales; ra.sg = c->sg; ra.grads = c->grads; ra.D = D; ra.F = c->F;
  if (gd.gg) {
    ra.gg = c->gg; ra.gmax_slots = c->gg_slots; ra.gmax_n0 = gd.n_of[0]; ra.gmax_n1 = gd.n_of[1]; ra.gmax_stride = c->gg_nslot;
    ra.gmax_host = c->gmax_host_dev; ra.seq = seq; ra.guard_last_round = n_rounds;
    if (proactive) { ra.gbound = c->gg_bound; ra.gcnt = c->dd_info + 1; }
  }
  ra.ip_scale = cfg->ip_regularization > 0.f ? 1.f + cfg->ip_regularization * 0.5f : 1.f;     // inner_product_layer.cpp:80-90
  ra.loss_part = c->loss_part; ra.viol_part = c->viol_part; ra.loss_scale = cfg->loss_weight / (float)count; ra.loss_out = c->loss2;

  // Data-parallel overlap: the gradient buffer is laid out chunk-major (a few column blocks, each one contiguous
  // all-reduce message) and vv_apply_update runs the update chunk by chunk on the communication stream.
  const int shard_rows = c->comm ? c->D / vv::comm_world(c->comm) : 0;
  const bool sharded = c->comm && c->comm_sharded && c->F % 4 == 0 && c->grads == c->grads_own && !c->grads_exposed &&
                       shard_rows * vv::comm_world(c->comm) == c->D && shard_rows % 4 == 0 && vv::comm_world(c->comm) * (SGD_BLOCKS / vv::comm_world(c->comm)) <= WMAX_SLOTS;
  if (sharded) ra.shard_rows = shard_rows;
  const bool chunked = !sharded && c->comm && c->comm_overlap && c->F % 4 == 0 && c->grads == c->grads_own && !c->grads_exposed;   // (a holder of vv_grads_device's pointer reads the documented flat layout)
  if (chunked) {
    ra.n_chunks = chunk_plan(c);
    for (int i = 0; i <= ra.n_chunks; ++i) ra.chunk_c0[i] = std::min(c->F, c->chunk_kt[i] * BK);
  }
  // Lazy reduction (vv_ctx::red_lazy): with no communicator in the way, dW stays in the slabs until somebody wants it --
  // normally vv_apply_update, which reduces and updates in one launch.  VV_FUSE_UPDATE=0: reduce here, as ever.
  const bool fuse_on = c->fuse_update;
  const bool lazy = fuse_on && !c->comm && !c->grads_exposed && c->grads == c->grads_own && c->F % 4 == 0 && c->S <= 8;
  // the W -> half scale update the previous vv_apply_update left pending rides in this step's reduction launch
  if (!lazy && c->scale_pending) {
    ra.scale_sc = c->scales; ra.scale_wmax = c->wmax_blocks + c->wmax_cur * WMAX_SLOTS; ra.scale_n = c->wmax_n; ra.scale_prec = c->prec;
    c->scale_pending = false; c->wmax_seed_live = false;
  }
  // vv_update_hint + one split of K: the tile in the weight-gradient GEMM's accumulators IS the gradient -- the solver's rule is applied
  // there (WgradUpd) and the 4 D F bytes of dW are neither written nor read back (the shipped configuration: D = F = 4096, 134 MB of the
  // update's 370).  vv_apply_update then runs only the bias / loss workgroups.
  const bool fuse_w = upd_hint && c->wgrad_update && lazy && c->S == 1 && wgrad_can_fuse_update() && !c->fuse_keep_grads &&
                      (c->Dp / BM) * (c->Fp / BN) <= WMAX_SLOTS;
  if (fuse_w) {
    const vv_step_cfg& uc = c->upd_cfg;
    WgradUpd& u = wa.upd;
    wa.fuse_upd = 1;
    u.W = c->W; u.hW = c->hW; u.Wh = c->Wh; u.scales = c->scales;
    u.wmax_blocks = c->wmax_blocks + (1 - c->wmax_cur) * WMAX_SLOTS;
    u.wmax_prev = c->wmax_blocks + c->wmax_cur * WMAX_SLOTS; u.wmax_prev_n = c->wmax_n;
    u.recompute_scale = c->scale_pending ? 1 : 0; u.prec = c->prec;
    u.D = c->D; u.F = c->F;
    u.rate = uc.lr; u.momentum = uc.momentum; u.weight_decay = uc.weight_decay; u.lr_mult_w = uc.lr_mult[0]; u.decay_mult_w = uc.decay_mult[0];
    u.delta = uc.delta; u.reg = uc.reg; u.solver_type = uc.solver_type;
    u.sg = ra.sg; u.gg = ra.gg; u.ip_scale = ra.ip_scale;
  }
  // f16 split-K partial products (option "slab16"): the phase-staggered kernel with several splits only (one split: the update rides in the
  // epilogue or the slab is the gradient); whoever reduces the slabs -- k_reduce, k_reduce_sgd, now or lazily -- reads the same flag
  if (c->slab16 && !fuse_w && c->S > 1 && c->S <= 8 && c->F % 8 == 0 && wgrad_can_fuse_update()) {
    wa.slab16 = 1; wa.slab_sc = c->slab_sc;
    ra.slab16 = 1; ra.slab_sc = c->slab_sc;
  }
  PROFILED(c, "wgrad_gemm", launch_wgrad_gemm(c->prec, wa, s));
  c->red_lazy = false; c->grads_stale = false;        // (the slabs now hold this step's gradient)
  c->grads_lost = false;
  if (fuse_w) {
    if (wa.upd.recompute_scale && c->wmax_seed_live) {       // vv_params_set's seed has now been folded into a scale: clear it behind the launch
      HIPCHK(hipMemsetAsync(&c->scales->wmax_bits, 0, sizeof(unsigned), s));
      c->wmax_seed_live = false;
    }
    c->upd_in_wgrad = true; c->upd_wgrad_blocks = (c->Dp / BM) * (c->Fp / BN);
  }
  if (lazy) {
    c->red_args = ra; c->red_lazy = true;
    c->grads_pending = false; c->grads_chunked = false; c->chunk0_event = false;
    HIPCHK(hipGetLastError());
    c->have_fwd = true;
    return VV_OK;
  }
  if (chunked && ra.n_chunks > 1 && ra.chunk_c0[1] < c->F) {
    // the first F-chunk is reduced by a launch of its own and an event marks it: the communication stream starts on
    // chunk 0 (all-reduce, SGD) while the other chunks, db and the loss are still being reduced here
    ReduceArgs r0 = ra;
    r0.f_begin = 0; r0.f_count = ra.chunk_c0[1]; r0.parts = 1; r0.gmax_host = nullptr;
    PROFILED(c, "reduce", launch_reduce(r0, s));
    HIPCHK(hipEventRecord(c->ev_chunk0, s));
    ra.scale_sc = nullptr;
    ra.f_begin = ra.chunk_c0[1]; ra.f_count = c->F - ra.f_begin;
    launch_reduce(ra, s);
    c->chunk0_event = true;
  } else {
    PROFILED(c, "reduce", launch_reduce(ra, s));
    c->chunk0_event = false;
  }
  c->grads_pending = c->comm != nullptr;       // (a one-rank communicator still runs its collective: same code path)
  c->grads_chunked = chunked;
  c->grads_sharded = sharded;

  HIPCHK(hipGetLastError());
  c->have_fwd = true;
  return VV_OK;
}

int vv_forward_backward(vv_ctx* c, const vv_step_cfg* cfg, const int32_t* idx, int idx_on_device) {
  HintScope hint_scope(c);
  if (idx_on_device < 0 || idx_on_device > 2) return fail(VV_ERR_ARG, "vv_forward_backward: idx_on_device must be 0, 1 or 2");
  return fb_impl(c, cfg, idx, idx_on_device, c ? c->n_rows : 0);
}

// BasePrefetchingDataLayer::Forward_gpu (base_data_layer.cu:7-21) joins the prefetch thread and copies the batch to the
// device; here the batch is 4*B*(C+Nn) bytes of indices out of the sampler's ring.  They are copied into one of kStage
// pinned buffers that the GPU reads IN PLACE (the first kernel of the step, k_dd_claim / k_map_rows, reads every index
// exactly once, coalesced, straight over PCIe): no copy engine, no second stream, no event.  (An asynchronous H2D copy
// per step was measured first: ~25 us per step of queue switching between the copy and the kernels, and a one-off 6-7 ms
// host stall inside hipMemcpyAsync after a handful of copies.)  A slot is reused only after the forward GEMM of the
// step that read it has stamped its sequence number into host-visible memory.
int vv_forward_backward_ring(vv_ctx* c, const vv_step_cfg* cfg, vv_batch_ring* ring, int32_t consumer, int32_t item_begin,
                             int32_t* label_out, double timeout_s) {
  HintScope hint_scope(c);
  int rc = check_cfg(c, cfg);
  if (rc) return rc;
  if (!ring) return fail(VV_ERR_ARG, "vv_forward_backward_ring: ring is NULL");
  int32_t rb = 0, rcn = 0;
  if (vv_batch_ring_info(ring, &rb, &rcn, nullptr, nullptr)) return fail(VV_ERR_ARG, "vv_forward_backward_ring: bad ring");
  if (rcn != cfg->C + cfg->Nn || item_begin < 0 || item_begin + cfg->B > rb)
    return fail(VV_ERR_ARG, "vv_forward_backward_ring: the ring holds batches of %d x %d slots; asked for items [%d, %d) x %d",
                rb, rcn, item_begin, item_begin + cfg->B, cfg->C + cfg->Nn);
  VV_ENTER(c);
  const size_t bytes = (size_t)cfg->B * rcn * sizeof(int32_t);
  int sl = 0;
  const double t0 = c->trace_host_ms >= 0 ? host_now_ms() : 0.0;
  if ((rc = stage_acquire(c, bytes, &sl))) return rc;
  const double t1 = c->trace_host_ms >= 0 ? host_now_ms() : 0.0;
  if (vv_batch_ring_next(ring, consumer, item_begin, cfg->B, c->stage_host[sl], label_out, timeout_s))
    return fail(VV_ERR_STATE, "vv_forward_backward_ring: no batch (the sampler's prefetch stopped, or timeout)");
  if (c->trace_host_ms >= 0) {
    const double t2 = host_now_ms();
    if (t2 - t0 > c->trace_host_ms)
      fprintf(stderr, "[vv host] ring stage: slot wait %.3f, batch wait + copy %.3f ms (call %llu)\n", t1 - t0, t2 - t1, (unsigned long long)c->iter);
  }
  const int32_t seq = ++c->step_seq;
  c->stage_seq[sl] = seq;
  // (Issuing the grouping of the NEXT batch from here, a whole step ahead, was built and measured: 0.2345 against 0.2334 ms
  // per step over 20-step timed regions, two runs each -- the host is ahead of the GPU anyway, and the first step after a
  // synchronisation does not wait for its grouping in any noticeable way.  Taken out again.)
  return fb_impl(c, cfg, c->stage_dev[sl], 2, c->n_rows, seq);
}

// Quirk Q1 (video_sampled_shots_data_layer.cpp:492): a same-video negative is copied WITHOUT its
// last feature, which keeps whatever the prefetch slot held before.  Such a slot is described by
// (idx = row of features 0..F-2, last_src = row of feature F-1, -1 = zero).  Each one is
// materialised as a scratch row behind the table and the batch then points at the scratch row.
// scratch rows behind the table's zero row (quirk-Q1 composites, TEST-branch means)
static int ensure_scratch_rows(vv_ctx* c, int64_t need) {
  if (need <= c->patch_cap) return VV_OK;
  HIPCHK(hipStreamSynchronize(c->stream));
  const int64_t cap = std::max<int64_t>(2 * need, 1024);
  uint16_t* nt = nullptr;
  const size_t old_bytes = (size_t)(c->n_rows + 1) * c->Fp * 2;
  HIPCHK(hipMalloc(&nt, (size_t)(c->n_rows + 1 + cap) * c->Fp * 2));
  HIPCHK(hipMemcpy(nt, c->table, old_bytes, hipMemcpyDeviceToDevice));
  dfree(c->table); c->table = nt; c->patch_cap = cap;
  return VV_OK;
}

int vv_forward_backward_q1(vv_ctx* c, const vv_step_cfg* cfg, const int32_t* idx, const int32_t* last_src) {
  HintScope hint_scope(c);
  int rc = check_cfg(c, cfg);
  if (rc) return rc;
  if (!idx || !last_src) return fail(VV_ERR_ARG, "vv_forward_backward_q1: NULL index array");
  VV_ENTER(c);
  const int64_t R = (int64_t)cfg->B * (cfg->C + cfg->Nn);
  // The patched indices AND the composite rows' descriptors travel in one pinned, device-mapped staging slot that the kernels read
  // in place (as vv_forward_backward_ring's indices do): no host-side copy to wait for.  (Round 3 copied the descriptors from a
  // host temporary and synchronised the stream on EVERY step: the device idled ~30 us of the shipped configuration's 0.29 ms.)
  int sl = 0;
  if ((rc = stage_acquire(c, (size_t)R * 12, &sl))) return rc;          // R indices + at most R descriptors of two words
  int32_t* dst = c->stage_host[sl];
  int32_t* desc = dst + R;
  int64_t P = 0;
  for (int64_t i = 0; i < R; ++i) {
    if (idx[i] < -1 || idx[i] >= c->n_rows || last_src[i] < -1 || last_src[i] >= c->n_rows)
      return fail(VV_ERR_ARG, "index %lld out of range", (long long)i);
    int32_t v = idx[i];
    if (idx[i] != last_src[i] && idx[i] >= 0) {
      v = (int32_t)(c->n_rows + 1 + P);
      desc[2 * P] = idx[i]; desc[2 * P + 1] = last_src[i];
      ++P;
    }
    dst[i] = v;
  }
  if (P > 0 && (rc = ensure_scratch_rows(c, P))) return rc;
  const int32_t seq = ++c->step_seq;
  c->stage_seq[sl] = seq;                                               // the slot is free again when this step's forward GEMM has started
  if (P > 0) launch_patch_rows(c->table, c->stage_dev[sl] + R, P, c->n_rows + 1, c->F, c->Fp, c->stream);
  return fb_impl(c, cfg, c->stage_dev[sl], 2, c->n_rows + 1 + P, seq);
}

int vv_apply_update(vv_ctx* c, const vv_step_cfg* cfg) {
  int rc = check_cfg(c, cfg);
  if (rc) return rc;
  if (!c->have_fwd) return fail(VV_ERR_STATE, "vv_apply_update: no gradients (call vv_forward_backward)");
  VV_ENTER(c);
  // two updates in a row: the first one completes first.  (After a gated forward GEMM -- upd_unjoined -- nothing is due here: the
  // new update is queued on the communication stream behind the old one, and what this step's kernels on the compute stream read
  // of the old one -- the half copy, its scale, the bias, the per-block maxima -- was stored at agent scope before the gates opened.)
  // That holds only while the new update goes to the communication stream too.  One that runs on the COMPUTE stream -- the schedule was
  // switched to `sync`, the gradient buffer was handed out or bound after an overlapped step, the sharded update runs in-stream -- reads
  // W and the history, which the previous update wrote with plain stores released only at its end: it joins first.
  const bool overlapped = c->comm && c->grads_pending && c->grads_chunked;
  const bool sharded = c->comm && c->grads_pending && c->grads_sharded;
  const bool on_comm_stream = overlapped || (sharded && !c->comm_inline);
  if ((c->upd_inflight || (c->upd_unjoined && !on_comm_stream)) && (rc = comm_join(c))) return rc;
  // A sharded update left the other ranks' rows of the fp32 master W and of the history stale here; an update of the WHOLE matrix needs
  // them (the step left the sharded path -- gradients exposed or bound, the schedule changed -- without a gather).  A
  // collective, like the update itself: every rank takes this branch in the same iteration.
  if (!sharded && c->params_partial && (rc = gather_params(c))) return rc;
  if (!overlapped && !sharded && c->grads_pending && (rc = vv_allreduce_grads(c))) return rc;     // data-parallel: the update consumes the SUM over the ranks
  SgdArgs a;
  float* const wmax_new = c->wmax_blocks + (1 - c->wmax_cur) * WMAX_SLOTS;     // the buffer the previous update did not write
  a.W = c->W; a.b = c->b; a.hW = c->hW; a.hb = c->hb; a.grads = c->grads; a.Wh = c->Wh; a.scales = c->scales; a.wmax_blocks = wmax_new;
  a.D = c->D; a.F = c->F; a.Dp = c->Dp; a.Fp = c->Fp;
  a.rate = cfg->lr; a.momentum = cfg->momentum; a.weight_decay = cfg->weight_decay;
  a.lr_mult_w = cfg->lr_mult[0]; a.lr_mult_b = cfg->lr_mult[1];
  a.decay_mult_w = cfg->decay_mult[0]; a.decay_mult_b = cfg->decay_mult[1];
  a.reg = cfg->reg; a.solver_type = cfg->solver_type; a.delta = cfg->delta;
  a.skip_if = c->comm ? vv::comm_fail_flag(c->comm) : nullptr;
  if (c->upd_in_wgrad) {
    // the parameter matrix was updated in the weight-gradient GEMM (vv_update_hint): bias, loss, the guard's report -- k_reduce_sgd's
    // special workgroups alone -- and the bookkeeping of the scale
    const vv_step_cfg& uc = c->upd_cfg;
    if (cfg->lr != uc.lr || cfg->momentum != uc.momentum || cfg->weight_decay != uc.weight_decay || cfg->lr_mult[0] != uc.lr_mult[0] ||
        cfg->decay_mult[0] != uc.decay_mult[0] || cfg->reg != uc.reg || cfg->solver_type != uc.solver_type || cfg->delta != uc.delta)
      return fail(VV_ERR_ARG, "vv_apply_update: the solver parameters differ from those announced by vv_update_hint (the weights were updated with the announced ones)");
    FusedUpdArgs fa;
    fa.r = c->red_args; fa.g = a; fa.prec = c->prec; fa.no_params = 1; fa.recompute_scale = 0; fa.store_grads = 0;
    c->red_lazy = false; c->grads_stale = false; c->upd_in_wgrad = false;
    c->grads_lost = true;                       // (dW of this step never existed outside the GEMM's registers)
    PROFILED(c, "reduce_sgd", (void)launch_reduce_sgd(fa, c->stream));
    c->wmax_cur = 1 - c->wmax_cur; c->wmax_n = c->upd_wgrad_blocks;
    c->scale_pending = true;
    HIPCHK(hipGetLastError());
    c->iter++;
    c->prof_calls++;
    return VV_OK;
  }
  if (c->red_lazy && !overlapped) {
    // the reduction is still due: reduce and update in one launch (k_reduce_sgd)
    FusedUpdArgs fa;
    fa.r = c->red_args; fa.g = a; fa.prec = c->prec;
    fa.recompute_scale = c->scale_pending ? 1 : 0;
    fa.wmax_prev = c->wmax_blocks + c->wmax_cur * WMAX_SLOTS; fa.wmax_prev_n = c->wmax_n;
    const bool keep_grads = c->fuse_keep_grads;
    fa.store_grads = keep_grads;
    c->red_lazy = false; c->grads_stale = !keep_grads;
    int n_new = 0;
    PROFILED(c, "reduce_sgd", (n_new = launch_reduce_sgd(fa, c->stream)));
    if (fa.recompute_scale && c->wmax_seed_live) {       // vv_params_set's seed has now been folded into a scale: clear it behind the launch
      HIPCHK(hipMemsetAsync(&c->scales->wmax_bits, 0, sizeof(unsigned), c->stream));
      c->wmax_seed_live = false;
    }
    c->wmax_cur = 1 - c->wmax_cur; c->wmax_n = n_new;
    c->scale_pending = true;
    HIPCHK(hipGetLastError());
    c->iter++;
    c->prof_calls++;
    return VV_OK;
  }
  if ((rc = reduce_now(c))) return rc;
  flush_scale_update(c);               // two updates in a row without a step between them
  if (sharded) {
    // Exact synchronous SGD, the update SHARDED over the ranks (Multi-GPU): on the communication stream, behind one event of
    // the compute stream,   reduce-scatter (fp32, shard-major buffer: 7/8 of it on the wire at N = 8, once)  ->  k_sgd on THIS rank's
    // D / N rows (1/N of the update's 46 MB)  ->  ONE grouped all-gather of what every rank's next forward pass reads: the 16-bit copy
    // of W (4 MB instead of the all-reduce's second 8.4 MB), the bias, the per-block maxima the next W -> half scale comes from  ->
    // publish.  3/4 of the all-reduce's wire bytes.  The next forward GEMM is gated on the one flag (it starts, and waits in front of
    // its first W tile); the fp32 master W and the history stay sharded until vv_params_get gathers them.
    // comm_inline (default): this schedule has ONE gate, in front of the GEMM's first W tile -- nothing of the exchange hides behind the
    // GEMM, and the second stream costs its hand-off chain (event -> wake -> ... -> publish: + 17 us on one rank).  So the
    // three steps are queued on the COMPUTE stream itself, in order, and the next forward GEMM is the plain kernel (with its sibling lead).
    const bool inl = c->comm_inline;
    hipStream_t cs = inl ? c->stream : vv::comm_stream(c->comm);
    const int world = vv::comm_world(c->comm), rank = vv::comm_rank(c->comm);
    const int rps = c->D / world;
    const size_t shard_f = (size_t)rps * c->F + rps;
    int32_t useq = c->upd_seq;
    if (!inl) { HIPCHK(hipEventRecord(c->ev_chunk, c->stream)); useq = ++c->upd_seq; }
    if (c->comm_test_delay_us > 0) { if (!inl) HIPCHK(hipStreamWaitEvent(cs, c->ev_chunk, 0)); launch_delay(c->comm_test_delay_us, cs); }
    struct StreamScope { vv::Comm* k; ~StreamScope() { vv::comm_use_stream(k, nullptr); } } scope{c->comm};
    vv::comm_use_stream(c->comm, inl ? c->stream : nullptr);
    if (vv::comm_reduce_scatter(c->comm, c->grads, shard_f, inl ? nullptr : c->ev_chunk)) return fail(VV_ERR_HIP, "reduce-scatter: %s", vv::comm_error(c->comm));
    const int nb = SGD_BLOCKS / world;
    SgdArgs g = a;                                    // the shard as a parameter matrix of its own: rps rows, its gradient buffer [dW rows | db entries]
    g.D = rps;
    g.W = c->W + (size_t)rank * rps * c->F; g.hW = c->hW + (size_t)rank * rps * c->F;
    g.b = c->b + (size_t)rank * rps; g.hb = c->hb + (size_t)rank * rps;
    g.grads = c->grads + (size_t)rank * shard_f;
    g.Wh = c->Wh + (size_t)rank * rps * c->Fp;
    g.chunked = 1; g.f_begin = 0; g.f_count = c->F; g.do_bias = 1; g.set_scale = 1;      // (chunked = 1 with the whole width: this launch's own grid and slots)
    g.blk_off = rank * nb; g.n_blk = nb;
    PROFILED(c, "sgd", launch_sgd(c->prec, g, cs));
    void* bufs[3] = {c->Wh, c->b, wmax_new};
    const size_t sbytes[3] = {(size_t)rps * c->Fp * 2, (size_t)rps * 4, (size_t)nb * 4};
    if (vv::comm_allgather(c->comm, bufs, sbytes, 3)) return fail(VV_ERR_HIP, "all-gather: %s", vv::comm_error(c->comm));
    if (!inl) {
      launch_publish(c->w_gate, useq, cs);
      if (vv::comm_record_done(c->comm)) return fail(VV_ERR_HIP, "all-gather: %s", vv::comm_error(c->comm));
    }
    c->grads_pending = false; c->upd_inflight = !inl; c->upd_unjoined = false;
    c->params_partial = world > 1;
    c->scale_pending = true;
    c->wmax_cur = 1 - c->wmax_cur; c->wmax_n = nb * world;
    HIPCHK(hipGetLastError());
    c->iter++;
    c->prof_calls++;
    return VV_OK;
  }
  if (overlapped) {
    // Exact synchronous SGD with the exchange hidden behind the NEXT step's forward GEMM: behind one event of the compute
    // stream, the communication stream runs per F-chunk  all-reduce -> SGD on the chunk's columns -> publish w_gate[chunk];
    // the next forward GEMM (fb_impl) starts right away and waits for each chunk where its K loop reaches it.  db rides
    // with the last chunk; the bias is updated there.
    hipStream_t cs = vv::comm_stream(c->comm);
    const int32_t useq = ++c->upd_seq;
    const int nch = chunk_plan(c);               // (the layout k_reduce wrote: same Fp, same plan)
    a.pub_count = c->pub_count; a.pub_seq = useq;
    // Round 5, an option (overlap_first_inline; OFF by default: measured slower, vv_ctx.h): the FIRST chunk on the compute stream.  The next forward GEMM cannot pass its first
    // gate before chunk 0 has arrived whatever stream brings it -- on the communication stream it arrived behind the hand-off chain
    // [event -> that stream wakes -> the collective's launch -> k_sgd on the CUs the waiting GEMM leaves free -> publish], ~17 us during
    // which the GEMM sat at the gate (profiles/r04_overlap_cost.txt: overlap = sync + 20 us on one rank).  In-stream, chunk 0's exchange and
    // update are the synchronous schedule's (no hand-off, the whole chip for its k_sgd), the GEMM starts behind them with gate 0 open, and the
    // chain of chunk 1 runs beside the GEMM's first K-tiles (chunk 0's columns: half of the loop) instead of in front of them.
    // The other chunks follow chunk 0 through ONE event, recorded behind its kernel: the collectives of a step then run in one order on every
    // transport (the direct peer transport's meeting points are numbered in host order and must be reached in that order: with chunk 1 free to
    // start beside chunk 0 its meeting overtook chunk 0's and the ranks waited for each other for good -- the first build of this, caught by
    // tests/test_gpu_dist.py [peer-overlap]).  RCCL serialises a communicator's collectives anyway.
    const bool first_inl = c->overlap_first_inline && nch > 1;
    if (!first_inl) HIPCHK(hipEventRecord(c->ev_chunk, c->stream));
    struct StreamScope { vv::Comm* k; ~StreamScope() { vv::comm_use_stream(k, nullptr); } } scope{c->comm};
    for (int k = 0; k < nch; ++k) {
      const bool inl = first_inl && k == 0;
      hipStream_t ks = inl ? c->stream : cs;
      // (second-stream form) chunk 0 may start as soon as ITS reduction is done (ev_chunk0, fb_impl); the others follow the whole backward pass
      hipEvent_t after = inl ? nullptr : (first_inl ? (k == 1 ? c->ev_chunk : nullptr)
                          : (k == 0 ? (c->chunk0_event ? c->ev_chunk0 : c->ev_chunk) : (k == 1 && c->chunk0_event ? c->ev_chunk : nullptr)));
      const int c0 = std::min(c->F, c->chunk_kt[k] * BK), c1 = std::min(c->F, c->chunk_kt[k + 1] * BK);
      const bool last = k == nch - 1;
      const size_t off = (size_t)c->D * c0, n = (size_t)c->D * (c1 - c0) + (last ? (size_t)c->D : 0);
      const int delay_us = c->comm_test_delay_us;
      if (after) HIPCHK(hipStreamWaitEvent(cs, after, 0));
      if (delay_us > 0) launch_delay(delay_us, ks);      // test hook: a slow exchange, so that the next forward GEMM really waits at its gates
      const bool skip_ar1 = c->comm_skip_ar1;   // (lab) diagnosis: no collective call at world 1
      vv::comm_use_stream(c->comm, inl ? c->stream : nullptr);
      if (!(skip_ar1 && vv::comm_world(c->comm) == 1) && n > 0 && vv::comm_allreduce(c->comm, c->grads, off, n, nullptr))
        return fail(VV_ERR_HIP, "all-reduce: %s", vv::comm_error(c->comm));
      a.chunked = 1; a.f_begin = c0; a.f_count = c1 - c0; a.do_bias = last; a.set_scale = k == 0;
      a.blk_off = k * (SGD_BLOCKS / nch); a.n_blk = last ? SGD_BLOCKS - a.blk_off : SGD_BLOCKS / nch;      // together: every slot of wmax_blocks
      a.pub_flag = c->w_gate + k * W_GATE_STRIDE;          // the kernel's last workgroup publishes the chunk (SgdArgs::pub_flag)
      a.pub_count = inl ? c->pub_count0 : c->pub_count;    // (a counter of its own: the chunk-1 kernel may run at the same time)
      if (k == 0) PROFILED(c, "sgd", launch_sgd(c->prec, a, ks)); else launch_sgd(c->prec, a, ks);     // (an empty chunk: its wmax slots become 0, the bias if it is the last)
      if (inl) HIPCHK(hipEventRecord(c->ev_chunk, c->stream));
    }
    vv::comm_use_stream(c->comm, nullptr);
    if (vv::comm_record_done(c->comm)) return fail(VV_ERR_HIP, "all-reduce: %s", vv::comm_error(c->comm));
    c->grads_pending = false; c->upd_inflight = true; c->upd_unjoined = false;    // (comm_done_event now marks the end of THIS update, behind the old one)
  } else
  PROFILED(c, "sgd", launch_sgd(c->prec, a, c->stream));
  // The next W -> half scale (k_scale_update: folds this kernel's per-block max |w|) is needed by the NEXT k_sgd only.  It
  // is left pending and performed by one extra workgroup of the next step's k_reduce; anything else that touches the
  // scales or the parameters first flushes it as its own launch.
  c->scale_pending = true;
  c->wmax_cur = 1 - c->wmax_cur; c->wmax_n = SGD_BLOCKS;
  HIPCHK(hipGetLastError());
  c->iter++;
  c->prof_calls++;
  return VV_OK;
}

int vv_step(vv_ctx* c, const vv_step_cfg* cfg, const int32_t* idx, int idx_on_device) {
  int rc = vv_update_hint(c, cfg);                  // (nothing reads the gradient between the two halves of this call)
  if (rc) return rc;
  rc = vv_forward_backward(c, cfg, idx, idx_on_device);
  if (rc) return rc;
  return vv_apply_update(c, cfg);
}

int vv_loss_get(vv_ctx* c, float* loss, float* violations) {
  if (!c) return fail(VV_ERR_ARG, "vv_loss_get: ctx is NULL");
  if (!c->have_fwd) return fail(VV_ERR_STATE, "vv_loss_get: no forward pass yet");
  VV_ENTER(c);
  if (c->gate_err && *(volatile int32_t*)c->gate_err) { const int rcj = comm_join(c); if (rcj) return rcj; }
  { const int rcr = reduce_now(c); if (rcr) return rcr; }
  float h[2];
  HIPCHK(hipMemcpyAsync(h, c->loss2, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (loss) *loss = h[0];
  if (violations) *violations = h[1];
  return VV_OK;
}

int vv_grads_device(vv_ctx* c, void** dev_ptr, int64_t* n_floats) {
  if (!c || !dev_ptr || !n_floats) return fail(VV_ERR_ARG, "vv_grads_device: NULL argument");
  { const int rcg = upd_pending_guard(c, "vv_grads_device"); if (rcg) return rcg; }
  if (!c->grads) return fail(VV_ERR_STATE, "vv_grads_device: no parameters");
  { const int rcr = reduce_now(c); if (rcr) return rcr; }      // (whoever reads the buffer on the stream finds the gradient queued in front)
  c->grads_exposed = true;                                     // ... and every later one at the end of its vv_forward_backward
  *dev_ptr = c->grads;
  *n_floats = (int64_t)c->D * c->F + c->D;
  return VV_OK;
}

int vv_grads_bind(vv_ctx* c, void* dev_ptr) {
  { const int rcg = upd_pending_guard(c, "vv_grads_bind"); if (rcg) return rcg; }
  if (!c) return fail(VV_ERR_ARG, "vv_grads_bind: ctx is NULL");
  if (!c->grads_own) return fail(VV_ERR_STATE, "vv_grads_bind: no parameters");
  // no synchronisation: kernels already queued keep the pointer they were launched with; the next
  // vv_forward_backward writes, and the next vv_apply_update reads, the newly bound buffer
  { const int rcr = reduce_now(c); if (rcr) return rcr; }      // (a reduction still due goes to the buffer it was computed for)
  c->grads = dev_ptr ? (float*)dev_ptr : c->grads_own;
  return VV_OK;
}

int vv_grads_get(vv_ctx* c, float* dW, float* db) {
  { const int rcg = upd_pending_guard(c, "vv_grads_get"); if (rcg) return rcg; }
  if (!c) return fail(VV_ERR_ARG, "vv_grads_get: ctx is NULL");
  if (!c->have_fwd) return fail(VV_ERR_STATE, "vv_grads_get: no backward pass yet");
  VV_ENTER(c);
  { const int rcj = comm_join(c); if (rcj) return rcj; }       // block-wise all-reduces in flight: the buffer is read whole
  { const int rcr = reduce_now(c); if (rcr) return rcr; }
  HIPCHK(hipStreamSynchronize(c->stream));
  const size_t nW = (size_t)c->D * c->F;
  if (c->grads_sharded) {              // shard-major buffer (ReduceArgs::shard_rows) -> the blob's [dW | db]
    const int rps = c->D / vv::comm_world(c->comm);
    const size_t sf = (size_t)rps * c->F + rps;
    std::vector<float> tmp(nW + c->D);
    HIPCHK(hipMemcpy(tmp.data(), c->grads, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int s = 0; s * rps < c->D; ++s) {
      if (dW) memcpy(dW + (size_t)s * rps * c->F, tmp.data() + s * sf, (size_t)rps * c->F * 4);
      if (db) memcpy(db + (size_t)s * rps, tmp.data() + s * sf + (size_t)rps * c->F, (size_t)rps * 4);
    }
    return VV_OK;
  }
  if (dW && !c->grads_chunked) HIPCHK(hipMemcpy(dW, c->grads, nW * 4, hipMemcpyDeviceToHost));
  if (dW && c->grads_chunked) {        // chunk-major buffer (ReduceArgs::n_chunks) -> the blob's row-major D x F
    std::vector<float> tmp(nW);
    HIPCHK(hipMemcpy(tmp.data(), c->grads, nW * 4, hipMemcpyDeviceToHost));
    const int nch = chunk_plan(c);
    for (int k = 0; k < nch; ++k) {
      const int c0 = std::min(c->F, c->chunk_kt[k] * BK), c1 = std::min(c->F, c->chunk_kt[k + 1] * BK);
      for (int d = 0; d < c->D && c1 > c0; ++d)
        memcpy(dW + (size_t)d * c->F + c0, tmp.data() + (size_t)c->D * c0 + (size_t)d * (c1 - c0), (size_t)(c1 - c0) * 4);
    }
  }
  if (db) HIPCHK(hipMemcpy(db, c->grads + nW, c->D * 4, hipMemcpyDeviceToHost));
  return VV_OK;
}

int vv_blobs_get(vv_ctx* c, float* ip2, float* target_score, float* negative_scores, float* ip1_diff) {
  if (!c) return fail(VV_ERR_ARG, "vv_blobs_get: ctx is NULL");
  if (!c->have_fwd) return fail(VV_ERR_STATE, "vv_blobs_get: no forward pass yet");
  VV_ENTER(c);
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  HIPCHK(hipStreamSynchronize(c->stream));
  const int B = c->B, CN = c->C + c->Nn, D = c->D, Nn = c->Nn;
  const size_t n = (size_t)c->R * D;
  auto reorder = [&](const std::vector<float>& src, float* dst) {   // (b*CN+ch) -> (ch*B+b)
    for (int bb = 0; bb < B; ++bb)
      for (int ch = 0; ch < CN; ++ch)
        memcpy(dst + ((size_t)ch * B + bb) * D, src.data() + ((size_t)bb * CN + ch) * D, (size_t)D * 4);
  };
  if (ip2) {
    std::vector<float> tmp(n);
    if (c->last_dedup) {               // expand the per-slot rows back to one row per instance
      DevTmp<float> d;
      HIPCHK(d.alloc(n));
      if (c->last_drop.mode) launch_gather_rows_dropout(c->H, c->dd_map, c->R, D, c->last_drop, d, c->stream, c->last_h16);    // the instance's own mask on the shared row
      else launch_gather_rows_f32(c->H, c->dd_map, c->R, D, d, c->stream, c->last_h16);
      HIPCHK(hipStreamSynchronize(c->stream));
      HIPCHK(hipMemcpy(tmp.data(), d, n * 4, hipMemcpyDeviceToHost));
    } else {
      HIPCHK(hipMemcpy(tmp.data(), c->H, n * 4, hipMemcpyDeviceToHost));
    }
    reorder(tmp, ip2);
  }
  if (target_score) {
    std::vector<float> st(B);
    HIPCHK(hipMemcpy(st.data(), c->s_true, (size_t)B * 4, hipMemcpyDeviceToHost));
    for (int bb = 0; bb < B; ++bb) for (int k = 0; k < Nn; ++k) target_score[(size_t)bb * Nn + k] = st[bb];
  }
  if (negative_scores) HIPCHK(hipMemcpy(negative_scores, c->s_bogus, (size_t)B * Nn * 4, hipMemcpyDeviceToHost));
  if (ip1_diff) {
    DevTmp<float> d;
    HIPCHK(d.alloc(n));
    DevTmp<uint16_t> ungrouped;
    // the scale the step's 16-bit gradients really carry: the host's sg times what the guard's repeats took off
    float sgf = c->sg;
    if (c->prec == VV_PREC_F16) {
      GradGuard gh;
      HIPCHK(hipMemcpy(&gh, c->gg, sizeof(gh), hipMemcpyDeviceToHost));
      sgf *= gh.mul;
    }
    DevTmp<float> yrows; DevTmp<uint16_t> dyrows;
    bool expanded = false;
    if (c->last_seg_bwd && (c->last_drop.mode || c->last_h16)) {
      // ... with dropout: the instances' masked rows are materialised (item-major) and the per-instance kernel runs on them as on a
      // dense batch whose forward pass applied the mask -- its gradient rows come out per instance, nothing to ungroup
      // (... and with ip2 stored as f16: the per-instance kernel reads fp32 rows -- the instances' rows are materialised as such)
      HIPCHK(yrows.alloc(n));
      HIPCHK(dyrows.alloc((size_t)(c->Rp + BK) * c->Dp));
      if (c->last_drop.mode) launch_gather_rows_dropout(c->H, c->dd_map, c->R, D, c->last_drop, yrows, c->stream, c->last_h16);
      else launch_gather_rows_f32(c->H, c->dd_map, c->R, D, yrows, c->stream, c->last_h16);
      ScoreArgs la = c->last_score;
      la.H = yrows; la.dYh = dyrows; la.map = nullptr; la.seg_start = nullptr; la.ord = nullptr; la.V = nullptr; la.rec = nullptr;
      la.drop = DropSpec(); la.bound_out = nullptr; la.h16 = 0;
      la.sg = sgf; la.guard = GuardArgs();
      launch_score_loss(c->prec, la, c->stream);
      expanded = true;
    } else if (c->last_seg_bwd) {
      // the step kept its backward factored: produce the per-instance rows now (same forward values; the loss partials
      // it rewrites are the ones already there, its bias partials go to the buffer the step did not use)
      ScoreArgs la = c->last_score;
      la.sg = sgf; la.guard = GuardArgs();
      launch_score_loss(c->prec, la, c->stream);
    }
    if (expanded) {
      launch_dyh_to_float(c->prec, dyrows, c->R, D, c->Dp, 1.f / sgf, d, c->stream);
      HIPCHK(hipStreamSynchronize(c->stream));
      std::vector<float> tmp(n);
      HIPCHK(hipMemcpy(tmp.data(), d, n * 4, hipMemcpyDeviceToHost));
      reorder(tmp, ip1_diff);
      return VV_OK;
    }
    if (c->last_dedup) {
      HIPCHK(ungrouped.alloc((size_t)c->R * c->Dp));
      DedupArgs da;
      memset(&da, 0, sizeof(da));
      da.map = c->dd_map; da.ord = c->dd_ord; da.seg_start = c->dd_seg; da.pos = c->dd_pos; da.R = c->R;
      launch_dedup_pos(da, c->stream);
      launch_gather_rows_u16(c->dYh, c->dd_pos, c->R, c->Dp, ungrouped, c->stream);
    }
    launch_dyh_to_float(c->prec, ungrouped.p ? ungrouped.p : c->dYh, c->R, D, c->Dp, 1.f / sgf, d, c->stream);
    HIPCHK(hipStreamSynchronize(c->stream));
    std::vector<float> tmp(n);
    HIPCHK(hipMemcpy(tmp.data(), d, n * 4, hipMemcpyDeviceToHost));
    reorder(tmp, ip1_diff);
  }
  return VV_OK;
}

// ------------------------------------------------------------------------------- embed --------
int vv_embed(vv_ctx* c, const int32_t* rows, int64_t n, int relu, int l2norm, float* out) {
  if (!c || !out || n <= 0) return fail(VV_ERR_ARG, "vv_embed: bad argument");
  if (!c->table || !c->W) return fail(VV_ERR_STATE, "vv_embed: table and parameters must be set first");
  if (n > (1ll << 30)) return fail(VV_ERR_ARG, "vv_embed: n too large");
  { const int rcg = upd_pending_guard(c, "vv_embed"); if (rcg) return rcg; }     // (a hinted step half-way: new half copy, old bias)
  VV_ENTER(c);
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  const int D = c->D;
  const int Rp = (int)round_up(n, R_ALIGN);
  std::vector<int32_t> h(Rp, (int32_t)c->n_rows);
  for (int64_t i = 0; i < n; ++i) {
    const int64_t r = rows ? rows[i] : i;
    if (r < 0 || r >= c->n_rows) return fail(VV_ERR_ARG, "vv_embed: row %lld out of range", (long long)r);
    h[i] = (int32_t)r;
  }
  DevTmp<int32_t> drows; DevTmp<float> dout;
  HIPCHK(drows.alloc((size_t)Rp));
  HIPCHK(dout.alloc((size_t)n * D));
  HIPCHK(hipMemcpyAsync(drows, h.data(), (size_t)Rp * 4, hipMemcpyHostToDevice, c->stream));
  FwdArgs fa;
  fa.table = c->table; fa.rows = drows; fa.Wh = c->Wh; fa.bias = c->b; fa.scales = c->scales;
  fa.H = dout; fa.R = (int)n; fa.D = D; fa.Fp = c->Fp; fa.relu = relu ? 1 : 0; fa.zero_row = (int32_t)c->n_rows;
  fa.drop_ratio = 0.f; fa.mask = nullptr; fa.drop_seed = 0; fa.B = 1; fa.CN = 1;
  launch_fwd_gemm(c->prec, fa, c->stream);
  if (l2norm) launch_row_normalize(dout, (int)n, D, c->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, dout, (size_t)n * D * 4, hipMemcpyDeviceToHost));
  return VV_OK;
}

int vv_embed_mean(vv_ctx* c, const int32_t* rows, int64_t n, int32_t k, const float* coeff, int relu,
                  int l2norm, float* out) {
  if (!c || !rows || !out || n <= 0 || k <= 0) return fail(VV_ERR_ARG, "vv_embed_mean: bad argument");
  if (!c->table || !c->W) return fail(VV_ERR_STATE, "vv_embed_mean: table and parameters must be set first");
  if (n > (1ll << 24)) return fail(VV_ERR_ARG, "vv_embed_mean: n too large");
  { const int rcg = upd_pending_guard(c, "vv_embed_mean"); if (rcg) return rcg; }     // (a hinted step half-way: new half copy, old bias)
  VV_ENTER(c);
  { const int rcj = comm_join(c); if (rcj) return rcj; }
  for (int64_t i = 0; i < n * k; ++i)
    if (rows[i] < 0 || rows[i] >= c->n_rows) return fail(VV_ERR_ARG, "vv_embed_mean: row %d out of range", rows[i]);
  int rc = ensure_scratch_rows(c, n);
  if (rc) return rc;
  std::vector<float> hc(k);
  for (int j = 0; j < k; ++j) hc[j] = coeff ? coeff[j] : 1.0f / k;
  const int D = c->D;
  const int Rp = (int)round_up(n, R_ALIGN);
  std::vector<int32_t> h(Rp, (int32_t)c->n_rows);
  for (int64_t i = 0; i < n; ++i) h[i] = (int32_t)(c->n_rows + 1 + i);
  DevTmp<int32_t> drows_in, drows; DevTmp<float> dcoeff, dout;
  HIPCHK(drows_in.alloc((size_t)n * k)); HIPCHK(drows.alloc((size_t)Rp));
  HIPCHK(dcoeff.alloc((size_t)k)); HIPCHK(dout.alloc((size_t)n * D));
  HIPCHK(hipMemcpyAsync(drows_in, rows, (size_t)n * k * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(drows, h.data(), (size_t)Rp * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(dcoeff, hc.data(), (size_t)k * 4, hipMemcpyHostToDevice, c->stream));
  launch_mean_rows(c->prec, c->table, drows_in, n, k, dcoeff, c->n_rows + 1, c->Fp, c->stream);
  FwdArgs fa;
  fa.table = c->table; fa.rows = drows; fa.Wh = c->Wh; fa.bias = c->b; fa.scales = c->scales;
  fa.H = dout; fa.R = (int)n; fa.D = D; fa.Fp = c->Fp; fa.relu = relu ? 1 : 0; fa.zero_row = (int32_t)c->n_rows;
  fa.drop_ratio = 0.f; fa.mask = nullptr; fa.drop_seed = 0; fa.B = 1; fa.CN = 1;
  launch_fwd_gemm(c->prec, fa, c->stream);
  if (l2norm) launch_row_normalize(dout, (int)n, D, c->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, dout, (size_t)n * D * 4, hipMemcpyDeviceToHost));
  return VV_OK;
}

int vv_retrieval_stats(vv_ctx* c, const float* feat, int32_t n, int32_t dim, const int32_t* video_ids,
                       const int32_t* map_ids, const int32_t* map_cls, int32_t n_map,
                       int exclude_same, float* mean_ap, float* hit1, float* hit5) {
  if (!c || !feat || !video_ids || n < 2 || dim < 1 || (n_map > 0 && (!map_ids || !map_cls)))
    return fail(VV_ERR_ARG, "vv_retrieval_stats: bad argument");
  if (n_map < 1) return fail(VV_ERR_ARG, "need atleast one entry in id-to-class map!");   // retrieval_stats_layer.cpp:49
  VV_ENTER(c);
  DevTmp<float> dx, dd;
  HIPCHK(dx.alloc((size_t)n * dim)); HIPCHK(dd.alloc((size_t)n * n));
  HIPCHK(hipMemcpyAsync(dx, feat, (size_t)n * dim * 4, hipMemcpyHostToDevice, c->stream));
  launch_gram(dx, n, dim, -2.0f, dd, c->stream);                                           // :208-209
  std::vector<float> dist((size_t)n * n);
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(dist.data(), dd, dist.size() * 4, hipMemcpyDeviceToHost));
  std::unordered_map<int, int> cls;
  for (int i = 0; i < n_map; ++i) cls[map_ids[i]] = map_cls[i];
  auto cls_of = [&](int id) { auto it = cls.find(id); return it == cls.end() ? 0 : it->second; };
  std::vector<int> order(n);
  double s_ap = 0, s_1 = 0, s_5 = 0, npos = 0;
  for (int i = 0; i < n; ++i) {
    float* row = dist.data() + (size_t)i * n;
    row[i] = -1e15f;                                                                       // :228-229
    for (int j = 0; j < n; ++j) order[j] = j;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return row[a] < row[b] || (row[a] == row[b] && a < b); });
    const int label = cls_of(video_ids[i]);
    if (label < 0) continue;                                                               // :246-248
    double ap = 0, a1 = 0, a5 = 0, val = 0, ret = 0;                                       // :104-141
    for (int kk = 1; kk < n; ++kk) {
      const int j = order[kk];
      if (video_ids[j] != video_ids[i] || !exclude_same) {
        val += 1;
        if (cls_of(video_ids[j]) == label) { if (val <= 1) a1 += 1; if (val <= 5) a5 += 1; ret += 1; ap += ret / val; }
      }
    }
    if (ret > 0) ap /= ret;
    s_ap += ap; s_1 += a1; s_5 += a5 / 5; npos += 1;
  }
  if (npos == 0) return fail(VV_ERR_ARG, "vv_retrieval_stats: no sample with a non-negative class");
  if (mean_ap) *mean_ap = (float)(s_ap / npos);
  if (hit1) *hit1 = (float)(s_1 / npos);
  if (hit5) *hit5 = (float)(s_5 / npos);
  return VV_OK;
}

// ------------------------------------------------------------------------------- data parallel -
int vv_comm_init(vv_ctx* c, int32_t world, int32_t rank, const char* id_path, int32_t transport) {
  if (!c) return fail(VV_ERR_ARG, "vv_comm_init: ctx is NULL");
  if (world < 1 || rank < 0 || rank >= world) return fail(VV_ERR_ARG, "vv_comm_init: rank %d of %d", rank, world);
  if (world > 1 && (!id_path || !*id_path)) return fail(VV_ERR_ARG, "vv_comm_init: id_path is required for world > 1");
  if (transport != VV_COMM_RCCL && transport != VV_COMM_SHM && transport != VV_COMM_PEER) return fail(VV_ERR_ARG, "vv_comm_init: unknown transport %d", transport);
  if (!c->W) return fail(VV_ERR_STATE, "vv_comm_init: set the parameters first (they size the gradient buffer)");
  if (c->comm) return fail(VV_ERR_STATE, "vv_comm_init: a communicator already exists");
  { const int rcg = upd_pending_guard(c, "vv_comm_init"); if (rcg) return rcg; }
  VV_ENTER(c);
  std::string err;
  c->comm = vv::comm_create(world, rank, id_path ? id_path : "", transport, (size_t)c->D * c->F + c->D, &err);
  if (!c->comm) return fail(VV_ERR_HIP, "vv_comm_init: %s", err.c_str());
  if (!c->ev_chunk) HIPCHK(hipEventCreateWithFlags(&c->ev_chunk, hipEventDisableTiming));
  return VV_OK;
}

int vv_comm_overlap(vv_ctx* c, int on) {
  if (!c) return fail(VV_ERR_ARG, "vv_comm_overlap: ctx is NULL");
  if (on && c->comm_sharded && c->params_partial) { VV_ENTER(c); const int rcg = gather_params(c); if (rcg) return rcg; }   // (collective, as in vv_comm_schedule: leaving the sharded schedule)
  c->comm_overlap = on != 0;
  if (on) c->comm_sharded = false;
  return VV_OK;
}

int vv_comm_schedule(vv_ctx* c, int schedule) {
  if (!c) return fail(VV_ERR_ARG, "vv_comm_schedule: ctx is NULL");
  if (schedule < 0 || schedule > 2) return fail(VV_ERR_ARG, "vv_comm_schedule: 0 sync, 1 overlap, 2 sharded");
  { const int rcg = upd_pending_guard(c, "vv_comm_schedule"); if (rcg) return rcg; }
  if (schedule != 2 && c->params_partial) { VV_ENTER(c); const int rcg = gather_params(c); if (rcg) return rcg; }     // (collective: the other schedules update the whole matrix)
  c->comm_overlap = schedule == 1;
  c->comm_sharded = schedule == 2;
  return VV_OK;
}

int vv_allreduce_grads(vv_ctx* c) {
  { const int rcg = upd_pending_guard(c, "vv_allreduce_grads"); if (rcg) return rcg; }
  if (!c) return fail(VV_ERR_ARG, "vv_allreduce_grads: ctx is NULL");
  if (!c->comm) { c->grads_pending = false; return VV_OK; }
  if (!c->have_fwd) return fail(VV_ERR_STATE, "vv_allreduce_grads: no gradients (call vv_forward_backward)");
  if (!c->grads_pending) return VV_OK;                         // already summed
  if (c->grads_chunked || c->grads_sharded) return VV_OK;      // overlapped / sharded schedule: vv_apply_update runs the exchange
  VV_ENTER(c);
  // synchronous schedule over RCCL: the collective goes straight into the compute stream (nothing would run beside it)
  const int rc = vv::comm_allreduce_inline(c->comm, c->grads, (size_t)c->D * c->F + c->D, c->stream);
  if (rc < 0) return fail(VV_ERR_HIP, "all-reduce: %s", vv::comm_error(c->comm));
  if (rc == 0) { c->grads_pending = false; return VV_OK; }
  HIPCHK(hipEventRecord(c->ev_chunk, c->stream));
  if (vv::comm_allreduce(c->comm, c->grads, 0, (size_t)c->D * c->F + c->D, c->ev_chunk))
    return fail(VV_ERR_HIP, "all-reduce: %s", vv::comm_error(c->comm));
  HIPCHK(hipStreamWaitEvent(c->stream, vv::comm_done_event(c->comm), 0));   // the update waits for the sum; the host does not
  c->grads_pending = false;
  return VV_OK;
}

int vv_comm_destroy(vv_ctx* c) {
  if (!c) return VV_OK;
  if (c->comm) {
    (void)hipSetDevice(c->device);
    (void)gather_params(c);                    // (collective: a sharded update leaves the master parameters whole again)
    (void)comm_join(c); (void)hipStreamSynchronize(c->stream); vv::comm_destroy(c->comm); c->comm = nullptr;
  }
  c->grads_pending = c->grads_chunked = c->grads_sharded = c->upd_inflight = c->upd_unjoined = c->params_partial = false;
  return VV_OK;
}

// ------------------------------------------------------------------------------- profiling ----
int vv_profile_enable(vv_ctx* c, int on) {
  if (!c) return fail(VV_ERR_ARG, "vv_profile_enable: ctx is NULL");
  HIPCHK(hipStreamSynchronize(c->stream));
  c->prof_map.clear();
  c->ev_used = 0;
  c->prof = on != 0;
  c->prof_every = on > 1 ? on : 1;
  c->prof_calls = 0;
  return VV_OK;
}

int vv_profile_select(vv_ctx* c, const char* kernels) {
  if (!c) return fail(VV_ERR_ARG, "vv_profile_select: ctx is NULL");
  c->prof_only = (kernels && *kernels) ? "," + std::string(kernels) + "," : std::string();
  return VV_OK;
}

int vv_profile_get(vv_ctx* c, const char* kernel, double* avg_ms, int64_t* launches) {
  if (!c || !kernel) return fail(VV_ERR_ARG, "vv_profile_get: NULL argument");
  HIPCHK(hipStreamSynchronize(c->stream));
  auto it = c->prof_map.find(kernel);
  double tot = 0; int64_t n = 0;
  if (it != c->prof_map.end())
    for (auto& e : it->second.ev) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, e.first, e.second) == hipSuccess) { tot += ms; ++n; }
    }
  if (avg_ms) *avg_ms = n ? tot / n : 0.0;
  if (launches) *launches = n;
  return VV_OK;
}

}  // extern "C"

#ifdef VV_LAB
// (lab) copies the 16-word stamp records of the last k_score_fwd launch (VV_LAB_SCORE_TS=1) to `out` (n_items x 16 uint32)
extern "C" int vv_lab_score_ts(vv_ctx* c, uint32_t* out, int n_items) {
  if (!c || !c->lab_score_ts || !out) return VV_ERR_STATE;
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(out, c->lab_score_ts, (size_t)n_items * 16 * 4, hipMemcpyDeviceToHost));
  return VV_OK;
}
#endif
