// ops.hip -- per-layer device operators: what the reference's Forward_gpu / Backward_gpu of the 13 hot-path layer
// classes compute, one small HIP kernel each, on fp32 device buffers.  The fused plan (api.hip) is what training runs;
// these serve the facade's sequential executor (a graph the fused-plan matcher does not recognise, gradient checks,
// single-layer Forward / Backward) behind the same C ABI.  Everything is queued on the context's stream.
//
//   SLICE / CONCAT / SPLIT   slice_layer.cu:10-64, concat_layer.cu:10-75, split_layer.cu:10-33   -> vv_op_copy2d, vv_op_axpby
//   RELU                     relu_layer.cu:10-59                                                  -> vv_op_relu, vv_op_relu_bwd
//   DROPOUT                  dropout_layer.cu:14-73                                               -> vv_op_dropout, vv_op_dropout_bwd
//   ELTWISE SUM / PROD       eltwise_layer.cu:34-119                                              -> vv_op_axpby, vv_op_mul
//   SUM                      sum_layer.cu:10-55                                                   -> vv_op_rowsum, vv_op_rowsum_bwd
//   NORMALIZATION            normalization_layer.cu:10-97                                         -> vv_op_normalize, vv_op_normalize_bwd
//   MAX_MARGIN_LOSS          max_margin_loss_layer.cpp:53-214 (CPU only in the reference)          -> vv_op_max_margin, _bwd
//   INNER_PRODUCT            inner_product_layer.cu:12-59                                         -> vv_op_inner_product, _bwd
//   VIDEO_SAMPLED_SHOTS_DATA base_data_layer.cu:7-21 (the batch copy)                             -> vv_op_gather_rows
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "vv_ctx.h"

using namespace vv;

namespace {

constexpr int EB = 256;
inline dim3 egrid(int64_t n) { return dim3((unsigned)std::min<int64_t>((n + EB - 1) / EB, 4096)); }

__global__ __launch_bounds__(EB) void k_copy2d(const float* src, int64_t ss, float* dst, int64_t ds, int64_t rows, int64_t cols, int acc) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
    const int64_t r = i / cols, c = i - r * cols;
    const float v = src[r * ss + c];
    float* d = dst + r * ds + c;
    *d = acc ? *d + v : v;
  }
}
__global__ __launch_bounds__(EB) void k_axpby(int64_t n, float a, const float* x, float b, float* y) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB)
    y[i] = b == 0.f ? a * x[i] : a * x[i] + b * y[i];
}
__global__ __launch_bounds__(EB) void k_mul(int64_t n, const float* a, const float* b, float* y, int acc) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB)
    y[i] = acc ? y[i] + a[i] * b[i] : a[i] * b[i];
}
// relu_layer.cu:10-27: y = max(x, 0) + slope * min(x, 0)
__global__ __launch_bounds__(EB) void k_relu(int64_t n, const float* x, float* y, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
    const float v = x[i];
    y[i] = v > 0.f ? v : v * slope;
  }
}
// relu_layer.cu:36-59: dx = dy * ((x > 0) + slope * (x <= 0))
__global__ __launch_bounds__(EB) void k_relu_bwd(int64_t n, const float* x, const float* dy, float* dx, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB)
    dx[i] = dy[i] * (x[i] > 0.f ? 1.f : slope);
}
// dropout_layer.cu:14-41: mask = uniform >= ratio (the reference draws curand uints and compares with UINT_MAX * ratio);
// here a counter-based hash of (seed, element), the generator of the fused path
__global__ __launch_bounds__(EB) void k_dropout(int64_t n, const float* x, float* y, uint8_t* mask, float ratio, float scale,
                                                uint64_t seed, int make_mask) {
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
    uint8_t m = mask[i];
    if (make_mask) { m = (float)(mix64(seed, (uint64_t)i) >> 40) * (1.0f / 16777216.0f) >= ratio; mask[i] = m; }
    y[i] = m ? x[i] * scale : 0.f;
  }
}
// sum_layer.cu:10-32: y[r][o] = sum_c x[r][c] for every o < num_output; one wave per row
__global__ __launch_bounds__(EB) void k_rowsum(int64_t rows, int cols, const float* x, int num_output, float* y) {
  const int64_t r = (int64_t)blockIdx.x * (EB / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += x[r * cols + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  for (int o = lane; o < num_output; o += 64) y[r * num_output + o] = s;
}
// sum_layer.cu:34-55: dx[r][c] = sum_o dy[r][o]
__global__ __launch_bounds__(EB) void k_rowsum_bwd(int64_t rows, int cols, int num_output, const float* dy, float* dx) {
  const int64_t r = (int64_t)blockIdx.x * (EB / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int o = lane; o < num_output; o += 64) s += dy[r * num_output + o];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  for (int c = lane; c < cols; c += 64) dx[r * cols + c] = s;
}
// normalization_layer.cu:10-48: y = x / (sqrt(sum x^2) + 1e-10)
__global__ __launch_bounds__(EB) void k_normalize(int64_t rows, int cols, const float* x, float* y) {
  const int64_t r = (int64_t)blockIdx.x * (EB / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) { const float v = x[r * cols + c]; s += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const float inv = 1.f / (sqrtf(s) + 1e-10f);
  for (int c = lane; c < cols; c += 64) y[r * cols + c] = x[r * cols + c] * inv;
}
// normalization_layer.cu:50-97: dx = (s dy - x (x . dy)) / (s^1.5 + 1e-10), s = sum x^2
__global__ __launch_bounds__(EB) void k_normalize_bwd(int64_t rows, int cols, const float* x, const float* dy, float* dx) {
  const int64_t r = (int64_t)blockIdx.x * (EB / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f, d = 0.f;
  for (int c = lane; c < cols; c += 64) { const float v = x[r * cols + c]; s += v * v; d += v * dy[r * cols + c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); d += __shfl_xor(d, o, 64); }
  const float inv = 1.f / (s * sqrtf(s) + 1e-10f);
  for (int c = lane; c < cols; c += 64) dx[r * cols + c] = (s * dy[r * cols + c] - x[r * cols + c] * d) * inv;
}
// max_margin_loss_layer.cpp:53-127: one workgroup, fixed summation order.  out = {loss, violations}
__global__ __launch_bounds__(EB) void k_max_margin(int count, const float* st, const float* sb, const float* w, float margin,
                                                   int norm, float* out) {
  __shared__ double sl[EB];
  __shared__ float sv[EB];
  double acc = 0.0; float nv = 0.f;
  for (int i = threadIdx.x; i < count; i += EB) {
    const float d = st[i] - sb[i];
    if (d < 0.f) nv += 1.f;
    float h = fmaxf(0.f, margin - d);
    if (w) h *= norm == 2 ? sqrtf(w[i]) : w[i];
    acc += norm == 2 ? (double)h * h : fabs((double)h);
  }
  sl[threadIdx.x] = acc; sv[threadIdx.x] = nv;
  __syncthreads();
  for (int o = EB / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sl[threadIdx.x] += sl[threadIdx.x + o]; sv[threadIdx.x] += sv[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[0] = (float)(sl[0] / count); out[1] = sv[0]; }
}
// max_margin_loss_layer.cpp:129-214
__global__ __launch_bounds__(EB) void k_max_margin_bwd(int count, const float* st, const float* sb, const float* w, float margin,
                                                       int norm, float loss_weight, float* dt, float* db) {
  for (int i = blockIdx.x * EB + threadIdx.x; i < count; i += gridDim.x * EB) {
    float h = fmaxf(0.f, margin - (st[i] - sb[i]));
    if (w) h *= w[i];
    float g;
    if (norm == 1) g = (h > 0.f ? (w ? w[i] : 1.f) : h) * (loss_weight / count);
    else g = h * (loss_weight * 2 / count);
    db[i] = g; dt[i] = -g;
  }
}
// fp32 rows -> scaled 16-bit rows [R][Dp] (the weight-gradient kernel's dY operand)
// the scale of those rows, on the device: max |dY| (k_absmax's float bits) placed in [2^11, 2^12) for f16 -- nothing can
// leave f16's range, whatever the caller's gradients are; 1 for bf16
__global__ void k_pick_scale(const unsigned* max_bits, int prec, float* sg_out) {
  const float m = __uint_as_float(*max_bits);
  float sg = 1.f;
  if (prec == 0 && m > 0.f && isfinite(m)) { int e; (void)frexpf(m, &e); sg = ldexpf(1.f, 12 - e); }
  *sg_out = sg;
}
template <typename T>
__global__ __launch_bounds__(EB) void k_to_half_rows(const float* src, uint16_t* dst, int64_t rows, int cols, int cols_p, const float* scale_dev) {
  const float scale = *scale_dev;
  const int64_t n = rows * cols_p;
  for (int64_t i = (int64_t)blockIdx.x * EB + threadIdx.x; i < n; i += (int64_t)gridDim.x * EB) {
    const int64_t r = i / cols_p; const int c = (int)(i - r * cols_p);
    dst[i] = c < cols ? T::from_float(src[r * cols + c] * scale) : (uint16_t)0;
  }
}
// db = column sums of dY (inner_product_layer.cu:45-49: gemv with the ones vector); one workgroup per 64 columns
__global__ __launch_bounds__(EB) void k_colsum(const float* dy, int64_t rows, int cols, float* out) {
  __shared__ float sm[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  float s = 0.f;
  if (c < cols) for (int64_t r = g; r < rows; r += 4) s += dy[r * cols + c];
  sm[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && c < cols) out[c] = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
}
__global__ __launch_bounds__(EB) void k_iota(int32_t* p, int n, int n_pad, int32_t pad) {
  const int i = blockIdx.x * EB + threadIdx.x;
  if (i < n_pad) p[i] = i < n ? i : pad;
}

struct OpScratch {                 // per-context buffers of the INNER_PRODUCT operators, kept across calls
  uint16_t* x16 = nullptr; int64_t x_rows = 0;       // [rows + 1][Fp], last row zero
  int32_t* ident = nullptr;
  uint16_t* dy16 = nullptr; float* slabs = nullptr; size_t slab_bytes = 0;
  float* loss2 = nullptr;
  float* sgs = nullptr;                              // {max |dY| bits, scale} of the last backward call
  int Fp = 0, Dp = 0;                                // the padded widths x16 / dy16 were sized for
};

}  // namespace

// one entry per context; the map itself is shared by every context of the process (one lock around look-ups: contexts
// may live on different host threads; an entry is only ever used by its own context's caller)
static std::mutex g_scratch_mu;
static std::map<vv_ctx*, OpScratch>& scratch_map_unlocked() { static std::map<vv_ctx*, OpScratch> m; return m; }
static OpScratch& scratch_of(vv_ctx* c) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  return scratch_map_unlocked()[c];              // (std::map: references stay valid across other insertions)
}
void vv_ops_release(vv_ctx* c) {
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  auto& scratch = scratch_map_unlocked();
  auto it = scratch.find(c);
  if (it == scratch.end()) return;
  OpScratch& s = it->second;
  if (s.x16) (void)hipFree(s.x16);
  if (s.ident) (void)hipFree(s.ident);
  if (s.dy16) (void)hipFree(s.dy16);
  if (s.slabs) (void)hipFree(s.slabs);
  if (s.loss2) (void)hipFree(s.loss2);
  if (s.sgs) (void)hipFree(s.sgs);
  scratch.erase(it);
}

#define NEED(c) do { if (!(c)) return vv_fail(VV_ERR_ARG, "%s: ctx is NULL", __func__); HIPCHK(hipSetDevice((c)->device)); vv::g_ko = &(c)->ko; } while (0)

extern "C" {

int vv_dev_alloc(vv_ctx* c, size_t bytes, void** out) {
  NEED(c);
  if (!out) return vv_fail(VV_ERR_ARG, "vv_dev_alloc: out is NULL");
  *out = nullptr;
  if (bytes == 0) return VV_OK;
  HIPCHK(hipMalloc(out, bytes));
  HIPCHK(hipMemsetAsync(*out, 0, bytes, c->stream));
  return VV_OK;
}
int vv_dev_free(vv_ctx* c, void* p) {
  NEED(c);
  if (p) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipFree(p)); }
  return VV_OK;
}
int vv_dev_upload(vv_ctx* c, void* dst, const void* src, size_t bytes) {
  NEED(c);
  if (bytes) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
  return VV_OK;
}
int vv_dev_download(vv_ctx* c, void* dst, const void* src, size_t bytes) {
  NEED(c);
  if (bytes) { HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream)); HIPCHK(hipStreamSynchronize(c->stream)); }
  return VV_OK;
}
int vv_dev_memset(vv_ctx* c, void* dst, int value, size_t bytes) {
  NEED(c);
  if (bytes) HIPCHK(hipMemsetAsync(dst, value, bytes, c->stream));
  return VV_OK;
}

int vv_op_copy2d(vv_ctx* c, const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t rows, int64_t cols,
                 int accumulate) {
  NEED(c);
  if (rows * cols > 0) hipLaunchKernelGGL(k_copy2d, egrid(rows * cols), dim3(EB), 0, c->stream, src, src_stride, dst, dst_stride, rows, cols, accumulate);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_axpby(vv_ctx* c, int64_t n, float a, const float* x, float b, float* y) {
  NEED(c);
  if (n > 0) hipLaunchKernelGGL(k_axpby, egrid(n), dim3(EB), 0, c->stream, n, a, x, b, y);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_mul(vv_ctx* c, int64_t n, const float* a, const float* b, float* y, int accumulate) {
  NEED(c);
  if (n > 0) hipLaunchKernelGGL(k_mul, egrid(n), dim3(EB), 0, c->stream, n, a, b, y, accumulate);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_relu(vv_ctx* c, int64_t n, const float* x, float* y, float negative_slope) {
  NEED(c);
  if (n > 0) hipLaunchKernelGGL(k_relu, egrid(n), dim3(EB), 0, c->stream, n, x, y, negative_slope);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_relu_bwd(vv_ctx* c, int64_t n, const float* x, const float* dy, float* dx, float negative_slope) {
  NEED(c);
  if (n > 0) hipLaunchKernelGGL(k_relu_bwd, egrid(n), dim3(EB), 0, c->stream, n, x, dy, dx, negative_slope);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_dropout(vv_ctx* c, int64_t n, const float* x, float* y, uint8_t* mask, float ratio, uint64_t seed, int make_mask) {
  NEED(c);
  if (!(ratio >= 0.f && ratio < 1.f)) return vv_fail(VV_ERR_ARG, "vv_op_dropout: dropout_ratio must be in [0,1) (dropout_layer.cpp:17-19)");
  if (n > 0) hipLaunchKernelGGL(k_dropout, egrid(n), dim3(EB), 0, c->stream, n, x, y, mask, ratio, 1.f / (1.f - ratio), seed, make_mask);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_rowsum(vv_ctx* c, int64_t rows, int32_t cols, const float* x, int32_t num_output, float* y) {
  NEED(c);
  if (rows > 0) hipLaunchKernelGGL(k_rowsum, dim3((unsigned)((rows + 3) / 4)), dim3(EB), 0, c->stream, rows, cols, x, num_output, y);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_rowsum_bwd(vv_ctx* c, int64_t rows, int32_t cols, int32_t num_output, const float* dy, float* dx) {
  NEED(c);
  if (rows > 0) hipLaunchKernelGGL(k_rowsum_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(EB), 0, c->stream, rows, cols, num_output, dy, dx);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_normalize(vv_ctx* c, int64_t rows, int32_t cols, const float* x, float* y) {
  NEED(c);
  if (rows > 0) hipLaunchKernelGGL(k_normalize, dim3((unsigned)((rows + 3) / 4)), dim3(EB), 0, c->stream, rows, cols, x, y);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_normalize_bwd(vv_ctx* c, int64_t rows, int32_t cols, const float* x, const float* dy, float* dx) {
  NEED(c);
  if (rows > 0) hipLaunchKernelGGL(k_normalize_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(EB), 0, c->stream, rows, cols, x, dy, dx);
  HIPCHK(hipGetLastError());
  return VV_OK;
}
int vv_op_max_margin(vv_ctx* c, int32_t count, const float* s_true, const float* s_bogus, const float* weight, float margin,
                     int32_t norm, float* loss, float* violations) {
  NEED(c);
  if (count < 1 || (norm != VV_NORM_L1 && norm != VV_NORM_L2)) return vv_fail(VV_ERR_ARG, "vv_op_max_margin: bad count / Unknown Norm");
  OpScratch& s = scratch_of(c);
  if (!s.loss2) HIPCHK(hipMalloc(&s.loss2, 2 * sizeof(float)));
  hipLaunchKernelGGL(k_max_margin, dim3(1), dim3(EB), 0, c->stream, count, s_true, s_bogus, weight, margin, norm, s.loss2);
  float h[2];
  HIPCHK(hipMemcpyAsync(h, s.loss2, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (loss) *loss = h[0];
  if (violations) *violations = h[1];
  return VV_OK;
}
int vv_op_max_margin_bwd(vv_ctx* c, int32_t count, const float* s_true, const float* s_bogus, const float* weight, float margin,
                         int32_t norm, float loss_weight, float* d_true, float* d_bogus) {
  NEED(c);
  if (count < 1 || (norm != VV_NORM_L1 && norm != VV_NORM_L2)) return vv_fail(VV_ERR_ARG, "vv_op_max_margin_bwd: bad count / Unknown Norm");
  hipLaunchKernelGGL(k_max_margin_bwd, egrid(count), dim3(EB), 0, c->stream, count, s_true, s_bogus, weight, margin, norm, loss_weight, d_true, d_bogus);
  HIPCHK(hipGetLastError());
  return VV_OK;
}

int vv_op_gather_rows(vv_ctx* c, const int32_t* idx, int64_t n, float* out) {
  NEED(c);
  if (!c->table) return vv_fail(VV_ERR_STATE, "vv_op_gather_rows: no feature table");
  if (!idx || n <= 0 || !out) return vv_fail(VV_ERR_ARG, "vv_op_gather_rows: bad argument");
  std::vector<int32_t> h(idx, idx + n);
  for (int64_t i = 0; i < n; ++i) {
    if (h[i] < -1 || h[i] >= c->n_rows) return vv_fail(VV_ERR_ARG, "vv_op_gather_rows: idx[%lld] = %d out of range", (long long)i, h[i]);
    if (h[i] < 0) h[i] = (int32_t)c->n_rows;                       // the all-zero row
  }
  DevTmp<int32_t> d;
  HIPCHK(d.alloc((size_t)n));
  HIPCHK(hipMemcpyAsync(d, h.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
  launch_table_read(c->prec, c->table, d, n, c->F, c->Fp, 1.f / c->sx, out, c->stream);
  HIPCHK(hipStreamSynchronize(c->stream));
  return VV_OK;
}

// InnerProductLayer::Forward_gpu (inner_product_layer.cu:12-27) with the context's parameters: Y = X W^T + b.
// X is rounded to the MFMA operand type exactly as the feature table is, and multiplied by the forward kernel of the
// fused path (identity row index), so the per-layer path and the fused plan compute the same numbers.
int vv_op_inner_product(vv_ctx* c, const float* X, int64_t R, float* Y) {
  NEED(c);
  if (!c->W || !c->table) return vv_fail(VV_ERR_STATE, "vv_op_inner_product: table (it defines F) and parameters must be set first");
  if (!X || !Y || R <= 0 || R > (1ll << 30)) return vv_fail(VV_ERR_ARG, "vv_op_inner_product: bad argument");
  { const int rcj = vv_comm_join(c); if (rcj) return rcj; }
  OpScratch& s = scratch_of(c);
  const int64_t Rp = round_up(R, R_ALIGN);
  if (Rp > s.x_rows || s.Fp != c->Fp || s.Dp != c->Dp) {     // (new parameters / a new table on the same context: other widths)
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s.x16) (void)hipFree(s.x16);
    if (s.ident) (void)hipFree(s.ident);
    if (s.dy16) (void)hipFree(s.dy16);
    s.x16 = nullptr; s.ident = nullptr; s.dy16 = nullptr;
    HIPCHK(hipMalloc(&s.x16, (size_t)(Rp + 1) * c->Fp * 2));
    HIPCHK(hipMalloc(&s.ident, (size_t)Rp * 4));
    HIPCHK(hipMalloc(&s.dy16, (size_t)(Rp + BK) * c->Dp * 2));
    s.x_rows = Rp; s.Fp = c->Fp; s.Dp = c->Dp;
  }
  HIPCHK(hipMemsetAsync(s.x16, 0, (size_t)(Rp + 1) * c->Fp * 2, c->stream));
  launch_table_convert(c->prec, X, s.x16, R, c->F, c->Fp, c->sx, c->stream);
  hipLaunchKernelGGL(k_iota, dim3((unsigned)((Rp + EB - 1) / EB)), dim3(EB), 0, c->stream, s.ident, (int)R, (int)Rp, (int32_t)Rp);
  FwdArgs fa;
  fa.table = s.x16; fa.rows = s.ident; fa.Wh = c->Wh; fa.bias = c->b; fa.scales = c->scales;
  fa.H = Y; fa.R = (int)R; fa.D = c->D; fa.Fp = c->Fp; fa.relu = 0; fa.zero_row = (int32_t)Rp;
  fa.drop_ratio = 0.f; fa.mask = nullptr; fa.drop_seed = 0; fa.B = 1; fa.CN = 1;
  launch_fwd_gemm(c->prec, fa, c->stream);
  HIPCHK(hipGetLastError());
  return VV_OK;
}

// InnerProductLayer::Backward_gpu (inner_product_layer.cu:29-59): dW = dY^T X (overwritten, beta = 0), scaled by
// 1 + regularization / 2 when set (:36-42 as patched in inner_product_layer.cpp:80-90), db = dY^T 1, into the context's flat
// gradient buffer.  X must be the bottom of the preceding vv_op_inner_product call (its 16-bit copy is reused).
// Propagation to the bottom (dX = dY W, :51-57) is not built: the fc layer of this path sits on the data layer.
int vv_op_inner_product_bwd(vv_ctx* c, const float* dY, int64_t R, float ip_regularization) {
  NEED(c);
  OpScratch& s = scratch_of(c);
  const int64_t Rp = round_up(R, R_ALIGN);
  if (!dY || R <= 0 || !s.x16 || Rp > s.x_rows || s.Fp != c->Fp || s.Dp != c->Dp) return vv_fail(VV_ERR_STATE, "vv_op_inner_product_bwd: call vv_op_inner_product on the same rows first");
  const int D = c->D, Dp = c->Dp;
  // half-precision gradient scale: a power of two picked on the device from max |dY| itself (the caller's gradients are
  // arbitrary: no value may reach the product clipped), exact to undo
  if (!s.sgs) HIPCHK(hipMalloc(&s.sgs, 2 * sizeof(float)));
  HIPCHK(hipMemsetAsync(s.sgs, 0, sizeof(float), c->stream));
  launch_absmax(dY, R * D, (unsigned*)s.sgs, c->stream);
  hipLaunchKernelGGL(k_pick_scale, dim3(1), dim3(1), 0, c->stream, (const unsigned*)s.sgs, c->prec, s.sgs + 1);
  HIPCHK(hipMemsetAsync(s.dy16, 0, (size_t)(Rp + BK) * Dp * 2, c->stream));
  if (c->prec == 0) hipLaunchKernelGGL(k_to_half_rows<F16>, egrid(R * Dp), dim3(EB), 0, c->stream, dY, s.dy16, R, D, Dp, (const float*)(s.sgs + 1));
  else hipLaunchKernelGGL(k_to_half_rows<BF16>, egrid(R * Dp), dim3(EB), 0, c->stream, dY, s.dy16, R, D, Dp, (const float*)(s.sgs + 1));
  const int tiles = (Dp / BM) * (c->Fp / BN);
  const int total_steps = (int)(Rp / BK);
  int S = std::max(1, std::min((256 + tiles - 1) / tiles, total_steps));
  const size_t need = (size_t)S * (size_t)slab_pitch(Dp, c->Fp) * 4;
  if (need > s.slab_bytes) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (s.slabs) (void)hipFree(s.slabs);
    s.slabs = nullptr;
    HIPCHK(hipMalloc(&s.slabs, need));
    s.slab_bytes = need;
  }
  WgradArgs wa;
  wa.dYh = s.dy16; wa.table = s.x16; wa.rows = s.ident; wa.slabs = s.slabs;
  wa.Rp = (int)Rp; wa.Dp = Dp; wa.Fp = c->Fp; wa.S = S; wa.ksteps_per_split = (total_steps + S - 1) / S;
  wa.n_dev = nullptr; wa.zero_row = (int32_t)Rp;
  launch_wgrad_gemm(c->prec, wa, c->stream);
  c->red_lazy = false; c->grads_stale = false;     // (a lazily kept gradient of the fused step is superseded by this one)
  ReduceArgs ra;
  ra.slabs = s.slabs; ra.S = S; ra.Dp = Dp; ra.Fp = c->Fp; ra.dbp = nullptr; ra.B = 0;
  ra.scales = c->scales; ra.sg = 1.f; ra.sg_dev = s.sgs + 1; ra.grads = c->grads; ra.D = D; ra.F = c->F;
  ra.ip_scale = ip_regularization > 0.f ? 1.f + ip_regularization * 0.5f : 1.f;
  ra.loss_part = nullptr; ra.viol_part = nullptr; ra.loss_scale = 0.f; ra.loss_out = nullptr;
  ra.parts = 1;
  launch_reduce(ra, c->stream);
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)((D + 63) / 64)), dim3(EB), 0, c->stream, dY, R, D, c->grads + (size_t)D * c->F);
  HIPCHK(hipGetLastError());
  c->have_fwd = true;                      // gradients exist: vv_apply_update / vv_grads_get may follow
  c->grads_pending = c->comm != nullptr;
  c->grads_chunked = false;
  return VV_OK;
}

}  // extern "C"
