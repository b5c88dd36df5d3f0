// Frozen-statistics batch norm of ModelSkeleton._conv_bn_layer (reference src/nn_skeleton.py:374-468)
// folded into the conv it follows.  tf.nn.batch_normalization computes
//   inv = rsqrt(var + eps) * gamma;  y = x * inv + (beta - mean * inv)
// with x = conv2d(in, W) [+ biases]; mean / var are non-trainable variables (:437-438), so the whole
// thing is the conv with W[..., c] * inv[c] and bias (biases[c] - mean[c]) * inv[c] + beta[c].
// A load-time transform: one pass over the kernel, HBM-streaming, one thread per 4 output channels.
#include "common.h"

namespace sqdet {

// w / wf may alias (in-place fold): every thread reads and writes its own 4 elements only.
__global__ __launch_bounds__(256) void fold_bn_kernel(const float* w, const float* __restrict__ cbias,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                      float eps, float* wf, float* __restrict__ bf,
                                                      size_t rows, int cout) {
  const int cv = cout / 4;
  const size_t total = rows * cv;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv) * 4;
    f32x4 inv;
#pragma unroll
    for (int e = 0; e < 4; ++e) inv[e] = gamma[c + e] / sqrtf(var[c + e] + eps);
    const f32x4 v = *reinterpret_cast<const f32x4*>(w + idx * 4);
    *reinterpret_cast<f32x4*>(wf + idx * 4) = v * inv;
    if (idx < (size_t)cv) {  // first kernel row's threads also produce the folded bias
#pragma unroll
      for (int e = 0; e < 4; ++e) bf[c + e] = ((cbias ? cbias[c + e] : 0.f) - mean[c + e]) * inv[e] + beta[c + e];
    }
  }
}

int fold_bn_launch(const float* w, const float* cbias, const float* gamma, const float* beta, const float* mean,
                   const float* var, float eps, float* wf, float* bf, int k, int cin, int cout, hipStream_t st) {
  SQDET_REQUIRE(w && gamma && beta && mean && var && wf && bf, "fold_batchnorm: null pointer");
  SQDET_REQUIRE(k > 0 && cin > 0 && cout > 0 && eps >= 0.f, "fold_batchnorm: bad dims");
  SQDET_UNSUPPORTED(cout % 4 != 0, "fold_batchnorm: cout %d not a multiple of 4", cout);
  const size_t rows = (size_t)k * k * cin;
  const size_t total = rows * (cout / 4);
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fold_bn_kernel, dim3((unsigned)blocks), dim3(256), 0, st, w, cbias, gamma, beta, mean, var, eps, wf,
                     bf, rows, cout);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

// Backward of the fold (training of the _conv_bn_layer convs, float32): given the gradients of the
// FOLDED kernel / bias (what the conv backward kernels produce), the gradients of the variables:
//   Wf = W * gamma * r,  bf = (cb - mean) * gamma * r + beta,   r = 1/sqrt(var + eps)
//   dW = dWf * gamma * r;   dgamma = r * (sum_rows(dWf * W) + (cb - mean) * dbf);   dbeta = dbf
// Two kernels, deterministic.  (1) A workgroup owns 64 output channels x FB_ROWS rows of the [k*k*cin, cout] matrix:
// thread (row partition rp = tid/64, channel) walks its rows rp, rp+4, ..., writes dW and leaves the column sum of
// dWf * W over the block in partial[row block][channel] (the four partitions summed in fixed order through LDS).
// (2) One thread per channel adds the row blocks' partials in order.  (The first version walked ALL rows in cout/64
// workgroups -- 4 for a 256-channel conv -- and was 29 % of the ResNet50 mixed-precision step.)
constexpr int FB_ROWS = 32;

__global__ __launch_bounds__(256) void fold_bn_bwd_kernel(const float* __restrict__ w, const float* dwf,
                                                          const float* __restrict__ gamma, const float* __restrict__ var,
                                                          float eps, float* dw, float* __restrict__ partial, int rows, int cout) {
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * FB_ROWS;
  float s = 0.f;
  if (c < cout) {
    const float inv = gamma[c] * (1.f / sqrtf(var[c] + eps));
    const int r1 = min(rows, r0 + FB_ROWS);
    for (int row = r0 + rp; row < r1; row += 4) {
      const size_t i = (size_t)row * cout + c;
      const float g = dwf[i];
      s += g * w[i];
      dw[i] = g * inv;   // dw may alias dwf: each element is read, then written, by this thread only
    }
  }
  part[rp][cl] = s;
  __syncthreads();
  if (rp == 0 && c < cout) partial[(size_t)blockIdx.y * cout + c] = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
}

__global__ void fold_bn_bwd_finish_kernel(const float* __restrict__ partial, const float* __restrict__ dbf,
                                          const float* __restrict__ cbias, const float* __restrict__ mean,
                                          const float* __restrict__ var, float eps, float* __restrict__ dgamma,
                                          float* __restrict__ dbeta, int nblocks, int cout) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cout) return;
  float tot = 0.f;
  for (int b = 0; b < nblocks; ++b) tot += partial[(size_t)b * cout + c];
  const float r = 1.f / sqrtf(var[c] + eps);
  const float db = dbf[c];
  dgamma[c] = r * (tot + ((cbias ? cbias[c] : 0.f) - mean[c]) * db);
  dbeta[c] = db;
}

size_t fold_bn_bwd_workspace_bytes(int k, int cin, int cout) {
  const long rows = (long)k * k * cin;
  return (size_t)((rows + FB_ROWS - 1) / FB_ROWS) * cout * sizeof(float);
}

int fold_bn_bwd_launch(const float* w, const float* dwf, const float* dbf, const float* cbias, const float* gamma,
                       const float* mean, const float* var, float eps, float* dw, float* dgamma, float* dbeta,
                       float* workspace, int k, int cin, int cout, hipStream_t st) {
  SQDET_REQUIRE(w && dwf && dbf && gamma && mean && var && dw && dgamma && dbeta && workspace, "fold_batchnorm_bwd: null pointer");
  SQDET_REQUIRE(k > 0 && cin > 0 && cout > 0 && eps >= 0.f, "fold_batchnorm_bwd: bad dims");
  const int rows = k * k * cin, nblocks = (rows + FB_ROWS - 1) / FB_ROWS;
  hipLaunchKernelGGL(fold_bn_bwd_kernel, dim3((unsigned)((cout + 63) / 64), (unsigned)nblocks), dim3(256), 0, st, w, dwf, gamma,
                     var, eps, dw, workspace, rows, cout);
  SQDET_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(fold_bn_bwd_finish_kernel, dim3((unsigned)((cout + 255) / 256)), dim3(256), 0, st, workspace, dbf, cbias,
                     mean, var, eps, dgamma, dbeta, nblocks, cout);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

// The same for MANY _conv_bn_layer convs in two launches (a ResNet50 step has 19 trainable ones: 38 launches of 5-10 us
// behind as many slab reductions): a table entry per conv, workgroup b of launch (1) serves the entry whose range holds b
// with fold_bn_bwd_kernel's block shape and arithmetic, launch (2) likewise with fold_bn_bwd_finish_kernel's -- per conv the
// results are bitwise sqdet_fold_batchnorm_bwd's.
struct FoldBwdItem {
  const float *w, *dwf, *dbf, *cbias, *gamma, *mean, *var;
  float *dw, *dgamma, *dbeta, *partial;
  int rows, cout, nblocks_y, blocks_x;
  unsigned first_block, first_finish;     // launch (1): blocks_x * nblocks_y workgroups; launch (2): ceil(cout / 256)
};

__global__ __launch_bounds__(256) void fold_bn_bwd_many_kernel(const FoldBwdItem* __restrict__ items, int n, float eps) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FoldBwdItem it = items[lo];
  const unsigned local = blockIdx.x - it.first_block;
  const int bx = (int)(local % (unsigned)it.blocks_x), by = (int)(local / (unsigned)it.blocks_x);
  __shared__ float part[4][64];
  const int cl = threadIdx.x & 63, rp = threadIdx.x >> 6;
  const int c = bx * 64 + cl;
  const int r0 = by * FB_ROWS;
  float s = 0.f;
  if (c < it.cout) {
    const float inv = it.gamma[c] * (1.f / sqrtf(it.var[c] + eps));
    const int r1 = min(it.rows, r0 + FB_ROWS);
    for (int row = r0 + rp; row < r1; row += 4) {
      const size_t i = (size_t)row * it.cout + c;
      const float g = it.dwf[i];
      s += g * it.w[i];
      it.dw[i] = g * inv;
    }
  }
  part[rp][cl] = s;
  __syncthreads();
  if (rp == 0 && c < it.cout) it.partial[(size_t)by * it.cout + c] = ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl];
}

__global__ void fold_bn_bwd_finish_many_kernel(const FoldBwdItem* __restrict__ items, int n, float eps) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_finish <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const FoldBwdItem it = items[lo];
  const int c = (int)(blockIdx.x - it.first_finish) * blockDim.x + threadIdx.x;
  if (c >= it.cout) return;
  float tot = 0.f;
  for (int b = 0; b < it.nblocks_y; ++b) tot += it.partial[(size_t)b * it.cout + c];
  const float r = 1.f / sqrtf(it.var[c] + eps);
  const float db = it.dbf[c];
  it.dgamma[c] = r * (tot + ((it.cbias ? it.cbias[c] : 0.f) - it.mean[c]) * db);
  it.dbeta[c] = db;
}

// y[n, oy, ox, :] = x[n, oy*stride, ox*stride, :]: the pixels a 1x1 / stride-s SAME conv reads (the
// projection shortcut and branch2a of res3a / res4a, resnet50_convDet.py:71-73,150-156), gathered so the
// stride-1 filter-gradient kernel can be used on them.  16 bytes per thread.
__global__ __launch_bounds__(256) void subsample_kernel(const i32x4* __restrict__ x, i32x4* __restrict__ y, int N, int H,
                                                        int W, int cv, int stride, int Ho, int Wo) {
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv);
    size_t p = idx / cv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    y[idx] = x[(((size_t)n * H + (size_t)oy * stride) * W + (size_t)ox * stride) * cv + c];
  }
}

int subsample_launch(const void* x, void* y, int n, int h, int w, int c, int stride, int dtype, hipStream_t st) {
  SQDET_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && stride > 0, "subsample: bad arguments");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "subsample: bad dtype %d", dtype);
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  SQDET_UNSUPPORTED((c * esz) % 16 != 0, "subsample: channel bytes %d not a multiple of 16", c * esz);
  const int Ho = (h + stride - 1) / stride, Wo = (w + stride - 1) / stride, cv = c * esz / 16;
  const size_t total = (size_t)n * Ho * Wo * cv;
  size_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(subsample_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const i32x4*)x, (i32x4*)y, n, h, w, cv,
                     stride, Ho, Wo);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

}  // namespace sqdet

extern "C" size_t sqdet_fold_batchnorm_bwd_workspace_bytes(int k, int cin, int cout) {
  return k > 0 && cin > 0 && cout > 0 ? sqdet::fold_bn_bwd_workspace_bytes(k, cin, cout) : 0;
}

extern "C" int sqdet_fold_batchnorm_bwd(const float* w_hwio, const float* dw_folded, const float* db_folded,
                                        const float* conv_bias, const float* gamma, const float* mean, const float* var,
                                        float eps, float* dw, float* dgamma, float* dbeta, float* workspace, int k, int cin,
                                        int cout, sqdet_stream_t stream) {
  return sqdet::fold_bn_bwd_launch(w_hwio, dw_folded, db_folded, conv_bias, gamma, mean, var, eps, dw, dgamma, dbeta,
                                   workspace, k, cin, cout, sqdet::as_stream(stream));
}

extern "C" int sqdet_subsample_nhwc(const void* x, void* y, int n, int h, int w, int c, int stride, int dtype,
                                    sqdet_stream_t stream) {
  return sqdet::subsample_launch(x, y, n, h, w, c, stride, dtype, sqdet::as_stream(stream));
}

extern "C" int sqdet_fold_batchnorm(const float* w_hwio, const float* conv_bias, const float* gamma, const float* beta,
                                    const float* mean, const float* var, float eps, float* w_folded, float* b_folded,
                                    int k, int cin, int cout, sqdet_stream_t stream) {
  return sqdet::fold_bn_launch(w_hwio, conv_bias, gamma, beta, mean, var, eps, w_folded, b_folded, k, cin, cout,
                               sqdet::as_stream(stream));
}

extern "C" size_t sqdet_fold_batchnorm_bwd_many_table_bytes(int n_items) {
  return n_items > 0 ? (size_t)n_items * sizeof(sqdet::FoldBwdItem) : 0;
}

extern "C" int sqdet_fold_batchnorm_bwd_many_prepare(const float* const* w_hwio, const float* const* dw_folded,
                                                     const float* const* db_folded, const float* const* conv_bias,
                                                     const float* const* gamma, const float* const* mean,
                                                     const float* const* var, float* const* dw, float* const* dgamma,
                                                     float* const* dbeta, float* const* workspace, const int* k, const int* cin,
                                                     const int* cout, int n_items, void* table_host, int* blocks,
                                                     int* finish_blocks) {
  SQDET_REQUIRE(w_hwio && dw_folded && db_folded && conv_bias && gamma && mean && var && dw && dgamma && dbeta && workspace && k &&
                    cin && cout && table_host && blocks && finish_blocks && n_items > 0, "fold_batchnorm_bwd_many_prepare: bad arguments");
  sqdet::FoldBwdItem* t = static_cast<sqdet::FoldBwdItem*>(table_host);
  unsigned nb = 0, nf = 0;
  for (int i = 0; i < n_items; ++i) {
    SQDET_REQUIRE(w_hwio[i] && dw_folded[i] && db_folded[i] && gamma[i] && mean[i] && var[i] && dw[i] && dgamma[i] && dbeta[i] &&
                      workspace[i] && k[i] > 0 && cin[i] > 0 && cout[i] > 0, "fold_batchnorm_bwd_many_prepare: bad item %d", i);
    sqdet::FoldBwdItem& it = t[i];
    it.w = w_hwio[i]; it.dwf = dw_folded[i]; it.dbf = db_folded[i]; it.cbias = conv_bias[i]; it.gamma = gamma[i];
    it.mean = mean[i]; it.var = var[i]; it.dw = dw[i]; it.dgamma = dgamma[i]; it.dbeta = dbeta[i]; it.partial = workspace[i];
    it.rows = k[i] * k[i] * cin[i]; it.cout = cout[i];
    it.nblocks_y = (it.rows + sqdet::FB_ROWS - 1) / sqdet::FB_ROWS;
    it.blocks_x = (cout[i] + 63) / 64;
    it.first_block = nb; it.first_finish = nf;
    nb += (unsigned)(it.blocks_x * it.nblocks_y);
    nf += (unsigned)((cout[i] + 255) / 256);
  }
  *blocks = (int)nb; *finish_blocks = (int)nf;
  return SQDET_OK;
}

extern "C" int sqdet_fold_batchnorm_bwd_many(const void* table_dev, int n_items, int blocks, int finish_blocks, float eps,
                                             sqdet_stream_t stream) {
  SQDET_REQUIRE(table_dev && n_items > 0 && blocks > 0 && finish_blocks > 0 && eps >= 0.f, "fold_batchnorm_bwd_many: bad arguments");
  const sqdet::FoldBwdItem* t = static_cast<const sqdet::FoldBwdItem*>(table_dev);
  hipLaunchKernelGGL(sqdet::fold_bn_bwd_many_kernel, dim3((unsigned)blocks), dim3(256), 0, sqdet::as_stream(stream), t, n_items, eps);
  SQDET_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(sqdet::fold_bn_bwd_finish_many_kernel, dim3((unsigned)finish_blocks), dim3(256), 0, sqdet::as_stream(stream), t,
                     n_items, eps);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
