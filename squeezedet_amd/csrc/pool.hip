// NHWC max-pool for gfx950, replacing ModelSkeleton._pooling_layer
// (reference src/nn_skeleton.py:565-586): tf.nn.max_pool with TF SAME/VALID semantics --
// SAME-padded cells never win (they are skipped, which equals -inf padding).
// Pure HBM streaming: one thread owns one output pixel x 16 bytes of channels.
#include "common.h"

namespace sqdet {

template <typename T> struct PoolTr;
template <> struct PoolTr<f16> { static constexpr int V = 8; typedef f16x8 vec; };
template <> struct PoolTr<float> { static constexpr int V = 4; typedef f32x4 vec; };

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                      int C, int k, int stride, int pt, int pl, int Ho, int Wo) {
  constexpr int V = PoolTr<T>::V;
  typedef typename PoolTr<T>::vec vec;
  const int cv = C / V;
  const size_t total = (size_t)N * Ho * Wo * cv;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv);
    size_t p = idx / cv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int y0 = oy * stride - pt, x0 = ox * stride - pl;
    vec m;
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = (T)(-__builtin_huge_valf());
    for (int dy = 0; dy < k; ++dy) {
      const int iy = y0 + dy;
      if (iy < 0 || iy >= H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int ix = x0 + dx;
        if (ix < 0 || ix >= W) continue;
        const vec v = *reinterpret_cast<const vec*>(x + (((size_t)n * H + iy) * W + ix) * C + c * V);
#pragma unroll
        for (int e = 0; e < V; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
      }
    }
    *reinterpret_cast<vec*>(y + idx * V) = m;
  }
}

// 3x3 window, any stride (every pool of the reference's nets): the nine 16-byte loads of an output are
// issued back to back from clamped addresses and only then reduced -- with a run-time window size the
// loop above waits for each load before the next max, i.e. pays nine memory latencies per output.
template <typename T>
__device__ __forceinline__ typename PoolTr<T>::vec vmax(typename PoolTr<T>::vec a, typename PoolTr<T>::vec b) {
  return __builtin_elementwise_max(a, b);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                       int C, int stride, int pt, int pl, int Ho, int Wo) {
  constexpr int V = PoolTr<T>::V;
  typedef typename PoolTr<T>::vec vec;
  const int cv = C / V;
  const size_t total = (size_t)N * Ho * Wo * cv;
  // XCD-aware order: workgroup b runs on XCD b % 8 (each XCD has its own L2).  Neighbouring output rows share
  // an input row, so every XCD gets a CONTIGUOUS band of the output instead of every 8th 256-thread slice --
  // otherwise the shared rows are fetched once per XCD (measured 1.47x the algorithmic bytes at the fabric).
  const unsigned per = (gridDim.x + 7) / 8;
  const size_t slice = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  {
    const size_t idx = slice * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cv);
    size_t p = idx / cv;
    const int ox = (int)(p % Wo); p /= Wo;
    const int oy = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const int y0 = oy * stride - pt, x0 = ox * stride - pl;
    vec v[9];
    bool ok[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = y0 + t / 3, ix = x0 + t % 3;
      ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);   // clamped: a valid address, masked below
      v[t] = *reinterpret_cast<const vec*>(x + (((size_t)n * H + cy) * W + cx) * C + c * V);
    }
    vec m;
#pragma unroll
    for (int e = 0; e < V; ++e) m[e] = (T)(-__builtin_huge_valf());
#pragma unroll
    for (int t = 0; t < 9; ++t) m = ok[t] ? vmax<T>(m, v[t]) : m;
    *reinterpret_cast<vec*>(y + idx * V) = m;
  }
}

// The training forward's pool: maxpool3_kernel that also records WHICH cell won -- 3 * (window row) + (window column) of the
// FIRST maximum in row-major order (tf.nn.max_pool's gradient convention, the rule maxpool3s2_bwd_kernel re-derives from
// x), one byte per output element -- so that the backward pass reads idx + y + dy (a quarter-size map each) instead of
// searching the full-resolution input again.
template <typename T>
__global__ __launch_bounds__(256) void maxpool3_idx_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                           unsigned char* __restrict__ widx, int N, int H, int W, int C,
                                                           int stride, int pt, int pl, int Ho, int Wo) {
  constexpr int V = PoolTr<T>::V;
  typedef typename PoolTr<T>::vec vec;
  const int cv = C / V;
  const size_t total = (size_t)N * Ho * Wo * cv;
  const unsigned per = (gridDim.x + 7) / 8;
  const size_t slice = (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3);
  const size_t idx = slice * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % cv);
  size_t p = idx / cv;
  const int ox = (int)(p % Wo); p /= Wo;
  const int oy = (int)(p % Ho);
  const int n = (int)(p / Ho);
  const int y0 = oy * stride - pt, x0 = ox * stride - pl;
  vec v[9];
  bool ok[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int iy = y0 + t / 3, ix = x0 + t % 3;
    ok[t] = iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
    v[t] = *reinterpret_cast<const vec*>(x + (((size_t)n * H + cy) * W + cx) * C + c * V);
  }
  vec m;
  unsigned char pos[V];
#pragma unroll
  for (int e = 0; e < V; ++e) { m[e] = (T)(-__builtin_huge_valf()); pos[e] = 255; }
  // VALUE: the maximum starts at -inf and is replaced on a strict '>' only -- a NaN never wins, wherever it sits, exactly as in
  // maxpool_kernel / maxpool3_kernel (the training forward's pooled tensor is bitwise the inference pool's on ANY input).
  // POSITION: the first VALID cell names the window until a cell wins the comparison, so a window whose valid cells are all
  // -inf / NaN still names a cell -- the first one, as the x-searching backward kernels do.
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < V; ++e)
      if (ok[t]) {
        if (pos[e] == 255) pos[e] = (unsigned char)t;
        if (v[t][e] > m[e]) { m[e] = v[t][e]; pos[e] = (unsigned char)t; }
      }
  *reinterpret_cast<vec*>(y + idx * V) = m;
  if constexpr (V == 8) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 o;
    o[0] = pos[0] | (pos[1] << 8) | (pos[2] << 16) | ((unsigned)pos[3] << 24);
    o[1] = pos[4] | (pos[5] << 8) | (pos[6] << 16) | ((unsigned)pos[7] << 24);
    *reinterpret_cast<u32x2*>(widx + idx * V) = o;
  } else {
    *reinterpret_cast<unsigned int*>(widx + idx * V) = pos[0] | (pos[1] << 8) | (pos[2] << 16) | ((unsigned)pos[3] << 24);
  }
}

int maxpool_idx_launch(const void* x, void* y, unsigned char* widx, int n, int h, int w, int c, int k, int stride, int pad_mode,
                       int dtype, hipStream_t st) {
  SQDET_REQUIRE(x && y && widx, "maxpool_idx: null pointer");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "maxpool_idx: bad dtype %d", dtype);
  SQDET_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && stride > 0, "maxpool_idx: bad dims");
  SQDET_REQUIRE(pad_mode == SQDET_PAD_SAME || pad_mode == SQDET_PAD_VALID, "maxpool_idx: bad pad_mode");
  SQDET_UNSUPPORTED(k != 3, "maxpool_idx: 3x3 windows only (every pool of the reference's nets)");
  SQDET_REQUIRE(pad_mode == SQDET_PAD_SAME || (h >= k && w >= k), "maxpool_idx: VALID needs h,w >= k");
  const int V = dtype == SQDET_F16 ? 8 : 4;
  SQDET_UNSUPPORTED(c % V != 0, "maxpool_idx: channels %d not a multiple of %d", c, V);
  const int Ho = out_size(h, k, stride, pad_mode), Wo = out_size(w, k, stride, pad_mode);
  const int pt = pad_before(h, k, stride, pad_mode), pl = pad_before(w, k, stride, pad_mode);
  const size_t total = (size_t)n * Ho * Wo * (c / V);
  const size_t blocks = ((total + 255) / 256 + 7) / 8 * 8;
  SQDET_UNSUPPORTED(blocks > 0x7fffffffULL, "maxpool_idx: too many outputs");
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(maxpool3_idx_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16*)x, (f16*)y, widx, n, h,
                       w, c, stride, pt, pl, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool3_idx_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)y, widx,
                       n, h, w, c, stride, pt, pl, Ho, Wo);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

int maxpool_launch(const void* x, void* y, int n, int h, int w, int c, int k, int stride, int pad_mode, int dtype,
                   hipStream_t st) {
  SQDET_REQUIRE(x && y, "maxpool: null pointer");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "maxpool: bad dtype %d", dtype);
  SQDET_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && stride > 0, "maxpool: bad dims");
  SQDET_REQUIRE(pad_mode == SQDET_PAD_SAME || pad_mode == SQDET_PAD_VALID, "maxpool: bad pad_mode");
  SQDET_REQUIRE(pad_mode == SQDET_PAD_SAME || (h >= k && w >= k), "maxpool: VALID needs h,w >= k");
  const int V = dtype == SQDET_F16 ? 8 : 4;
  SQDET_UNSUPPORTED(c % V != 0, "maxpool: channels %d not a multiple of %d", c, V);
  const int Ho = out_size(h, k, stride, pad_mode), Wo = out_size(w, k, stride, pad_mode);
  const int pt = pad_before(h, k, stride, pad_mode), pl = pad_before(w, k, stride, pad_mode);
  const size_t total = (size_t)n * Ho * Wo * (c / V);
  size_t blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (k == 3) {
    blocks = ((total + 255) / 256 + 7) / 8 * 8;   // one 256-thread slice per workgroup, a multiple of 8 workgroups
    SQDET_UNSUPPORTED(blocks > 0x7fffffffULL, "maxpool: too many outputs");
    if (dtype == SQDET_F16)
      hipLaunchKernelGGL(maxpool3_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16*)x, (f16*)y, n, h,
                         w, c, stride, pt, pl, Ho, Wo);
    else
      hipLaunchKernelGGL(maxpool3_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)y,
                         n, h, w, c, stride, pt, pl, Ho, Wo);
  } else if (dtype == SQDET_F16)
    hipLaunchKernelGGL(maxpool_kernel<f16>, dim3((unsigned)blocks), dim3(256), 0, st, (const f16*)x, (f16*)y, n, h, w,
                       c, k, stride, pt, pl, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)y, n,
                       h, w, c, k, stride, pt, pl, Ho, Wo);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

}  // namespace sqdet

extern "C" int sqdet_maxpool_nhwc_fwd(const void* x, void* y, int n, int h, int w, int c, int k, int stride,
                                      int pad_mode, int dtype, sqdet_stream_t stream) {
  return sqdet::maxpool_launch(x, y, n, h, w, c, k, stride, pad_mode, dtype, sqdet::as_stream(stream));
}

extern "C" int sqdet_maxpool_nhwc_fwd_idx(const void* x, void* y, unsigned char* window_index, int n, int h, int w, int c, int k,
                                          int stride, int pad_mode, int dtype, sqdet_stream_t stream) {
  return sqdet::maxpool_idx_launch(x, y, window_index, n, h, w, c, k, stride, pad_mode, dtype, sqdet::as_stream(stream));
}
