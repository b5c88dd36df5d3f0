// Training-path kernels for gfx950 (float32 storage, the reference's training dtype):
//   conv backward-data   -- the forward conv kernel run on the 180-degree-rotated, in/out-swapped
//                           kernel (stride-1 SAME convs: every trainable conv of SqueezeDet);
//   conv backward-filter -- split-K GEMM over pixels on the exact-f32 MFMA, deterministic two-pass
//                           reduction (no atomics), + bias gradient;
//   ReLU / max-pool / dropout backward;
//   loss forward+backward (reference src/nn_skeleton.py:142-327: _add_interpretation_graph +
//                           _add_loss_graph) writing dL/dpreds;
//   Momentum update with per-variable clip_by_norm and weight decay (nn_skeleton.py:329-361).
// Compiled with -ffp-contract=off (the loss decode must match interpret_output op for op).
#include <math.h>

#include <vector>

#include "conv_common.h"

namespace sqdet {

int conv2d_launch_ex(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                     int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                     int x_cstride, int x_coffset, int accum, hipStream_t st);
int conv2d_launch_masked(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                         int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                         int x_cstride, int x_coffset, int accum, const void* relu_of, hipStream_t st);

// ------------------------------------------------------------------ backward-data weight packing
// dgrad(dy)[ci] = sum_{tap,co} W[k-1-ty][k-1-tx][ci][co] * dy@tap[co]: a forward conv with
// kernel W'[ty][tx][co][ci].  Packed in the forward fragment order for a [k,k,cout,cin] kernel.
template <typename T>
__global__ void pack_bwd_data_kernel(const float* __restrict__ w, T* __restrict__ out, int k, int cin, int cout,
                                     int nchunk, int nt, size_t total) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t t = idx;
    const int e = t % KG; t /= KG;
    const int lane = t % 64; t /= 64;
    const int n = t % nt; t /= nt;
    const int step = t % (k * k * nchunk); t /= (k * k * nchunk);
    const int group = (int)t;
    const int tap = step / nchunk, chunk = step - tap * nchunk;
    const int i = lane & 15, g = lane >> 4;
    const int o = group * 16 * nt + (i >> 2) * 4 * nt + n * 4 + (i & 3);   // output channel of the dgrad = original cin
    const int c = chunk * KC + g * KG + e;                                  // input channel of the dgrad = original cout
    const int ty = tap / k, tx = tap - ty * k;
    float v = 0.f;
    if (o < cin && c < cout) v = w[(((size_t)(k - 1 - ty) * k + (k - 1 - tx)) * cin + o) * cout + c];
    out[idx] = (T)v;
  }
}


// ------------------------------------------------------------------ packing MANY kernels in one launch
// A training step re-packs every trainable conv kernel twice (forward fragment order + the backward-data conv's order) after
// the optimizer has changed it: 62 launches of ~4.7 us in SqueezeDet's step -- 0.29 ms of 3.5.  One launch walks a table.
struct PackItem {
  const float* w;        // HWIO float32 [k,k,cin,cout]
  void* out;             // packed buffer
  int k, cin, cout;
  int taps, kdim, nchunk, nt;   // of the conv that will READ the packed kernel (forward: cin -> cout; backward-data: cout -> cin)
  int bwd;               // 0 forward order (pack_weights_kernel), 1 backward-data order (pack_bwd_data_kernel)
  unsigned long long total;
  int first_block, nblocks;
  // gamma != NULL: the kernel of a _conv_bn_layer -- what is packed is the FOLDED kernel W[..., c] * gamma[c] / sqrt(var[c] + eps)
  // (fold_bn_kernel's expression, bn.hip), and the item's first workgroup also writes the folded bias b_folded (when not NULL)
  const float *gamma, *var, *beta, *mean, *cbias;
  float* b_folded;
  float eps;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_many_kernel(const PackItem* __restrict__ items, int nitems) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  int it = 0;
  while (it + 1 < nitems && (int)blockIdx.x >= items[it + 1].first_block) ++it;     // (<= a few dozen entries)
  const PackItem p = items[it];
  const float* __restrict__ w = p.w;
  T* __restrict__ out = reinterpret_cast<T*>(p.out);
  for (size_t idx = (size_t)(blockIdx.x - p.first_block) * 256 + threadIdx.x; idx < p.total; idx += (size_t)p.nblocks * 256) {
    size_t t = idx;
    const int e = t % KG; t /= KG;
    const int lane = t % 64; t /= 64;
    const int n = t % p.nt; t /= p.nt;
    const int steps = p.taps * p.nchunk;
    const int step = t % steps; t /= steps;
    const int group = (int)t;
    const int tap = step / p.nchunk, chunk = step - tap * p.nchunk;
    const int i = lane & 15, g = lane >> 4;
    const int co = group * 16 * p.nt + (i >> 2) * 4 * p.nt + n * 4 + (i & 3);
    const int ci = chunk * KC + g * KG + e;
    float v = 0.f;
    int oc = -1;             // the ORIGINAL output channel of the element (the batch-norm channel)
    if (!p.bwd) {            // exactly pack_weights_kernel (conv.hip)
      if (co < p.cout && ci < p.kdim) { v = w[((size_t)tap * p.kdim + ci) * p.cout + co]; oc = co; }
    } else {                 // exactly pack_bwd_data_kernel: output channel of the dgrad = original cin, input = original cout
      const int ty = tap / p.k, tx = tap - ty * p.k;
      if (co < p.cin && ci < p.cout) { v = w[(((size_t)(p.k - 1 - ty) * p.k + (p.k - 1 - tx)) * p.cin + co) * p.cout + ci]; oc = ci; }
    }
    if (p.gamma && oc >= 0) v = v * (p.gamma[oc] / sqrtf(p.var[oc] + p.eps));
    out[idx] = (T)v;
  }
  if (p.gamma && p.b_folded && (int)blockIdx.x == p.first_block) {
    for (int c = threadIdx.x; c < p.cout; c += 256) {
      const float inv = p.gamma[c] / sqrtf(p.var[c] + p.eps);
      p.b_folded[c] = ((p.cbias ? p.cbias[c] : 0.f) - p.mean[c]) * inv + p.beta[c];
    }
  }
}

// (conv backward-filter lives in wgrad.hip)
// out[e] = sum_z partial[z][e]  (+ decay * w[e] when w != NULL), z ascending: deterministic.
__global__ void slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, const float* __restrict__ w,
                                   float decay, size_t count, int nslabs) {
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < nslabs; ++z) s += partial[(size_t)z * count + e];
    if (w) s += decay * w[e];
    out[e] = s;
  }
}

// ------------------------------------------------------------------ elementwise backward ops
// (float32, or float16 for mixed-precision training; 16-byte vectors of EV elements)
template <typename T> struct Vec16 { typedef T type __attribute__((ext_vector_type(16 / sizeof(T)))); };

template <typename T>
__global__ void relu_bwd_kernel(const T* __restrict__ y, T* __restrict__ dy, size_t nv) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const V v = reinterpret_cast<const V*>(y)[i];
    V g = reinterpret_cast<V*>(dy)[i];
#pragma unroll
    for (int e = 0; e < EV; ++e) g[e] = v[e] > (T)0 ? g[e] : (T)0;
    reinterpret_cast<V*>(dy)[i] = g;
  }
}

// y = x * mask * scale  (tf.nn.dropout forward with mask = floor(keep_prob + U), scale = 1/keep_prob; and its backward)
template <typename T>
__global__ void scale_mask_kernel(const T* __restrict__ x, const T* __restrict__ mask, T* __restrict__ y, float scale, size_t nv) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const V v = reinterpret_cast<const V*>(x)[i];
    const V m = reinterpret_cast<const V*>(mask)[i];
    V o;
#pragma unroll
    for (int e = 0; e < EV; ++e) o[e] = (T)((float)v[e] * (float)m[e] * scale);
    reinterpret_cast<V*>(y)[i] = o;
  }
}

// the same followed by the ReLU backward of the tensor x is the gradient of (relu_of = that ReLU's output): the dropout
// backward in front of conv12 and fire11's ReLU backward in one pass
template <typename T>
__global__ void scale_mask_relu_kernel(const T* __restrict__ x, const T* __restrict__ mask, const T* __restrict__ relu_of,
                                       T* __restrict__ y, float scale, size_t nv) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const V v = reinterpret_cast<const V*>(x)[i];
    const V m = reinterpret_cast<const V*>(mask)[i];
    const V r = reinterpret_cast<const V*>(relu_of)[i];
    V o;
#pragma unroll
    for (int e = 0; e < EV; ++e) {
      const T t = (T)((float)v[e] * (float)m[e] * scale);
      o[e] = r[e] > (T)0 ? t : (T)0;
    }
    reinterpret_cast<V*>(y)[i] = o;
  }
}

// dst = (D)(src * scale): the float16 <-> float32 hand-offs of mixed-precision training (preds -> loss, dpreds * loss_scale)
template <typename S, typename D>
__global__ void convert_scale_kernel(const S* __restrict__ src, D* __restrict__ dst, float scale, size_t n4) {
  typedef S SV __attribute__((ext_vector_type(4)));
  typedef D DV __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const SV v = reinterpret_cast<const SV*>(src)[i];
    DV o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (D)((float)v[e] * scale);
    reinterpret_cast<DV*>(dst)[i] = o;
  }
}

// max-pool backward for the nets' 3x3 / stride-2 pools, gather form (deterministic, no atomics).  In padded
// coordinates (u, v) = (iy + pt, ix + pl) window (oy, ox) covers u in [2oy, 2oy+2]; a thread owns the 2x2 cell block
// u in {2a, 2a+1}, v in {2b, 2b+1}, which only the four windows (a-1..a, b-1..b) touch: it finds each window's FIRST
// maximum (row-major scan, tf.nn.max_pool's argmax) once and hands dy to whichever of its cells that is -- 36 loads
// per block instead of the 81 of a per-cell search.  Windows are visited (a,b), (a,b-1), (a-1,b), (a-1,b-1): the
// summation order of the generic kernel below, so the two agree bitwise.
template <typename T>
__global__ void maxpool3s2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                      int N, int H, int W, int C, int pt, int pl, int Ho, int Wo, int BA, int BB, int relu) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  const int cvn = C / EV;
  const size_t total = (size_t)N * BA * BB * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % cvn);
    size_t p = idx / cvn;
    const int b = (int)(p % BB); p /= BB;
    const int a = (int)(p % BA);
    const int n = (int)(p / BA);
    float g[4][EV];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < EV; ++e) g[q][e] = 0.f;
#pragma unroll
    for (int wy = 0; wy < 2; ++wy)
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        const int oy = a - wy, ox = b - wx;
        if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
        float best[EV];
        int bpos[EV];   // 3 * (row in window) + (col in window)
#pragma unroll
        for (int e = 0; e < EV; ++e) { best[e] = -INFINITY; bpos[e] = -1; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int yy = 2 * oy - pt + r;
          if (yy < 0 || yy >= H) continue;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const int xx = 2 * ox - pl + c;
            if (xx < 0 || xx >= W) continue;
            const V v = *reinterpret_cast<const V*>(x + ((((size_t)n * H + yy) * W + xx) * C + cv * EV));
#pragma unroll
            for (int e = 0; e < EV; ++e) {
              // the first valid cell NAMES the window until a cell wins the strict comparison against -inf (a NaN never wins): a
              // window of -inf / NaN cells still routes its gradient, to that cell -- maxpool3_idx_kernel's rule, bit for bit
              if (bpos[e] < 0) bpos[e] = 3 * r + c;
              if ((float)v[e] > best[e]) { best[e] = (float)v[e]; bpos[e] = 3 * r + c; }
            }
          }
        }
        const V d = *reinterpret_cast<const V*>(dy + ((((size_t)n * Ho + oy) * Wo + ox) * C + cv * EV));
        // this block's cells inside window (oy, ox): window rows 2*wy + {0,1}, cols 2*wx + {0,1} (those < 3)
#pragma unroll
        for (int du = 0; du < 2; ++du)
#pragma unroll
          for (int dv = 0; dv < 2; ++dv) {
            const int r = 2 * wy + du, c = 2 * wx + dv;
            if (r > 2 || c > 2) continue;
#pragma unroll
            for (int e = 0; e < EV; ++e)
              if (bpos[e] == 3 * r + c) g[du * 2 + dv][e] += (float)d[e];
          }
      }
#pragma unroll
    for (int du = 0; du < 2; ++du)
#pragma unroll
      for (int dv = 0; dv < 2; ++dv) {
        const int iy = 2 * a + du - pt, ix = 2 * b + dv - pl;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        V o;
#pragma unroll
        for (int e = 0; e < EV; ++e) o[e] = (T)g[du * 2 + dv][e];
        if (relu) {   // x is a ReLU output: the ReLU backward of the layer below, here instead of a pass of its own
          const V xv = *reinterpret_cast<const V*>(x + ((((size_t)n * H + iy) * W + ix) * C + cv * EV));
#pragma unroll
          for (int e = 0; e < EV; ++e) o[e] = xv[e] > (T)0 ? o[e] : (T)0;
        }
        *reinterpret_cast<V*>(dx + ((((size_t)n * H + iy) * W + ix) * C + cv * EV)) = o;
      }
  }
}

// The same from the window index the training forward recorded (pool.hip maxpool3_idx_kernel): per 2x2 cell block four
// windows' idx (one byte per channel) + dy -- and, for the fused ReLU backward, the POOLED y: a cell that receives anything is
// the argmax of a window whose output y equals the cell's x, so (x > 0) == (y > 0) for every contribution it sums -- instead
// of 25 full-resolution 16-byte cells of x.  Same window order and the same sums as maxpool3s2_bwd_kernel: bitwise equal.
template <typename T>
__global__ void maxpool3s2_bwd_idx_kernel(const unsigned char* __restrict__ widx, const T* __restrict__ y,
                                          const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int pt,
                                          int pl, int Ho, int Wo, int BA, int BB, int relu) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  const int cvn = C / EV;
  const size_t total = (size_t)N * BA * BB * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % cvn);
    size_t p = idx / cvn;
    const int b = (int)(p % BB); p /= BB;
    const int a = (int)(p % BA);
    const int n = (int)(p / BA);
    float g[4][EV];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < EV; ++e) g[q][e] = 0.f;
#pragma unroll
    for (int wy = 0; wy < 2; ++wy)
#pragma unroll
      for (int wx = 0; wx < 2; ++wx) {
        const int oy = a - wy, ox = b - wx;
        if (oy < 0 || oy >= Ho || ox < 0 || ox >= Wo) continue;
        const size_t o = (((size_t)n * Ho + oy) * Wo + ox) * C + cv * EV;
        unsigned char bp[EV];
        if constexpr (EV == 8) {
          typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 w2 = *reinterpret_cast<const u32x2*>(widx + o);
#pragma unroll
          for (int e = 0; e < 8; ++e) bp[e] = (unsigned char)(w2[e >> 2] >> (8 * (e & 3)));
        } else {
          const unsigned int w1 = *reinterpret_cast<const unsigned int*>(widx + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) bp[e] = (unsigned char)(w1 >> (8 * e));
        }
        V d = *reinterpret_cast<const V*>(dy + o);
        if (relu) {
          const V yv = *reinterpret_cast<const V*>(y + o);
#pragma unroll
          for (int e = 0; e < EV; ++e) d[e] = yv[e] > (T)0 ? d[e] : (T)0;
        }
#pragma unroll
        for (int du = 0; du < 2; ++du)
#pragma unroll
          for (int dv = 0; dv < 2; ++dv) {
            const int r = 2 * wy + du, c = 2 * wx + dv;
            if (r > 2 || c > 2) continue;
#pragma unroll
            for (int e = 0; e < EV; ++e)
              if (bp[e] == 3 * r + c) g[du * 2 + dv][e] += (float)d[e];
          }
      }
#pragma unroll
    for (int du = 0; du < 2; ++du)
#pragma unroll
      for (int dv = 0; dv < 2; ++dv) {
        const int iy = 2 * a + du - pt, ix = 2 * b + dv - pl;
        if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
        V o;
#pragma unroll
        for (int e = 0; e < EV; ++e) o[e] = (T)g[du * 2 + dv][e];
        *reinterpret_cast<V*>(dx + ((((size_t)n * H + iy) * W + ix) * C + cv * EV)) = o;
      }
  }
}

// max-pool backward, generic gather form (deterministic): an input cell receives dy of every window whose
// FIRST maximum (row-major scan, as tf.nn.max_pool's argmax) it is.
template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                   int N, int H, int W, int C, int k, int stride, int pt, int pl, int Ho, int Wo, int relu) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  const int cvn = C / EV;
  const size_t total = (size_t)N * H * W * cvn;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cv = (int)(idx % cvn);
    size_t p = idx / cvn;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const int n = (int)(p / H);
    float g[EV];
#pragma unroll
    for (int e = 0; e < EV; ++e) g[e] = 0.f;
    // windows (oy, ox) containing (iy, ix): oy*stride - pt <= iy <= oy*stride - pt + k - 1
    const int oy_hi = (iy + pt) / stride, ox_hi = (ix + pl) / stride;
    for (int oy = oy_hi; oy >= 0 && oy * stride - pt + k - 1 >= iy; --oy) {
      if (oy >= Ho) continue;
      for (int ox = ox_hi; ox >= 0 && ox * stride - pl + k - 1 >= ix; --ox) {
        if (ox >= Wo) continue;
        // first argmax of the window, per channel: position code yy * W + xx
        float best[EV];
        int bpos[EV];
#pragma unroll
        for (int e = 0; e < EV; ++e) { best[e] = -INFINITY; bpos[e] = -1; }
        for (int dy_ = 0; dy_ < k; ++dy_) {
          const int yy = oy * stride - pt + dy_;
          if (yy < 0 || yy >= H) continue;
          for (int dx_ = 0; dx_ < k; ++dx_) {
            const int xx = ox * stride - pl + dx_;
            if (xx < 0 || xx >= W) continue;
            const V v = *reinterpret_cast<const V*>(x + ((((size_t)n * H + yy) * W + xx) * C + cv * EV));
#pragma unroll
            for (int e = 0; e < EV; ++e) {
              if (bpos[e] < 0) bpos[e] = yy * W + xx;      // (the first valid cell names a window nothing wins: all -inf / NaN)
              if ((float)v[e] > best[e]) { best[e] = (float)v[e]; bpos[e] = yy * W + xx; }
            }
          }
        }
        const V d = *reinterpret_cast<const V*>(dy + ((((size_t)n * Ho + oy) * Wo + ox) * C + cv * EV));
#pragma unroll
        for (int e = 0; e < EV; ++e)
          if (bpos[e] == iy * W + ix) g[e] += (float)d[e];
      }
    }
    V o;
#pragma unroll
    for (int e = 0; e < EV; ++e) o[e] = (T)g[e];
    if (relu) {
      const V xv = *reinterpret_cast<const V*>(x + idx * EV);
#pragma unroll
      for (int e = 0; e < EV; ++e) o[e] = xv[e] > (T)0 ? o[e] : (T)0;
    }
    *reinterpret_cast<V*>(dx + idx * EV) = o;
  }
}

// ------------------------------------------------------------------ loss forward + backward
struct LossArgs {
  const void* preds;       // [B, cells, K*(C+5)], float32 or (mixed-precision form) float16
  const float* anchors;    // [A,4] float32
  const float* mask;       // [B,A]
  const float* delta_in;   // [B,A,4]
  const float* box_in;     // [B,A,4] (cx,cy,w,h)
  const float* labels;     // [B,A,C]
  float* dpreds;           // [B, cells, K*(C+5)]
  f16* g16;                // mixed-precision form: (float16)(dpreds * gscale), the gradient the float16 backward starts from
  float gscale;            // (the loss scale)
  float* losses3;
  float* ious;             // [B,A]
  float* partial;          // [blocks][4]: class, conf, bbox loss partial sums (+ unused)
  int B, cells, K, C;
  int Bmean;               // divisor of the confidence loss's reduce_mean over the batch (nn_skeleton.py:304-312): B, or the
                           // GLOBAL batch when N replicas of batch B compute one graph of batch N*B (SURVEY.md 8e option b)
  float w1, h1, thr, slope, eps;
  float coef_class, coef_pos, coef_neg, coef_bbox;
  float num_obj;           // sum(mask) over the whole batch (host computed from the label tensors) ...
  const float* num_obj_dev;   // ... or, when not NULL, read from the device (sqdet_sum_f32 of the mask: no host round trip)
};

template <typename PT>
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
  const float nobj = a.num_obj_dev ? a.num_obj_dev[0] : a.num_obj;
  __shared__ float red[3][256];
  const int A = a.cells * a.K;
  const int ch = a.K * (a.C + 5);
  const long total = (long)a.B * A;
  float l_class = 0.f, l_conf = 0.f, l_bbox = 0.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / A);
    const int an = (int)(idx - (long)b * A);
    const int cell = an / a.K, k = an - cell * a.K;
    const PT* p = reinterpret_cast<const PT*>(a.preds) + ((size_t)b * a.cells + cell) * ch;
    float* dp = a.dpreds + ((size_t)b * a.cells + cell) * ch;
    f16* gp = a.g16 ? a.g16 + ((size_t)b * a.cells + cell) * ch : nullptr;
    // float32 dpreds, and -- mixed precision -- its loss-scaled float16 copy (convert_scale_kernel's expression) in the same pass
    auto put = [&](int off, float v) {
      dp[off] = v;
      if (sizeof(PT) == 2) gp[off] = (f16)(v * a.gscale);
    };
    const float m = a.mask[idx];
    // class probabilities (softmax) and their loss (nn_skeleton.py:150-160, 292-299)
    const PT* lg = p + k * a.C;
    float mx = (float)lg[0];
    for (int c = 1; c < a.C; ++c) mx = fmaxf(mx, (float)lg[c]);
    float sum = expf((float)lg[0] - mx);
    for (int c = 1; c < a.C; ++c) sum = sum + expf((float)lg[c] - mx);
    const float inv = 1.0f / sum;
    float gdotp = 0.f;
    for (int c = 0; c < a.C; ++c) {
      const float pc = expf((float)lg[c] - mx) * inv;
      const float lab = a.labels[idx * a.C + c];
      l_class += (lab * (-logf(pc + a.eps)) + (1.0f - lab) * (-logf(1.0f - pc + a.eps))) * m * a.coef_class;
      const float gc = m * a.coef_class / nobj * (-lab / (pc + a.eps) + (1.0f - lab) / (1.0f - pc + a.eps));
      gdotp += gc * pc;
    }
    for (int c = 0; c < a.C; ++c) {
      const float pc = expf((float)lg[c] - mx) * inv;
      const float lab = a.labels[idx * a.C + c];
      const float gc = m * a.coef_class / nobj * (-lab / (pc + a.eps) + (1.0f - lab) / (1.0f - pc + a.eps));
      put(k * a.C + c, pc * (gc - gdotp));
    }
    // decode (as interpret_output) -> IoU with the ground-truth box (nn_skeleton.py:240-269)
    const float zc = (float)p[a.K * a.C + k];
    const float conf = 1.0f / (1.0f + expf(-zc));
    const PT* dlp = p + a.K * (a.C + 1) + 4 * k;
    const float dl[4] = {(float)dlp[0], (float)dlp[1], (float)dlp[2], (float)dlp[3]};
    const f32x4 anc = *reinterpret_cast<const f32x4*>(a.anchors + (size_t)an * 4);
    const float cx = anc[0] + dl[0] * anc[2];
    const float cy = anc[1] + dl[1] * anc[3];
    const float ew = dl[2] > a.thr ? a.slope * ((dl[2] - a.thr) + 1.0f) : expf(dl[2]);
    const float eh = dl[3] > a.thr ? a.slope * ((dl[3] - a.thr) + 1.0f) : expf(dl[3]);
    const float bw = anc[2] * ew, bh = anc[3] * eh;
    float xmin = fminf(fmaxf(0.0f, cx - bw / 2.0f), a.w1), ymin = fminf(fmaxf(0.0f, cy - bh / 2.0f), a.h1);
    float xmax = fmaxf(fminf(a.w1, cx + bw / 2.0f), 0.0f), ymax = fmaxf(fminf(a.h1, cy + bh / 2.0f), 0.0f);
    const float w2 = xmax - xmin + 1.0f, h2 = ymax - ymin + 1.0f;
    const float dcx = xmin + 0.5f * w2, dcy = ymin + 0.5f * h2;
    // bbox_transform of det box and of the GT box (utils/util.py:167-179), _tensor_iou (:241-262)
    const float ax0 = dcx - w2 / 2.0f, ay0 = dcy - h2 / 2.0f, ax1 = dcx + w2 / 2.0f, ay1 = dcy + h2 / 2.0f;
    const f32x4 gb = *reinterpret_cast<const f32x4*>(a.box_in + idx * 4);
    const float bx0 = gb[0] - gb[2] / 2.0f, by0 = gb[1] - gb[3] / 2.0f, bx1 = gb[0] + gb[2] / 2.0f, by1 = gb[1] + gb[3] / 2.0f;
    const float iw = fmaxf(0.0f, fminf(ax1, bx1) - fmaxf(ax0, bx0));
    const float ih = fmaxf(0.0f, fminf(ay1, by1) - fmaxf(ay0, by0));
    const float inter = iw * ih;
    const float uni = (ax1 - ax0) * (ay1 - ay0) + (bx1 - bx0) * (by1 - by0) - inter;
    const float iou = inter / (uni + a.eps) * m;
    a.ious[idx] = iou;
    // confidence loss (:304-312): mean over the batch of sum_a (iou-conf)^2 * w_a
    const float wgt = m * a.coef_pos / nobj + (1.0f - m) * a.coef_neg / ((float)A - nobj);
    const float dc = iou - conf;
    l_conf += dc * dc * wgt / (float)a.Bmean;
    put(a.K * a.C + k, 2.0f * (conf - iou) * wgt / (float)a.Bmean * conf * (1.0f - conf));
    // bbox loss (:317-323)
    for (int d = 0; d < 4; ++d) {
      const float df = m * (dl[d] - a.delta_in[idx * 4 + d]);
      l_bbox += a.coef_bbox * df * df / nobj;
      put(a.K * (a.C + 1) + 4 * k + d, 2.0f * a.coef_bbox * m * df / nobj);
    }
  }
  l_class = l_class / nobj;
  red[0][threadIdx.x] = l_class; red[1][threadIdx.x] = l_conf; red[2][threadIdx.x] = l_bbox;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
      red[2][threadIdx.x] += red[2][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x < 4) a.partial[blockIdx.x * 4 + threadIdx.x] = threadIdx.x < 3 ? red[threadIdx.x][0] : 0.f;
}

// losses3[j] = sum over the loss kernel's workgroups of partial[b][j]: wave j of one workgroup, lane l takes b = l, l + 64, ..
// in ascending order, then a fixed xor tree -- deterministic.  (One thread per output walking up to 512 dependent loads took
// 40 us, and a device-to-device copy of the three floats followed it.)
__global__ __launch_bounds__(256) void loss_finish_kernel(const float* __restrict__ partial, float* __restrict__ losses3, int blocks) {
  const int j = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float s = 0.f;
  for (int b = lane; b < blocks; b += 64) s += partial[b * 4 + j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if (lane == 0 && j < 3) losses3[j] = s;
}

// ------------------------------------------------------------------ optimizer
// Segment table: variable v occupies [seg[v].off, +count) of the flat parameter / gradient / momentum
// buffers.  Pass 1 adds the weight-decay gradient and writes per-block partial sums of squares; pass 2
// (one block) finishes the norms in a fixed order; pass 3 clips per variable and applies Momentum.
struct OptSeg { long off; long count; float decay; int first_block; int nblocks; };

__global__ __launch_bounds__(256) void opt_sumsq_kernel(const OptSeg* __restrict__ segs, const int* __restrict__ block_seg,
                                                        const float* __restrict__ w, float* __restrict__ g,
                                                        double* __restrict__ partial, float grad_scale) {
  __shared__ double red[256];
  const int v = block_seg[blockIdx.x];
  const OptSeg s = segs[v];
  const int lb = blockIdx.x - s.first_block;
  double acc = 0.0;
  for (long i = (long)lb * 256 + threadIdx.x; i < s.count; i += (long)s.nblocks * 256) {
    float gi = g[s.off + i] * grad_scale;   // grad_scale = 1/world_size after a SUM all-reduce
    if (s.decay != 0.f) gi = gi + s.decay * w[s.off + i];
    g[s.off + i] = gi;
    acc += (double)gi * (double)gi;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// one block: per-variable clip factors + the overflow flag of mixed-precision training (a non-finite gradient norm in
// ANY variable: flag[0] = 1 and the apply kernel leaves every parameter and momentum untouched)
__global__ void opt_norm_kernel(const OptSeg* __restrict__ segs, const double* __restrict__ partial, float* __restrict__ scale,
                                int nvars, float max_norm, int* __restrict__ flag, int* __restrict__ found_inf) {
  int bad = 0;
  for (int v = threadIdx.x; v < nvars; v += blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < segs[v].nblocks; ++b) s += partial[segs[v].first_block + b];
    const float nrm = (float)sqrt(s);
    if (!(nrm <= 3.0e38f)) bad = 1;               // inf or NaN
    scale[v] = max_norm / fmaxf(nrm, max_norm);   // tf.clip_by_norm: g * clip / max(||g||, clip)
  }
  bad = __syncthreads_or(bad);
  if (threadIdx.x == 0) {
    flag[0] = bad;
    if (found_inf) found_inf[0] = bad;
  }
}

__global__ __launch_bounds__(256) void opt_apply_kernel(const OptSeg* __restrict__ segs, const int* __restrict__ block_seg,
                                                        float* __restrict__ w, const float* __restrict__ g,
                                                        float* __restrict__ accum, const float* __restrict__ scale,
                                                        float lr, float momentum, const int* __restrict__ flag) {
  if (flag[0]) return;   // overflowed step: skipped
  const int v = block_seg[blockIdx.x];
  const OptSeg s = segs[v];
  const int lb = blockIdx.x - s.first_block;
  const float sc = scale[v];
  for (long i = (long)lb * 256 + threadIdx.x; i < s.count; i += (long)s.nblocks * 256) {
    const float gc = g[s.off + i] * sc;
    const float ac = momentum * accum[s.off + i] + gc;   // MomentumOptimizer: accum = m*accum + g ; var -= lr*accum
    accum[s.off + i] = ac;
    w[s.off + i] = w[s.off + i] - lr * ac;
  }
}

// out[0] = sum of x[0..n) in a fixed order (one 1024-thread workgroup: per-thread strided partial sums over 16-byte
// vectors with four independent accumulators -- the first version walked 1300 dependent 4-byte loads per thread, 300 us of a
// 12 ms training step --, then a tree over LDS): num_objects = sum(input_mask), nn_skeleton.py:180, without a device -> host
// round trip
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) {
  __shared__ float red[1024];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const bool vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const size_t nv = vec ? n / 4 : 0;
  size_t i = threadIdx.x;
  for (; i + 7 * 1024 < nv; i += 8 * 1024) {      // eight loads in flight (one at a time, 337 k floats took 36 us); same add order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const f32x4*>(x)[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) { s0 += v[u][0]; s1 += v[u][1]; s2 += v[u][2]; s3 += v[u][3]; }
  }
  for (; i < nv; i += 1024) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    s0 += v[0]; s1 += v[1]; s2 += v[2]; s3 += v[3];
  }
  for (size_t i = nv * 4 + threadIdx.x; i < n; i += 1024) s0 += x[i];
  red[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// y = max(a + b, 0) (tf.nn.relu(shortcut + branch), resnet50_convDet.py:55) -- the residual add where the producing conv
// could not take it in its epilogue; 16-byte vectors
template <typename T>
__global__ void add_relu_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, size_t nv) {
  typedef typename Vec16<T>::type V;
  constexpr int EV = 16 / sizeof(T);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const V va = reinterpret_cast<const V*>(a)[i], vb = reinterpret_cast<const V*>(b)[i];
    V r;
#pragma unroll
    for (int e = 0; e < EV; ++e) { const float t = (float)va[e] + (float)vb[e]; r[e] = (T)(t > 0.f ? t : 0.f); }
    reinterpret_cast<V*>(y)[i] = r;
  }
}

// y[p, coff .. coff + c) = x[p, 0 .. c): one input of tf.concat(axis=3) (nets/squeezeDet.py:106); 16-byte vectors
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ x, T* __restrict__ y, size_t pixels, int cv, int cstride_v, int coff_v) {
  typedef typename Vec16<T>::type V;
  const size_t total = pixels * (size_t)cv;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i / cv;
    const int c = (int)(i - p * cv);
    reinterpret_cast<V*>(y)[p * cstride_v + coff_v + c] = reinterpret_cast<const V*>(x)[i];
  }
}

// mask[i] = floor(keep_prob + u_i), u_i uniform in [0,1): tf.nn.dropout's keep mask (nets/squeezeDet.py:74).  Counter-based
// generator (one 64-bit mix of (seed, i) per element -- splitmix64), so the mask depends on (seed, index) only.
template <typename T>
__global__ void dropout_mask_kernel(T* __restrict__ mask, size_t n, float keep_prob, unsigned long long seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0, 1)
    mask[i] = (T)floorf(keep_prob + u);
  }
}

static int grid_for(size_t n, int cap = 4096) {
  size_t b = (n + 255) / 256;
  if (b > (size_t)cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace sqdet

using namespace sqdet;

extern "C" int sqdet_conv_pack_weights_bwd_data(const float* w_hwio_f32, void* packed, int k, int cin, int cout,
                                                int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(w_hwio_f32 && packed && k > 0 && cin > 0 && cout > 0, "pack_weights_bwd_data: bad arguments");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "pack_weights_bwd_data: bad dtype");
  const ConvGeom g = conv_geom(k, cout, cin, dtype);   // the dgrad conv: cout -> cin channels
  SQDET_UNSUPPORTED(g.gather, "pack_weights_bwd_data: cout %d must be a multiple of %d", cout, g.kg);
  const size_t total = (size_t)g.ngroups * g.steps * g.nt * 64 * g.kg;
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(pack_bwd_data_kernel<f16>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w_hwio_f32,
                       (f16*)packed, k, cin, cout, g.nchunk, g.nt, total);
  else
    hipLaunchKernelGGL(pack_bwd_data_kernel<float>, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), w_hwio_f32,
                       (float*)packed, k, cin, cout, g.nchunk, g.nt, total);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" size_t sqdet_conv_pack_many_table_bytes(int nitems) { return (size_t)(nitems > 0 ? nitems : 0) * sizeof(sqdet::PackItem); }

extern "C" int sqdet_conv_pack_many_prepare_bn(const float* const* w_hwio_f32, void* const* packed, const int* k, const int* cin,
                                               const int* cout, const int* bwd_data, const float* const* gamma,
                                               const float* const* beta, const float* const* mean, const float* const* var,
                                               const float* const* conv_bias, float* const* b_folded, float eps, int nitems,
                                               int dtype, void* table_host, int* total_blocks) {
  SQDET_REQUIRE(w_hwio_f32 && packed && k && cin && cout && bwd_data && table_host && total_blocks && nitems > 0, "pack_many_prepare: bad arguments");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "pack_many_prepare: bad dtype");
  SQDET_REQUIRE(!gamma || (beta && mean && var && conv_bias && b_folded && eps >= 0.f), "pack_many_prepare: incomplete batch-norm arrays");
  sqdet::PackItem* t = reinterpret_cast<sqdet::PackItem*>(table_host);
  int blocks = 0;
  for (int i = 0; i < nitems; ++i) {
    SQDET_REQUIRE(w_hwio_f32[i] && packed[i] && k[i] > 0 && cin[i] > 0 && cout[i] > 0, "pack_many_prepare: bad item %d", i);
    const ConvGeom g = bwd_data[i] ? conv_geom(k[i], cout[i], cin[i], dtype) : conv_geom(k[i], cin[i], cout[i], dtype);
    SQDET_UNSUPPORTED(bwd_data[i] && g.gather, "pack_many_prepare: item %d: cout %d must be a multiple of %d for the backward-data order", i, cout[i], g.kg);
    sqdet::PackItem& p = t[i];
    p.w = w_hwio_f32[i]; p.out = packed[i]; p.k = k[i]; p.cin = cin[i]; p.cout = cout[i];
    p.taps = g.taps; p.kdim = g.kdim; p.nchunk = g.nchunk; p.nt = g.nt; p.bwd = bwd_data[i] ? 1 : 0;
    p.total = (unsigned long long)g.ngroups * g.steps * g.nt * 64 * g.kg;
    int nb = (int)((p.total + 255) / 256);
    if (nb > 256) nb = 256;
    p.first_block = blocks; p.nblocks = nb;
    blocks += nb;
    p.gamma = p.var = p.beta = p.mean = p.cbias = nullptr; p.b_folded = nullptr; p.eps = eps;
    if (gamma && gamma[i]) {
      SQDET_REQUIRE(beta[i] && mean[i] && var[i], "pack_many_prepare: item %d: gamma without beta / mean / var", i);
      p.gamma = gamma[i]; p.beta = beta[i]; p.mean = mean[i]; p.var = var[i]; p.cbias = conv_bias[i]; p.b_folded = b_folded[i];
    }
  }
  *total_blocks = blocks;
  return SQDET_OK;
}

extern "C" int sqdet_conv_pack_many_prepare(const float* const* w_hwio_f32, void* const* packed, const int* k, const int* cin,
                                            const int* cout, const int* bwd_data, int nitems, int dtype, void* table_host,
                                            int* total_blocks) {
  return sqdet_conv_pack_many_prepare_bn(w_hwio_f32, packed, k, cin, cout, bwd_data, nullptr, nullptr, nullptr, nullptr, nullptr,
                                         nullptr, 0.f, nitems, dtype, table_host, total_blocks);
}

extern "C" int sqdet_conv_pack_many(const void* table_dev, int nitems, int total_blocks, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(table_dev && nitems > 0 && total_blocks > 0, "pack_many: bad arguments");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "pack_many: bad dtype");
  const sqdet::PackItem* t = reinterpret_cast<const sqdet::PackItem*>(table_dev);
  if (dtype == SQDET_F16) hipLaunchKernelGGL(sqdet::pack_many_kernel<f16>, dim3(total_blocks), dim3(256), 0, as_stream(stream), t, nitems);
  else hipLaunchKernelGGL(sqdet::pack_many_kernel<float>, dim3(total_blocks), dim3(256), 0, as_stream(stream), t, nitems);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_conv2d_nhwc_bwd_data(const void* dy, const void* w_packed_bwd, void* dx, int n, int h, int w, int cin,
                                          int cout, int k, int dtype, int dy_cstride, int dy_coffset, int accumulate,
                                          sqdet_stream_t stream) {
  // stride-1 SAME convs only (every trainable conv of the reference's nets on this path)
  SQDET_REQUIRE(k == 1 || k == 3, "conv2d_bwd_data: k must be 1 or 3 (stride 1, SAME)");
  return conv2d_launch_ex(dy, w_packed_bwd, nullptr, dx, n, h, w, cout, cin, k, 1, SQDET_PAD_SAME, 0, dtype, cin, 0,
                          dy_cstride, dy_coffset, accumulate, as_stream(stream));
}

extern "C" int sqdet_conv2d_nhwc_bwd_data_relu(const void* dy, const void* w_packed_bwd, void* dx, const void* relu_of, int n, int h,
                                               int w, int cin, int cout, int k, int dtype, int dy_cstride, int dy_coffset,
                                               int accumulate, sqdet_stream_t stream) {
  SQDET_REQUIRE(k == 1 || k == 3, "conv2d_bwd_data: k must be 1 or 3 (stride 1, SAME)");
  return conv2d_launch_masked(dy, w_packed_bwd, nullptr, dx, n, h, w, cout, cin, k, 1, SQDET_PAD_SAME, 0, dtype, cin, 0,
                              dy_cstride, dy_coffset, accumulate, relu_of, as_stream(stream));
}

extern "C" int sqdet_relu_bwd(const void* y, void* dy_inout, size_t count, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "relu_bwd: bad dtype");
  const size_t ev = 16 / dtype_size(dtype);
  SQDET_REQUIRE(y && dy_inout && count % ev == 0, "relu_bwd: bad arguments (count must be a multiple of 16 bytes)");
  const dim3 grid(grid_for(count / ev, 8192));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(relu_bwd_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (const f16*)y, (f16*)dy_inout, count / ev);
  else
    hipLaunchKernelGGL(relu_bwd_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)y, (float*)dy_inout, count / ev);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_scale_mask(const void* x, const void* mask, void* y, float scale, size_t count, int dtype,
                                sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "scale_mask: bad dtype");
  const size_t ev = 16 / dtype_size(dtype);
  SQDET_REQUIRE(x && mask && y && count % ev == 0, "scale_mask: bad arguments (count must be a multiple of 16 bytes)");
  const dim3 grid(grid_for(count / ev, 8192));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(scale_mask_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (const f16*)x, (const f16*)mask, (f16*)y, scale, count / ev);
  else
    hipLaunchKernelGGL(scale_mask_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (const float*)mask, (float*)y, scale, count / ev);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_scale_mask_relu(const void* x, const void* mask, const void* relu_of, void* y, float scale, size_t count,
                                     int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "scale_mask_relu: bad dtype");
  const size_t ev = 16 / dtype_size(dtype);
  SQDET_REQUIRE(x && mask && relu_of && y && count % ev == 0, "scale_mask_relu: bad arguments (count must be a multiple of 16 bytes)");
  const dim3 grid(grid_for(count / ev, 8192));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(scale_mask_relu_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (const f16*)x, (const f16*)mask,
                       (const f16*)relu_of, (f16*)y, scale, count / ev);
  else
    hipLaunchKernelGGL(scale_mask_relu_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (const float*)mask,
                       (const float*)relu_of, (float*)y, scale, count / ev);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_convert_scale(const void* src, int src_dtype, void* dst, int dst_dtype, float scale, size_t count,
                                   sqdet_stream_t stream) {
  SQDET_REQUIRE(src && dst && count % 4 == 0, "convert_scale: bad arguments (count must be a multiple of 4)");
  SQDET_REQUIRE((src_dtype == SQDET_F16 || src_dtype == SQDET_F32) && (dst_dtype == SQDET_F16 || dst_dtype == SQDET_F32),
                "convert_scale: bad dtype");
  const dim3 grid(grid_for(count / 4, 8192));
  hipStream_t st = as_stream(stream);
  if (src_dtype == SQDET_F16 && dst_dtype == SQDET_F32)
    hipLaunchKernelGGL((convert_scale_kernel<f16, float>), grid, dim3(256), 0, st, (const f16*)src, (float*)dst, scale, count / 4);
  else if (src_dtype == SQDET_F32 && dst_dtype == SQDET_F16)
    hipLaunchKernelGGL((convert_scale_kernel<float, f16>), grid, dim3(256), 0, st, (const float*)src, (f16*)dst, scale, count / 4);
  else if (src_dtype == SQDET_F32)
    hipLaunchKernelGGL((convert_scale_kernel<float, float>), grid, dim3(256), 0, st, (const float*)src, (float*)dst, scale, count / 4);
  else
    hipLaunchKernelGGL((convert_scale_kernel<f16, f16>), grid, dim3(256), 0, st, (const f16*)src, (f16*)dst, scale, count / 4);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

static int maxpool_bwd_launch(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int k, int stride,
                              int pad_mode, int dtype, int relu, sqdet_stream_t stream);

extern "C" int sqdet_maxpool_nhwc_bwd(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int k,
                                      int stride, int pad_mode, int dtype, sqdet_stream_t stream) {
  return maxpool_bwd_launch(x, dy, dx, n, h, w, c, k, stride, pad_mode, dtype, 0, stream);
}

extern "C" int sqdet_maxpool_nhwc_bwd_relu(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int k,
                                           int stride, int pad_mode, int dtype, sqdet_stream_t stream) {
  return maxpool_bwd_launch(x, dy, dx, n, h, w, c, k, stride, pad_mode, dtype, 1, stream);
}

static int maxpool_bwd_launch(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int k, int stride,
                              int pad_mode, int dtype, int relu, sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "maxpool_bwd: bad dtype");
  const int ev = 16 / (int)dtype_size(dtype);
  SQDET_REQUIRE(x && dy && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % ev == 0 && k > 0 && stride > 0,
                "maxpool_bwd: bad arguments");
  const int Ho = out_size(h, k, stride, pad_mode), Wo = out_size(w, k, stride, pad_mode);
  const int pt = pad_before(h, k, stride, pad_mode), pl = pad_before(w, k, stride, pad_mode);
  if (k == 3 && stride == 2) {
    const int BA = (h + pt + 1) / 2, BB = (w + pl + 1) / 2;
    const dim3 grid2(grid_for((size_t)n * BA * BB * (c / ev), 16384));
    if (dtype == SQDET_F16)
      hipLaunchKernelGGL(maxpool3s2_bwd_kernel<f16>, grid2, dim3(256), 0, as_stream(stream), (const f16*)x, (const f16*)dy,
                         (f16*)dx, n, h, w, c, pt, pl, Ho, Wo, BA, BB, relu);
    else
      hipLaunchKernelGGL(maxpool3s2_bwd_kernel<float>, grid2, dim3(256), 0, as_stream(stream), (const float*)x,
                         (const float*)dy, (float*)dx, n, h, w, c, pt, pl, Ho, Wo, BA, BB, relu);
    SQDET_CHECK_HIP(hipGetLastError());
    return SQDET_OK;
  }
  const size_t total = (size_t)n * h * w * (c / ev);
  const dim3 grid(grid_for(total, 16384));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(maxpool_bwd_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (const f16*)x, (const f16*)dy, (f16*)dx,
                       n, h, w, c, k, stride, pt, pl, Ho, Wo, relu);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (const float*)dy,
                       (float*)dx, n, h, w, c, k, stride, pt, pl, Ho, Wo, relu);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_maxpool_nhwc_bwd_idx(const unsigned char* window_index, const void* y, const void* dy, void* dx, int n, int h,
                                          int w, int c, int k, int stride, int pad_mode, int dtype, int relu,
                                          sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "maxpool_bwd_idx: bad dtype");
  const int ev = 16 / (int)dtype_size(dtype);
  SQDET_REQUIRE(window_index && dy && dx && (y || !relu) && n > 0 && h > 0 && w > 0 && c > 0 && c % ev == 0, "maxpool_bwd_idx: bad arguments");
  SQDET_UNSUPPORTED(k != 3 || stride != 2, "maxpool_bwd_idx: 3x3 / stride-2 pools only (every pool of the reference's nets)");
  const int Ho = out_size(h, k, stride, pad_mode), Wo = out_size(w, k, stride, pad_mode);
  const int pt = pad_before(h, k, stride, pad_mode), pl = pad_before(w, k, stride, pad_mode);
  const int BA = (h + pt + 1) / 2, BB = (w + pl + 1) / 2;
  const dim3 grid(grid_for((size_t)n * BA * BB * (c / ev), 16384));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(maxpool3s2_bwd_idx_kernel<f16>, grid, dim3(256), 0, as_stream(stream), window_index, (const f16*)y,
                       (const f16*)dy, (f16*)dx, n, h, w, c, pt, pl, Ho, Wo, BA, BB, relu);
  else
    hipLaunchKernelGGL(maxpool3s2_bwd_idx_kernel<float>, grid, dim3(256), 0, as_stream(stream), window_index, (const float*)y,
                       (const float*)dy, (float*)dx, n, h, w, c, pt, pl, Ho, Wo, BA, BB, relu);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_sum_f32(const float* x, size_t count, float* out, sqdet_stream_t stream) {
  SQDET_REQUIRE(x && out && count > 0, "sum_f32: bad arguments");
  hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, count, out);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_add_relu(const void* a, const void* b, void* y, size_t count, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "add_relu: bad dtype");
  const size_t ev = 16 / dtype_size(dtype);
  SQDET_REQUIRE(a && b && y && count % ev == 0, "add_relu: bad arguments (count must be a multiple of 16 bytes)");
  const dim3 grid(grid_for(count / ev, 8192));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(add_relu_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (const f16*)a, (const f16*)b, (f16*)y, count / ev);
  else
    hipLaunchKernelGGL(add_relu_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)a, (const float*)b, (float*)y, count / ev);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_copy_channels(const void* x, void* y, size_t pixels, int c, int y_cstride, int y_coffset, int dtype,
                                   sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "copy_channels: bad dtype");
  const int ev = 16 / (int)dtype_size(dtype);
  SQDET_REQUIRE(x && y && pixels > 0 && c > 0 && c % ev == 0 && y_cstride % ev == 0 && y_coffset % ev == 0 && y_coffset >= 0 &&
                    y_coffset + c <= y_cstride, "copy_channels: channel counts / offsets must be multiples of 16 bytes");
  const dim3 grid(grid_for(pixels * (size_t)(c / ev), 8192));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(copy_channels_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (const f16*)x, (f16*)y, pixels, c / ev, y_cstride / ev, y_coffset / ev);
  else
    hipLaunchKernelGGL(copy_channels_kernel<float>, grid, dim3(256), 0, as_stream(stream), (const float*)x, (float*)y, pixels, c / ev, y_cstride / ev, y_coffset / ev);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_dropout_mask(void* mask, size_t count, float keep_prob, uint64_t seed, int dtype, sqdet_stream_t stream) {
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "dropout_mask: bad dtype");
  SQDET_REQUIRE(mask && count > 0 && keep_prob > 0.f && keep_prob <= 1.f, "dropout_mask: bad arguments");
  const dim3 grid(grid_for(count, 8192));
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(dropout_mask_kernel<f16>, grid, dim3(256), 0, as_stream(stream), (f16*)mask, count, keep_prob, (unsigned long long)seed);
  else
    hipLaunchKernelGGL(dropout_mask_kernel<float>, grid, dim3(256), 0, as_stream(stream), (float*)mask, count, keep_prob, (unsigned long long)seed);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

static int loss_fwd_bwd_impl(const void* preds, const float* anchors, const float* input_mask,
                             const float* box_delta_input, const float* box_input, const float* labels,
                             float* dpreds, float* ious, float* losses3, float* workspace, int batch, int gh, int gw,
                             int apg, int classes, float img_w, float img_h, float exp_thresh, float epsilon,
                             float coef_class, float coef_conf_pos, float coef_conf_neg, float coef_bbox,
                             float num_objects, const float* num_objects_dev, int global_batch, sqdet_stream_t stream,
                             void* g16 = nullptr, float gscale = 1.f);

extern "C" int sqdet_loss_fwd_bwd_mixed(const void* preds_f16, const float* anchors, const float* input_mask,
                                        const float* box_delta_input, const float* box_input, const float* labels,
                                        float* dpreds, void* dpreds_scaled_f16, float loss_scale, float* ious, float* losses3,
                                        float* workspace, int batch, int gh, int gw, int apg, int classes, float img_w,
                                        float img_h, float exp_thresh, float epsilon, float coef_class, float coef_conf_pos,
                                        float coef_conf_neg, float coef_bbox, float num_objects, const float* num_objects_dev,
                                        int global_batch, sqdet_stream_t stream) {
  SQDET_REQUIRE(dpreds_scaled_f16 && loss_scale > 0.f, "loss_fwd_bwd_mixed: bad arguments");
  return loss_fwd_bwd_impl(preds_f16, anchors, input_mask, box_delta_input, box_input, labels, dpreds, ious, losses3, workspace,
                           batch, gh, gw, apg, classes, img_w, img_h, exp_thresh, epsilon, coef_class, coef_conf_pos,
                           coef_conf_neg, coef_bbox, num_objects_dev ? 1.0f : num_objects, num_objects_dev, global_batch, stream,
                           dpreds_scaled_f16, loss_scale);
}

extern "C" int sqdet_loss_fwd_bwd_dev(const float* preds, const float* anchors, const float* input_mask,
                                      const float* box_delta_input, const float* box_input, const float* labels,
                                      float* dpreds, float* ious, float* losses3, float* workspace, int batch, int gh, int gw,
                                      int apg, int classes, float img_w, float img_h, float exp_thresh, float epsilon,
                                      float coef_class, float coef_conf_pos, float coef_conf_neg, float coef_bbox,
                                      const float* num_objects_dev, int global_batch, sqdet_stream_t stream) {
  SQDET_REQUIRE(num_objects_dev, "loss_fwd_bwd_dev: null num_objects");
  return loss_fwd_bwd_impl(preds, anchors, input_mask, box_delta_input, box_input, labels, dpreds, ious, losses3, workspace,
                           batch, gh, gw, apg, classes, img_w, img_h, exp_thresh, epsilon, coef_class, coef_conf_pos,
                           coef_conf_neg, coef_bbox, 1.0f, num_objects_dev, global_batch, stream);
}

extern "C" int sqdet_loss_fwd_bwd(const float* preds, const float* anchors, const float* input_mask,
                                  const float* box_delta_input, const float* box_input, const float* labels,
                                  float* dpreds, float* ious, float* losses3, float* workspace, int batch, int gh, int gw,
                                  int apg, int classes, float img_w, float img_h, float exp_thresh, float epsilon,
                                  float coef_class, float coef_conf_pos, float coef_conf_neg, float coef_bbox,
                                  float num_objects, int global_batch, sqdet_stream_t stream) {
  return loss_fwd_bwd_impl(preds, anchors, input_mask, box_delta_input, box_input, labels, dpreds, ious, losses3, workspace,
                           batch, gh, gw, apg, classes, img_w, img_h, exp_thresh, epsilon, coef_class, coef_conf_pos,
                           coef_conf_neg, coef_bbox, num_objects, nullptr, global_batch, stream);
}

static int loss_fwd_bwd_impl(const void* preds, const float* anchors, const float* input_mask,
                                  const float* box_delta_input, const float* box_input, const float* labels,
                                  float* dpreds, float* ious, float* losses3, float* workspace, int batch, int gh, int gw,
                                  int apg, int classes, float img_w, float img_h, float exp_thresh, float epsilon,
                                  float coef_class, float coef_conf_pos, float coef_conf_neg, float coef_bbox,
                                  float num_objects, const float* num_objects_dev, int global_batch, sqdet_stream_t stream,
                                  void* g16, float gscale) {
  SQDET_REQUIRE(preds && anchors && input_mask && box_delta_input && box_input && labels && dpreds && ious && losses3 &&
                    workspace, "loss_fwd_bwd: null pointer");
  SQDET_REQUIRE(batch > 0 && gh > 0 && gw > 0 && apg > 0 && classes > 0 && num_objects > 0.f, "loss_fwd_bwd: bad arguments");
  SQDET_REQUIRE(global_batch <= 0 || global_batch >= batch, "loss_fwd_bwd: global_batch smaller than batch");
  LossArgs a;
  a.preds = preds; a.anchors = anchors; a.mask = input_mask; a.delta_in = box_delta_input; a.box_in = box_input;
  a.labels = labels; a.dpreds = dpreds; a.ious = ious; a.partial = workspace;
  a.g16 = static_cast<f16*>(g16); a.gscale = gscale; a.losses3 = losses3;
  a.B = batch; a.Bmean = global_batch > 0 ? global_batch : batch; a.cells = gh * gw; a.K = apg; a.C = classes;
  a.w1 = img_w - 1.0f; a.h1 = img_h - 1.0f; a.thr = exp_thresh; a.slope = (float)exp((double)exp_thresh); a.eps = epsilon;
  a.coef_class = coef_class; a.coef_pos = coef_conf_pos; a.coef_neg = coef_conf_neg; a.coef_bbox = coef_bbox;
  a.num_obj = num_objects;
  a.num_obj_dev = num_objects_dev;
  const long total = (long)batch * gh * gw * apg;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 512) blocks = 512;
  hipStream_t st = as_stream(stream);
  if (g16) hipLaunchKernelGGL(loss_kernel<f16>, dim3(blocks), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(loss_kernel<float>, dim3(blocks), dim3(256), 0, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(256), 0, st, workspace, losses3, blocks);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" size_t sqdet_loss_workspace_bytes(void) { return (4 * 512 + 4) * sizeof(float); }

// ---- optimizer: host-side segment table lives in a small opaque object ----
struct sqdet_optimizer {
  std::vector<OptSeg> segs;
  std::vector<int> block_seg;
  int nblocks = 0;
};

extern "C" int sqdet_optimizer_create(sqdet_optimizer** out, const long* offsets, const long* counts, const float* decays,
                                      int nvars) {
  SQDET_REQUIRE(out && offsets && counts && decays && nvars > 0, "optimizer_create: bad arguments");
  sqdet_optimizer* o = new sqdet_optimizer();
  for (int v = 0; v < nvars; ++v) {
    OptSeg s;
    s.off = offsets[v]; s.count = counts[v]; s.decay = decays[v];
    long nb = (counts[v] + 256 * 8 - 1) / (256 * 8);
    if (nb < 1) nb = 1;
    if (nb > 64) nb = 64;
    s.first_block = o->nblocks; s.nblocks = (int)nb;
    o->nblocks += (int)nb;
    for (long b = 0; b < nb; ++b) o->block_seg.push_back(v);
    o->segs.push_back(s);
  }
  *out = o;
  return SQDET_OK;
}

extern "C" void sqdet_optimizer_destroy(sqdet_optimizer* o) { delete o; }

extern "C" size_t sqdet_optimizer_workspace_bytes(const sqdet_optimizer* o) {
  if (!o) return 0;
  // segs | block_seg | partial (double) | scale
  size_t b = o->segs.size() * sizeof(OptSeg);
  b = (b + 255) / 256 * 256 + o->block_seg.size() * sizeof(int);
  b = (b + 255) / 256 * 256 + (size_t)o->nblocks * sizeof(double);
  b = (b + 255) / 256 * 256 + o->segs.size() * sizeof(float);
  return b + 256;
}

extern "C" int sqdet_optimizer_step(sqdet_optimizer* o, float* params, float* grads, float* accum, void* workspace,
                                    float lr, float momentum, float max_grad_norm, float grad_scale, int32_t* found_inf,
                                    sqdet_stream_t stream) {
  SQDET_REQUIRE(o && params && grads && accum && workspace, "optimizer_step: null pointer");
  hipStream_t st = as_stream(stream);
  char* ws = reinterpret_cast<char*>(workspace);
  size_t off = 0;
  OptSeg* d_segs = reinterpret_cast<OptSeg*>(ws + off);
  off = (off + o->segs.size() * sizeof(OptSeg) + 255) / 256 * 256;
  int* d_bs = reinterpret_cast<int*>(ws + off);
  off = (off + o->block_seg.size() * sizeof(int) + 255) / 256 * 256;
  double* d_part = reinterpret_cast<double*>(ws + off);
  off = (off + (size_t)o->nblocks * sizeof(double) + 255) / 256 * 256;
  float* d_scale = reinterpret_cast<float*>(ws + off);
  SQDET_CHECK_HIP(hipMemcpyAsync(d_segs, o->segs.data(), o->segs.size() * sizeof(OptSeg), hipMemcpyHostToDevice, st));
  SQDET_CHECK_HIP(hipMemcpyAsync(d_bs, o->block_seg.data(), o->block_seg.size() * sizeof(int), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(opt_sumsq_kernel, dim3(o->nblocks), dim3(256), 0, st, d_segs, d_bs, params, grads, d_part, grad_scale);
  SQDET_CHECK_HIP(hipGetLastError());
  int* d_flag = reinterpret_cast<int*>(d_scale + o->segs.size());
  hipLaunchKernelGGL(opt_norm_kernel, dim3(1), dim3(256), 0, st, d_segs, d_part, d_scale, (int)o->segs.size(), max_grad_norm,
                     d_flag, found_inf);
  SQDET_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(opt_apply_kernel, dim3(o->nblocks), dim3(256), 0, st, d_segs, d_bs, params, grads, accum, d_scale, lr,
                     momentum, d_flag);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

// Device -> PINNED HOST copy as a kernel (the destination is host memory mapped into the device's address space:
// hipHostMalloc / torch pin_memory).  hipMemcpyAsync device -> host was observed to hold the calling host thread until
// the stream reached the copy, which serialised the serving loop's two-stage pipeline; a kernel launch never does.
namespace sqdet {
__global__ void copy_to_mapped_host_kernel(const int4* __restrict__ src, int4* __restrict__ dst, size_t nv) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace sqdet

extern "C" int sqdet_copy_to_mapped_host(const void* src_device, void* dst_pinned_host, size_t nbytes, sqdet_stream_t stream) {
  SQDET_REQUIRE(src_device && dst_pinned_host && nbytes > 0 && nbytes % 16 == 0, "copy_to_mapped_host: bad arguments (16-byte multiples)");
  const size_t nv = nbytes / 16;
  size_t blocks = (nv + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(sqdet::copy_to_mapped_host_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream),
                     (const int4*)src_device, (int4*)dst_pinned_host, nv);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}
