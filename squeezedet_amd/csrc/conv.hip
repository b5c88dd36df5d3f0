// NHWC convolution (+bias +ReLU) for gfx950 on MFMA, replacing
// ModelSkeleton._conv_layer (reference src/nn_skeleton.py:471-563).
//
// Formulation: implicit GEMM  D[cout][pixel] = sum_{tap,cin} W[tap][cin][cout] * X[pixel@tap][cin]
//   * the WEIGHTS are the MFMA A operand (rows = output channels), the ACTIVATIONS the B
//     operand (cols = 16 output pixels).  With the 16x16 C/D layout (col = lane&15,
//     row = 4*(lane>>4)+reg) each lane then owns ONE pixel and 4 CONSECUTIVE output channels
//     per tile; the cout->row permutation chosen at pack time makes that 4*NT consecutive
//     channels per lane, i.e. contiguous NHWC stores, no transposing epilogue.
//   * K is walked in 64-byte "chunk-steps": lane group g = lane>>4 owns bytes [16g,16g+16) of
//     every 64-byte run of channels, so each B fragment is ONE 16-byte load per lane straight
//     from the NHWC tensor (8 f16 -> one mfma_f32_16x16x32_f16; 4 f32 -> four
//     mfma_f32_16x16x4f32).  Only A/B agreement on the slot->channel map matters.
//   * fp32 accumulate always; f32 storage uses the exact-f32 MFMA (fmaf-chain numerics).
//
// Kernels here:
//   conv_direct<T,MT,NT>  any k/stride/pad, Cin % (16/sizeof(T)) == 0; B fragments read from
//                         global/L1/L2 with per-tap bounds predication (zero padding).
//   conv_gather<T,MT,NT>  tiny Cin (the 3-channel stems): K' = k*k*Cin im2col-gathered.
#include <stdlib.h>
#include <string.h>

#include "conv_common.h"

namespace sqdet {

template <typename T, int MT, int NT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[MT][NT], const int (&pix)[MT],
                                              const bool (&valid)[MT], int group, int g) {
  T* y = reinterpret_cast<T*>(a.y);
  const int cbase = group * (16 * NT) + g * (4 * NT);
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int c = cbase + n * 4;
    if (c < a.Cout) {  // Cout % 4 == 0 (checked on the host)
      const f32x4 b = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        if (valid[m]) {
          f32x4 v = acc[m][n] + b;
          T* dst = y + (size_t)pix[m] * a.y_cstride + a.y_coffset + c;
          if (a.accum) {
            v[0] += (float)dst[0]; v[1] += (float)dst[1]; v[2] += (float)dst[2]; v[3] += (float)dst[3];
          }
          if (a.relu) {
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          }
          if (a.relu_of) {
            const T* r = reinterpret_cast<const T*>(a.relu_of) + (size_t)pix[m] * a.y_cstride + a.y_coffset + c;
            v[0] = (float)r[0] > 0.f ? v[0] : 0.f; v[1] = (float)r[1] > 0.f ? v[1] : 0.f;
            v[2] = (float)r[2] > 0.f ? v[2] : 0.f; v[3] = (float)r[3] > 0.f ? v[3] : 0.f;
          }
          store4<T>(dst, v);
        }
      }
    }
  }
}

template <typename T, int MT, int NT>
__global__ __launch_bounds__(256) void conv_direct(ConvArgs a) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= a.ntiles * a.ngroups) return;
  const int tile = wid / a.ngroups, group = wid - tile * a.ngroups;
  const int j = lane & 15, g = lane >> 4;

  int pix[MT], iy0[MT], ix0[MT], nb[MT];
  bool valid[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int p = (tile * MT + m) * 16 + j;
    valid[m] = p < a.P;
    if (!valid[m]) p = a.P - 1;
    pix[m] = p;
    const int n = p / (a.Ho * a.Wo);
    const int r = p - n * (a.Ho * a.Wo);
    const int oy = r / a.Wo, ox = r - oy * a.Wo;
    nb[m] = n * a.H;
    iy0[m] = oy * a.stride - a.pt;
    ix0[m] = ox * a.stride - a.pl;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* x = reinterpret_cast<const T*>(a.x);
  const i32x4* wp = reinterpret_cast<const i32x4*>(a.wp) + (size_t)group * a.steps * NT * 64 + lane;
  const i32x4 zero = {0, 0, 0, 0};

  for (int ty = 0; ty < a.k; ++ty) {
    for (int tx = 0; tx < a.k; ++tx) {
      const T* src[MT];
      bool inb[MT];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int iy = iy0[m] + ty, ix = ix0[m] + tx;
        inb[m] = valid[m] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        src[m] = x + ((size_t)(nb[m] + iy) * a.W + ix) * a.x_cstride + a.x_coffset + g * KG;
      }
      auto load = [&](int c, i32x4 (&bf)[MT], i32x4 (&af)[NT]) {
        const bool cin_ok = c * KC + g * KG < a.Cin;
#pragma unroll
        for (int m = 0; m < MT; ++m)
          bf[m] = (inb[m] && cin_ok) ? *reinterpret_cast<const i32x4*>(src[m] + c * KC) : zero;
#pragma unroll
        for (int n = 0; n < NT; ++n) af[n] = wp[(size_t)c * NT * 64 + n * 64];
      };
      if constexpr (sizeof(T) == 2) {
        // K loop of this tap, register double-buffered: chunk c + 1's fragments are requested before chunk c's MFMAs
        // (two register sets, the loop unrolled by two -- a plain load -> MFMA loop paid a memory round trip per
        // 64-byte chunk, 32 of them in a 1024-channel 1x1 conv, hidden only by occupancy).  ResNet50 fp16 inference
        // +6 %, SqueezeDet+ +8 %; in float32 (16-channel chunks, four MFMAs per fragment pair) the extra registers cost
        // more occupancy than the prefetch gains (SqueezeDet fp32 training -3 %): float32 keeps the plain loop.
        i32x4 bfa[MT], afa[NT], bfb[MT], afb[NT];
        load(0, bfa, afa);
#pragma unroll 1
        for (int c = 0; c < a.nchunk; c += 2) {
          if (c + 1 < a.nchunk) load(c + 1, bfb, afb);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) mma16<T>(acc[m][n], afa[n], bfa[m]);
          if (c + 1 < a.nchunk) {
            if (c + 2 < a.nchunk) load(c + 2, bfa, afa);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int n = 0; n < NT; ++n) mma16<T>(acc[m][n], afb[n], bfb[m]);
          }
        }
      } else {
        for (int c = 0; c < a.nchunk; ++c) {
          i32x4 bf[MT], af[NT];
          load(c, bf, af);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) mma16<T>(acc[m][n], af[n], bf[m]);
        }
      }
      wp += (size_t)a.nchunk * NT * 64;
    }
  }
  conv_epilogue<T, MT, NT>(a, acc, pix, valid, group, g);
}

// Tiny-Cin stems (conv1: 3x3x3 s2 / 7x7x3 s2): the K axis is the flattened (ty,tx,cin) index,
// gathered element by element (HWIO memory order == that flattening, so the packer sees a
// 1x1 kernel with k*k*Cin input channels).
template <typename T, int MT, int NT>
__global__ __launch_bounds__(256) void conv_gather(ConvArgs a) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= a.ntiles * a.ngroups) return;
  const int tile = wid / a.ngroups, group = wid - tile * a.ngroups;
  const int j = lane & 15, g = lane >> 4;
  const int kdim = a.k * a.k * a.Cin;

  int pix[MT], iy0[MT], ix0[MT], nb[MT];
  bool valid[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    int p = (tile * MT + m) * 16 + j;
    valid[m] = p < a.P;
    if (!valid[m]) p = a.P - 1;
    pix[m] = p;
    const int n = p / (a.Ho * a.Wo);
    const int r = p - n * (a.Ho * a.Wo);
    const int oy = r / a.Wo, ox = r - oy * a.Wo;
    nb[m] = n * a.H;
    iy0[m] = oy * a.stride - a.pt;
    ix0[m] = ox * a.stride - a.pl;
  }
  f32x4 acc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* x = reinterpret_cast<const T*>(a.x);
  const i32x4* wp = reinterpret_cast<const i32x4*>(a.wp) + (size_t)group * a.steps * NT * 64 + lane;

  for (int c = 0; c < a.nchunk; ++c) {
    typedef T TV __attribute__((ext_vector_type(KG)));
    TV bv[MT];
#pragma unroll
    for (int e = 0; e < KG; ++e) {
      const int q = c * KC + g * KG + e;
      const bool qok = q < kdim;
      const int tap = q / a.Cin, ch = q - tap * a.Cin;
      const int ty = tap / a.k, tx = tap - ty * a.k;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int iy = iy0[m] + ty, ix = ix0[m] + tx;
        const bool ok = qok && valid[m] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        bv[m][e] = ok ? x[((size_t)(nb[m] + iy) * a.W + ix) * a.Cin + ch] : (T)0;
      }
    }
    i32x4 af[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) af[n] = wp[n * 64];
    wp += NT * 64;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n) mma16<T>(acc[m][n], af[n], __builtin_bit_cast(i32x4, bv[m]));
  }
  conv_epilogue<T, MT, NT>(a, acc, pix, valid, group, g);
}

// ---- weight packing: float32 HWIO -> [group][step][nt][lane][KG] fragments ----
// element (group, step=(tap,chunk), n, lane=(i=lane&15, g=lane>>4), e):
//   cout = group*16*NT + (i>>2)*4*NT + n*4 + (i&3)      (row i of tile n <-> consecutive couts per lane)
//   cin  = chunk*KC + g*KG + e
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int taps, int kdim, int cout,
                                    int nchunk, int nt, size_t total) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t t = idx;
    const int e = t % KG; t /= KG;
    const int lane = t % 64; t /= 64;
    const int n = t % nt; t /= nt;
    const int step = t % (taps * nchunk); t /= (taps * nchunk);
    const int group = (int)t;
    const int tap = step / nchunk, chunk = step - tap * nchunk;
    const int i = lane & 15, g = lane >> 4;
    const int co = group * 16 * nt + (i >> 2) * 4 * nt + n * 4 + (i & 3);
    const int ci = chunk * KC + g * KG + e;
    float v = 0.f;
    if (co < cout && ci < kdim) v = w[((size_t)tap * kdim + ci) * cout + co];
    out[idx] = (T)v;
  }
}

template <typename T, int MT, int NT>
static void launch_conv(const ConvArgs& a, bool gather, hipStream_t st) {
  const int waves = a.ntiles * a.ngroups;
  const int blocks = (waves + 3) / 4;
  if (gather)
    hipLaunchKernelGGL((conv_gather<T, MT, NT>), dim3(blocks), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL((conv_direct<T, MT, NT>), dim3(blocks), dim3(256), 0, st, a);
}

template <typename T, int MT>
static int dispatch_nt(const ConvArgs& a, int nt, bool gather, hipStream_t st) {
  switch (nt) {
    case 1: launch_conv<T, MT, 1>(a, gather, st); break;
    case 2: launch_conv<T, MT, 2>(a, gather, st); break;
    case 3: launch_conv<T, MT, 3>(a, gather, st); break;
    case 4: launch_conv<T, MT, 4>(a, gather, st); break;
    case 5: launch_conv<T, MT, 5>(a, gather, st); break;
    case 6: launch_conv<T, MT, 6>(a, gather, st); break;
    default: set_error("conv: bad NT %d", nt); return SQDET_EINVAL;
  }
  return SQDET_OK;
}

template <typename T>
static int dispatch_mt(ConvArgs& a, int nt, bool gather, hipStream_t st) {
  // pixel blocks per wave: keep >= ~4 waves per CU in flight when the problem is small
  int mt = 4;
  auto waves = [&](int m) { return (long)((a.P + 16 * m - 1) / (16 * m)) * a.ngroups; };
  if (waves(4) < 2048) mt = 2;
  if (mt == 2 && waves(2) < 2048) mt = 1;
  if (nt >= 5 && mt == 4) mt = 2;  // accumulator budget
  a.ntiles = (a.P + 16 * mt - 1) / (16 * mt);
  switch (mt) {
    case 1: return dispatch_nt<T, 1>(a, nt, gather, st);
    case 2: return dispatch_nt<T, 2>(a, nt, gather, st);
    default: return dispatch_nt<T, 4>(a, nt, gather, st);
  }
}

int conv2d_launch_ex(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                     int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                     int x_cstride, int x_coffset, int accum, hipStream_t st);
int conv2d_launch_masked(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                         int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                         int x_cstride, int x_coffset, int accum, const void* relu_of, hipStream_t st);
int conv2d_launch_res(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                      int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                      int x_cstride, int x_coffset, int accum, const void* relu_of, const void* res, hipStream_t st);

int conv2d_launch(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                  int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                  hipStream_t st) {
  return conv2d_launch_ex(x, w_packed, bias, y, n, h, w, cin, cout, k, stride, pad_mode, relu, dtype, y_cstride,
                          y_coffset, cin, 0, 0, st);
}

// x_cstride / x_coffset: the input is channels [x_coffset, x_coffset+cin) of rows x_cstride wide;
// accum: y += conv(x) instead of y = conv(x).  Channel slices are served by the generic kernel only; accumulating 1x1 /
// stride-1 convs over a whole tensor also by conv1x1_tile.
int conv2d_launch_ex(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                     int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                     int x_cstride, int x_coffset, int accum, hipStream_t st) {
  return conv2d_launch_masked(x, w_packed, bias, y, n, h, w, cin, cout, k, stride, pad_mode, relu, dtype, y_cstride, y_coffset,
                              x_cstride, x_coffset, accum, nullptr, st);
}

// relu_of != NULL: y (after the optional accumulate) is zeroed where relu_of <= 0 (same layout as y): see ConvArgs
int conv2d_launch_masked(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                         int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                         int x_cstride, int x_coffset, int accum, const void* relu_of, hipStream_t st) {
  return conv2d_launch_res(x, w_packed, bias, y, n, h, w, cin, cout, k, stride, pad_mode, relu, dtype, y_cstride, y_coffset, x_cstride,
                           x_coffset, accum, relu_of, nullptr, st);
}

// res != NULL (with accum): y = conv(x) + res instead of y += conv(x); res has y's layout (whole rows of y_cstride channels)
int conv2d_launch_res(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cin,
                      int cout, int k, int stride, int pad_mode, int relu, int dtype, int y_cstride, int y_coffset,
                      int x_cstride, int x_coffset, int accum, const void* relu_of, const void* res, hipStream_t st) {
  SQDET_REQUIRE(x && w_packed && y, "conv2d: null pointer");  // bias == NULL means no bias (generic kernel)
  SQDET_REQUIRE(!res || accum, "conv2d: a residual tensor needs the accumulate form");
  const int kg_ = dtype == SQDET_F16 ? 8 : 4;
  SQDET_UNSUPPORTED((x_cstride != cin || x_coffset != 0) &&
                        (x_cstride % kg_ != 0 || x_coffset % kg_ != 0 || x_coffset < 0 || x_coffset + cin > x_cstride),
                    "conv2d: x_cstride %d / x_coffset %d must be multiples of %d with coffset+cin <= cstride",
                    x_cstride, x_coffset, kg_);
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "conv2d: bad dtype %d", dtype);
  SQDET_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && cout > 0 && k > 0 && stride > 0, "conv2d: bad dims");
  SQDET_REQUIRE(pad_mode == SQDET_PAD_SAME || pad_mode == SQDET_PAD_VALID, "conv2d: bad pad_mode %d", pad_mode);
  SQDET_REQUIRE(pad_mode == SQDET_PAD_SAME || (h >= k && w >= k), "conv2d: VALID needs h,w >= k");
  SQDET_UNSUPPORTED(cout % 4 != 0, "conv2d: cout %d not a multiple of 4", cout);
  SQDET_UNSUPPORTED(y_cstride % 4 != 0 || y_coffset % 4 != 0 || y_coffset < 0 || y_coffset + cout > y_cstride,
                    "conv2d: y_cstride %d / y_coffset %d must be multiples of 4 with coffset+cout <= cstride",
                    y_cstride, y_coffset);
  const ConvGeom g = conv_geom(k, cin, cout, dtype);
  ConvArgs a;
  a.x = x; a.wp = w_packed; a.bias = bias; a.y = y;
  a.N = n; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout; a.k = k; a.stride = stride;
  a.pt = pad_before(h, k, stride, pad_mode);
  a.pl = pad_before(w, k, stride, pad_mode);
  a.Ho = out_size(h, k, stride, pad_mode);
  a.Wo = out_size(w, k, stride, pad_mode);
  const long P = (long)n * a.Ho * a.Wo;
  SQDET_UNSUPPORTED(P > (1L << 30) || (long)n * h * w > (1L << 30), "conv2d: more than 2^30 pixels");
  a.P = (int)P;
  a.ntiles = 0;
  a.nchunk = g.nchunk; a.steps = g.steps; a.ngroups = g.ngroups;
  a.y_cstride = y_cstride; a.y_coffset = y_coffset; a.relu = relu;
  a.x_cstride = x_cstride; a.x_coffset = x_coffset; a.accum = accum;
  a.relu_of = relu_of;
  a.res = res == y ? nullptr : res;
  a.scores = nullptr; a.score_apg = 0; a.score_classes = 0;
  const bool plain = x_cstride == cin && x_coffset == 0 && !accum && bias != nullptr && !relu_of;
  SQDET_UNSUPPORTED(!plain && g.gather, "conv2d: channel-sliced / accumulating convs need Cin %% %d == 0", kg_);
  bool handled = false;
  int rc = SQDET_OK;
  if (a.res) {                // only conv1x1_pipe reads the residual from a tensor of its own: everybody else adds into a copy of it
    if (!g.gather) {
      rc = conv1x1_tile_launch(a, g, dtype, st, &handled);
      if (rc != SQDET_OK || handled) return rc;
    }
    SQDET_CHECK_HIP(hipMemcpyAsync(y, a.res, (size_t)P * y_cstride * dtype_size(dtype), hipMemcpyDeviceToDevice, st));
    a.res = nullptr;
  }
  if (plain || !g.gather) {   // (the tile kernels also take channel-sliced inputs, no bias and y += : the backward-data convs)
    rc = conv3x3_tile_launch(a, g, dtype, st, &handled);
    if (rc != SQDET_OK || handled) return rc;
  }
  if (plain) {
    rc = conv1x1_stream_launch(a, g, dtype, st, &handled);
    if (rc != SQDET_OK || handled) return rc;
  }
  if (plain) {       // deep-K 1x1 on large maps: weights resident in LDS, activations streamed once (conv1x1k.hip)
    rc = conv1x1_deepk_launch(a, g, dtype, st, &handled);
    if (rc != SQDET_OK || handled) return rc;
  }
  if (!g.gather) {   // deep-K 1x1 (incl. the accumulating / sliced forms): workgroup GEMM tile, gemm1x1.hip
    rc = conv1x1_tile_launch(a, g, dtype, st, &handled);
    if (rc != SQDET_OK || handled) return rc;
  }
  rc = dtype == SQDET_F16 ? dispatch_mt<f16>(a, g.nt, g.gather, st) : dispatch_mt<float>(a, g.nt, g.gather, st);
  if (rc != SQDET_OK) return rc;
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

// ConvDet + the score half of interpret_output in one launch (convdet.hip SCORE form): preds = conv3x3/SAME(x) + bias (no
// ReLU), scores[n, h*w*apg] = det_probs.  UNSUPPORTED unless the split-K ConvDet kernel takes the shape in float16.
int convdet_scored_launch(const void* x, const void* w_packed, const float* bias, void* preds, float* scores, int n, int h, int w,
                          int cin, int apg, int classes, int dtype, hipStream_t st) {
  SQDET_REQUIRE(x && w_packed && bias && preds && scores, "convdet: null pointer");
  SQDET_REQUIRE(n > 0 && h > 0 && w > 0 && cin > 0 && apg > 0 && classes > 0, "convdet: bad dims");
  const int cout = apg * (classes + 5);
  SQDET_UNSUPPORTED(conv_algo() != 0 || !convdet_score_supported(cout, apg, classes, dtype),
                    "convdet: the score epilogue needs float16, 9 anchors x (3 classes + 5) and conv_algo = auto");
  const ConvGeom g = conv_geom(3, cin, cout, dtype);
  SQDET_UNSUPPORTED(g.gather || !(g.nt == 5 && g.ngroups == 1 && g.nchunk >= 8 && g.nchunk % 4 == 0),
                    "convdet: Cin %d is not a split-K ConvDet shape", cin);
  ConvArgs a;
  a.x = x; a.wp = w_packed; a.bias = bias; a.y = preds;
  a.N = n; a.H = h; a.W = w; a.Cin = cin; a.Cout = cout; a.k = 3; a.stride = 1;
  a.pt = 1; a.pl = 1; a.Ho = h; a.Wo = w;
  const long P = (long)n * h * w;
  SQDET_UNSUPPORTED(P > (1L << 30), "convdet: more than 2^30 pixels");
  a.P = (int)P; a.ntiles = 0;
  a.nchunk = g.nchunk; a.steps = g.steps; a.ngroups = g.ngroups;
  a.y_cstride = cout; a.y_coffset = 0; a.relu = 0;
  a.x_cstride = cin; a.x_coffset = 0; a.accum = 0; a.relu_of = nullptr; a.res = nullptr;
  a.scores = scores; a.score_apg = apg; a.score_classes = classes;
  bool handled = false;
  const int rc = conv3x3_tile_launch(a, g, dtype, st, &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "convdet: the split-K kernel did not take this shape");
  return SQDET_OK;
}

static int g_conv_algo = -1;
int conv_algo() {
  if (g_conv_algo < 0) {
    const char* e = getenv("SQDET_CONV_ALGO");
    g_conv_algo = (e && !strcmp(e, "generic")) ? 1 : 0;
  }
  return g_conv_algo;
}
void set_conv_algo(int v) { g_conv_algo = v; }

// Experiment knobs (0 = built-in heuristic): see tune() call sites.
static const char* const kTuneNames[] = {"c1_waves", "c1_mt", "c1_min_tiles", "fire_fuse", "stem_algo", "dbg", "g1_wr", "g1_mbw", "g1_ntw", "g1_ns"};
constexpr int kNumTune = 10;
static int g_tune[kNumTune] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
int tune(int which) { return g_tune[which]; }

}  // namespace sqdet

using namespace sqdet;

extern "C" int sqdet_set_option(const char* name, int value) {
  SQDET_REQUIRE(name, "set_option: null name");
  if (!strcmp(name, "conv_algo")) {
    SQDET_REQUIRE(value == 0 || value == 1, "set_option: conv_algo must be 0 (auto) or 1 (generic kernels only)");
    set_conv_algo(value);
    return SQDET_OK;
  }
  for (int i = 0; i < kNumTune; ++i) {
    if (!strcmp(name, kTuneNames[i])) {
      g_tune[i] = value;
      return SQDET_OK;
    }
  }
  set_error("set_option: unknown option '%s'", name);
  return SQDET_EINVAL;
}

extern "C" size_t sqdet_conv_packed_bytes(int k, int cin, int cout, int dtype) {
  if (k <= 0 || cin <= 0 || cout <= 0 || (dtype != SQDET_F16 && dtype != SQDET_F32)) return 0;
  const ConvGeom g = conv_geom(k, cin, cout, dtype);
  return (size_t)g.ngroups * g.steps * g.nt * 1024;
}

extern "C" int sqdet_conv_pack_weights(const float* w_hwio_f32, void* packed, int k, int cin, int cout, int dtype,
                                       sqdet_stream_t stream) {
  SQDET_REQUIRE(w_hwio_f32 && packed, "pack_weights: null pointer");
  SQDET_REQUIRE(k > 0 && cin > 0 && cout > 0, "pack_weights: bad dims");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "pack_weights: bad dtype %d", dtype);
  const ConvGeom g = conv_geom(k, cin, cout, dtype);
  const size_t total = (size_t)g.ngroups * g.steps * g.nt * 64 * g.kg;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dtype == SQDET_F16)
    hipLaunchKernelGGL(pack_weights_kernel<f16>, dim3(blocks), dim3(256), 0, as_stream(stream), w_hwio_f32,
                       (f16*)packed, g.taps, g.kdim, cout, g.nchunk, g.nt, total);
  else
    hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, as_stream(stream), w_hwio_f32,
                       (float*)packed, g.taps, g.kdim, cout, g.nchunk, g.nt, total);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_conv2d_nhwc_fwd(const void* x, const void* w_packed, const float* bias, void* y, int n, int h,
                                     int w, int cin, int cout, int k, int stride, int pad_mode, int relu, int dtype,
                                     int y_cstride, int y_coffset, sqdet_stream_t stream) {
  return conv2d_launch(x, w_packed, bias, y, n, h, w, cin, cout, k, stride, pad_mode, relu, dtype, y_cstride,
                       y_coffset, as_stream(stream));
}

extern "C" int sqdet_convdet_fwd(const void* x, const void* w_packed, const float* bias, void* preds, float* scores, int n, int h,
                                 int w, int cin, int anchors_per_grid, int classes, int dtype, sqdet_stream_t stream) {
  return convdet_scored_launch(x, w_packed, bias, preds, scores, n, h, w, cin, anchors_per_grid, classes, dtype, as_stream(stream));
}

extern "C" int sqdet_convdet_scores_supported(int cin, int anchors_per_grid, int classes, int dtype) {
  const int cout = anchors_per_grid * (classes + 5);
  if (conv_algo() != 0 || !convdet_score_supported(cout, anchors_per_grid, classes, dtype)) return 0;
  const ConvGeom g = conv_geom(3, cin, cout, dtype);
  return (!g.gather && g.nt == 5 && g.ngroups == 1 && g.nchunk >= 8 && g.nchunk % 4 == 0) ? 1 : 0;
}

extern "C" int sqdet_conv2d_res_nhwc_fwd(const void* x, const void* w_packed, const float* bias, const void* residual, void* y, int n,
                                         int h, int w, int cin, int cout, int k, int stride, int pad_mode, int relu, int dtype,
                                         int y_cstride, int y_coffset, sqdet_stream_t stream) {
  SQDET_REQUIRE(residual, "conv2d_res: null residual");
  return conv2d_launch_res(x, w_packed, bias, y, n, h, w, cin, cout, k, stride, pad_mode, relu, dtype, y_cstride, y_coffset, cin, 0, 1,
                           nullptr, residual, as_stream(stream));
}

extern "C" int sqdet_conv2d_add_nhwc_fwd(const void* x, const void* w_packed, const float* bias, void* y_inout, int n,
                                         int h, int w, int cin, int cout, int k, int stride, int pad_mode, int relu,
                                         int dtype, int y_cstride, int y_coffset, sqdet_stream_t stream) {
  return conv2d_launch_ex(x, w_packed, bias, y_inout, n, h, w, cin, cout, k, stride, pad_mode, relu, dtype, y_cstride,
                          y_coffset, cin, 0, 1, as_stream(stream));
}
