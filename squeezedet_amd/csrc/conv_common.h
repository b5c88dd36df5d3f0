// Shared device helpers of the conv kernels (conv.hip, conv3x3.hip).
#pragma once
#include "common.h"

namespace sqdet {

template <typename T> struct Tr;
template <> struct Tr<f16> { static constexpr int KG = 8; };
template <> struct Tr<float> { static constexpr int KG = 4; };

template <typename T>
__device__ __forceinline__ void mma16(f32x4& acc, const i32x4& a, const i32x4& b);
template <>
__device__ __forceinline__ void mma16<f16>(f32x4& acc, const i32x4& a, const i32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ void mma16<float>(f32x4& acc, const i32x4& a, const i32x4& b) {
  f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
}

struct ConvArgs {
  const void* x;
  const void* wp;
  const float* bias;
  void* y;
  int N, H, W, Cin, Cout, k, stride, pt, pl, Ho, Wo;
  int P;        // N*Ho*Wo output pixels
  int ntiles;   // ceil(P / (16*MT))
  int nchunk, steps, ngroups;
  int y_cstride, y_coffset, relu;
  // generic kernel only: the input may be a channel slice [x_coffset, x_coffset+Cin) of rows
  // x_cstride channels wide, and the result may be ADDED to y (backward-data of a fire module:
  // d(squeeze) = dgrad_1x1(dY[:, :e1]) + dgrad_3x3(dY[:, e1:])).
  int x_cstride, x_coffset, accum;
  // accum only: the tensor the result is added to when it is NOT y itself (same layout as y; NULL = y).  ResNet50's training forward
  // keeps the shortcut for the backward pass: y = relu(conv(x) + b + res) without a copy of the shortcut first.  Honoured by
  // conv1x1_pipe; every other kernel gets a copy res -> y in front (conv2d_launch_res).
  const void* res;
  // backward-data only: the result is the gradient w.r.t. a ReLU OUTPUT r (same layout as y); it is zeroed where r <= 0 --
  // the ReLU backward of the layer below, taken in this conv's epilogue instead of a separate pass over the tensor
  const void* relu_of;
  // ConvDet only (convdet.hip, float16): when not NULL, the epilogue also writes one float32 SCORE per anchor --
  // det_probs of interpret_output (nn_skeleton.py:150-170, 274-283) computed from the float16-rounded preds it stores --
  // to scores[N, H*W*score_apg]; Cout = score_apg * (score_classes + 5), score_classes == 3
  float* scores;
  int score_apg, score_classes;
};

template <typename T>
__device__ __forceinline__ void store4(T* dst, const f32x4& v);
template <>
__device__ __forceinline__ void store4<f16>(f16* dst, const f32x4& v) {
  f16x4 h = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
  *reinterpret_cast<f16x4*>(dst) = h;
}
template <>
__device__ __forceinline__ void store4<float>(float* dst, const f32x4& v) {
  *reinterpret_cast<f32x4*>(dst) = v;
}

// Stores 4*NT consecutive output channels held by one lane (nt_valid < NT: only the first nt_valid
// 4-channel pieces exist).  fp16 rows go out as 16-byte vectors when the pieces pair up.
template <typename T, int NT>
__device__ __forceinline__ void store_couts(T* dst, const f32x4 (&v)[NT], int nt_valid) {
  if (nt_valid == NT) {
    // 16-byte stores need a 16-byte aligned row segment: even NT (8*NT bytes per lane) and a row
    // stride / channel offset that are multiples of 8 halves -- checked on the address itself.
    if (sizeof(T) == 2 && (NT & 1) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
#pragma unroll
      for (int t = 0; t + 1 < NT; t += 2) {
        f16x8 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3],
                   (f16)v[t + 1][0], (f16)v[t + 1][1], (f16)v[t + 1][2], (f16)v[t + 1][3]};
        *reinterpret_cast<f16x8*>(dst + t * 4) = h;
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) store4<T>(dst + t * 4, v[t]);
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt_valid) store4<T>(dst + t * 4, v[t]);
  }
}

// Which kernel family conv2d_launch may pick: 0 = auto (fast paths when eligible),
// 1 = generic only (conv_direct / conv_gather).  Set from SQDET_CONV_ALGO=generic (tests, A/B).
int conv_algo();
// experiment knobs set through sqdet_set_option (0 = built-in heuristic)
// fire_fuse: 0 / 1 = a fire module is one fused launch wherever a fused kernel takes it (the default since round 5), 2 = never,
// 10 = the round-1..4 rule (only maps of <= 100 k pixels); stem_algo: 0 phase kernel (stem4.hip), else persistent strip-lane kernel (stem3.hip), else strip kernel (in-register pool), whichever is eligible first; 3 skips the phase kernel; 2 strip kernel only
// g1_wr / g1_mbw / g1_ntw: conv1x1_pipe's wave layout (waves along the pixel blocks: 1, 2, 4), pixel blocks per wave (2, 4, 8) and cout tiles per wave -- tools/g1_sweep.py
enum { TUNE_C1_WAVES = 0, TUNE_C1_MT = 1, TUNE_C1_MIN_TILES = 2, TUNE_FIRE_FUSE = 3, TUNE_STEM_ALGO = 4, TUNE_DBG = 5, TUNE_G1_WR = 6, TUNE_G1_MBW = 7, TUNE_G1_NTW = 8, TUNE_G1_NS = 9 };
int tune(int which);

// fire2.hip: persistent streaming fused fire for the large, few-channel modules
bool fire_stream_eligible(int cin, int s, int e1, int e3, int dtype);
// (the *_keep forms also write the module's squeeze tensor: training)
int fire_stream_launch_keep(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                            const float* b3, void* sq_out, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                            hipStream_t st, bool* handled);
int fire_fused_launch_keep(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                           const float* b3, void* sq_out, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                           hipStream_t st, bool* handled);
int fire_stream_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                       const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                       hipStream_t st, bool* handled);
// pool != 0: fire module + max_pool 3x3/s2/SAME in one launch; y is the pooled tensor
int fire_stream_launch_ex(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                          const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                          int pool, hipStream_t st, bool* handled);
// the expand half of a fire module from its squeeze tensor (+ the 3x3/s2 SAME max-pool behind it when pool != 0)
bool fire_expand_stream_eligible(int s, int e1, int e3, int dtype);
int fire_expand_stream_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3, void* y,
                              int n, int h, int w, int s, int e1, int e3, int dtype, int pool, hipStream_t st, bool* handled);
// whole fire module from x, its concat tensor replaced by the NEXT module's squeeze tensor (fire2 / fire4 of SqueezeDet)
bool fire_squeeze_next_eligible(int cin, int s, int e1, int e3, int s2, int dtype);
int fire_squeeze_next_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                             const float* b3, const void* ws2, const float* bs2, void* s_out, int n, int h, int w, int cin,
                             int s, int e1, int e3, int s2, int dtype, hipStream_t st, bool* handled);
bool fire_expand_squeeze_next_eligible(int s, int e1, int e3, int s2, int pool, int dtype);
int fire_expand_squeeze_next_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3,
                                    const void* ws2, const float* bs2, void* s_out, int n, int h, int w, int s, int e1, int e3,
                                    int s2, int pool, int dtype, hipStream_t st, bool* handled);
// fire3.hip: the same launch as a DMA-fed kernel sized for four waves per SIMD (SqueezeDet's four shapes)
int fire_dma_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3, const void* ws2,
                    const float* bs2, void* s_out, int n, int h, int w, int s, int e1, int e3, int s2, int pool, int dtype,
                    hipStream_t st, bool* handled);
int conv3x3_tile_launch(const ConvArgs& a, const ConvGeom& g, int dtype, hipStream_t st, bool* handled);
// conv3x3.hip: both expands of a fire module from ONE staged squeeze tile (the tile kernel's PAIR form)
bool conv3x3_pair_eligible(int n, int h, int w, int s, int e1, int e3, int dtype);
int conv3x3_pair_launch(const void* sq_in, const void* w3, const float* b3, const void* w1, const float* b1, void* y, int n, int h, int w,
                        int s, int e1, int e3, int dtype, hipStream_t st, bool* handled);
// convdet.hip: the score epilogue's shapes; conv.hip: ConvDet + scores in one launch (sqdet_convdet_fwd)
bool convdet_score_supported(int cout, int apg, int classes, int dtype);
int convdet_scored_launch(const void* x, const void* w_packed, const float* bias, void* preds, float* scores, int n, int h, int w,
                          int cin, int apg, int classes, int dtype, hipStream_t st);
int conv1x1_stream_launch(const ConvArgs& a, const ConvGeom& g, int dtype, hipStream_t st, bool* handled);
int conv1x1_tile_launch(const ConvArgs& a, const ConvGeom& g, int dtype, hipStream_t st, bool* handled);   // gemm1x1.hip
int conv1x1_deepk_launch(const ConvArgs& a, const ConvGeom& g, int dtype, hipStream_t st, bool* handled);  // conv1x1k.hip

}  // namespace sqdet
