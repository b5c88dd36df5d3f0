// Fused network stem for gfx950: conv (KSxKS, stride 2, Cin = 3) + bias + ReLU + max-pool 3x3/s2,
// i.e. conv1 + pool1 of SqueezeDet (reference src/nets/squeezeDet.py:40-44: 3x3/s2 SAME + pool SAME)
// and SqueezeDet+ (src/nets/squeezeDetPlus.py:40-44: 7x7/s2 VALID + pool VALID) in ONE launch.
//
// Why: unfused, conv1 writes 188x621x64 and pool1 reads it back -- 36.4 MB of the 133 MB per image
// (fp16).  Fused, the conv activations never leave the CU: HBM traffic is the 2.8 MB input and the
// 3.7 MB pooled output.
//
// One 256-thread workgroup produces a 4 x 16 tile of POOLED pixels:
//   1. the input halo patch (19 x 67 pixels for 3x3) is staged into LDS;
//   2. the 9 x 33 conv outputs under the tile are computed on MFMA, 16 flattened conv pixels per
//      block; the B operand (K' = KS*KS*3 im2col patch, HWIO flattening = the packer's "gather"
//      order) is gathered element-wise from LDS; bias + ReLU applied, result written to an LDS conv
//      tile (row stride Cout*sizeof(T)+16 B: the +16 makes the 8-lane ds_write_b128 groups conflict
//      free); conv pixels outside the conv output get -inf so the pool ignores them (TF SAME max-pool);
//   3. each thread max-reduces 3x3 conv pixels x 16 B of channels from LDS and stores 16 B.
#include "stem.h"

namespace sqdet {

constexpr int SPH = 4, SPW = 16;              // pooled tile
constexpr int SNR = 2 * SPH + 1, SNC = 2 * SPW + 1;  // conv pixels under the tile: 9 x 33
constexpr int SNPIX = SNR * SNC;               // 297


template <typename T, int KS, int NT>
__global__ __launch_bounds__(256) void stem_conv_pool(StemArgs a) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  constexpr int TR = 2 * (SNR - 1) + KS;        // input rows staged
  constexpr int TC = 2 * (SNC - 1) + KS;        // input cols staged
  constexpr int IN_BYTES = (TR * TC * 3 * (int)sizeof(T) + 15) / 16 * 16;
  constexpr int CPIX = NT * 16 * (int)sizeof(T) + 16;  // conv tile pixel stride in bytes
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  T* lin = reinterpret_cast<T*>(lds);
  unsigned char* lconv = lds + IN_BYTES;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  int b = blockIdx.x;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int py0 = ty * SPH, px0 = tx * SPW;
  const int cy0 = 2 * py0 - a.ptp, cx0 = 2 * px0 - a.plp;   // first conv pixel under the tile (may be -1)
  const int iy0 = 2 * cy0 - a.ptc, ix0 = 2 * cx0 - a.plc;   // first input pixel staged

  // ---- 1. stage the input patch (zero outside the image: conv zero padding) ----
  // Thread t owns element column t of every staged row (TC*3 <= 256 elements per row): one address
  // increment and one wave-uniform row test per load, all TR loads in flight before the LDS stores.
  const T* x = reinterpret_cast<const T*>(a.x);
  {
    const int t = threadIdx.x;
    const int c = t / 3;
    const int ix = ix0 + c;
    const bool col_ok = t < TC * 3 && ix >= 0 && ix < a.W;
    const T* src = x + (((size_t)n * a.H) * a.W + (col_ok ? ix : 0)) * 3 + (t - c * 3);
    T stg[TR];
#pragma unroll
    for (int r = 0; r < TR; ++r) {
      const int iy = iy0 + r;
      stg[r] = (col_ok && iy >= 0 && iy < a.H) ? src[(size_t)iy * a.W * 3] : (T)0;
    }
    if (t < TC * 3) {
#pragma unroll
      for (int r = 0; r < TR; ++r) lin[r * (TC * 3) + t] = stg[r];
    }
  }
  __syncthreads();

  // ---- 2. conv on MFMA, 16 flattened conv pixels per block ----
  constexpr int NBLK = (SNPIX + 15) / 16;
  constexpr int NCHK = (KS * KS * 3 + KC - 1) / KC;     // K-chunks of the im2col patch
  constexpr bool PRE = NCHK * KG <= 16;                 // gather offsets precomputed per lane
  f32x4 bias[NT];
  const int cb = g * 4 * NT;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    bias[t] = cb + t * 4 < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  // element offset (within the staged patch) of k-slot e of chunk c for this lane group; -1 = padding slot
  auto slot_off = [&](int c, int e) -> int {
    const int kq = c * KC + g * KG + e;
    const int tap = kq / 3, ch = kq - tap * 3;
    const int dy = tap / KS, dx = tap - dy * KS;
    return kq < KS * KS * 3 ? (dy * TC + dx) * 3 + ch : -1;
  };
  int offs[PRE ? NCHK * KG : 1];
  i32x4 afr[PRE ? NCHK * NT : 1];   // weights resident in registers when the patch is small (3x3)
  if constexpr (PRE) {
#pragma unroll
    for (int c = 0; c < NCHK; ++c)
#pragma unroll
      for (int e = 0; e < KG; ++e) offs[c * KG + e] = slot_off(c, e);
    const i32x4* wp0 = reinterpret_cast<const i32x4*>(a.wp) + lane;
#pragma unroll
    for (int i = 0; i < NCHK * NT; ++i) afr[i] = wp0[i * 64];
  }
  const i32x4* wp = reinterpret_cast<const i32x4*>(a.wp) + lane;
  typedef T TV __attribute__((ext_vector_type(KG)));

  for (int blk = wave; blk < NBLK; blk += 4) {
    int q = blk * 16 + j;
    const bool inb = q < SNPIX;
    if (!inb) q = SNPIX - 1;
    const int cr = q / SNC, cc = q - cr * SNC;
    const T* patch = lin + (2 * cr * TC + 2 * cc) * 3;   // patch origin in the staged input
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCHK; ++c) {
      TV bv;
#pragma unroll
      for (int e = 0; e < KG; ++e) {
        const int o = PRE ? offs[PRE ? c * KG + e : 0] : slot_off(c, e);
        bv[e] = o >= 0 ? patch[o] : (T)0;
      }
      const i32x4 bfrag = __builtin_bit_cast(i32x4, bv);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const i32x4 af = PRE ? afr[PRE ? c * NT + t : 0] : wp[(c * NT + t) * 64];
        mma16<T>(acc[t], af, bfrag);
      }
    }
    // bias + ReLU; conv pixels outside the conv output are -inf for the pool
    const int cy = cy0 + cr, cx = cx0 + cc;
    const bool valid = cy >= 0 && cy < a.Hc && cx >= 0 && cx < a.Wc;
    if (inb) {
      T* dst = reinterpret_cast<T*>(lconv + q * CPIX) + cb;
      if (valid) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          f32x4 v = acc[t] + bias[t];
          v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
          store4<T>(dst + t * 4, v);
        }
      } else {
        const float ninf = -__builtin_huge_valf();
#pragma unroll
        for (int t = 0; t < NT; ++t) store4<T>(dst + t * 4, f32x4{ninf, ninf, ninf, ninf});
      }
    }
  }
  __syncthreads();

  // ---- 3. 3x3/s2 max-pool out of LDS, 16 bytes of channels per thread ----
  typedef T PV __attribute__((ext_vector_type(KG)));
  const int cgroups = a.Cout / KG;
  T* y = reinterpret_cast<T*>(a.y);
  for (int idx = threadIdx.x; idx < SPH * SPW * cgroups; idx += 256) {
    const int cg = idx % cgroups;
    const int pp = idx / cgroups;
    const int pr = pp / SPW, pc = pp - pr * SPW;
    const int py = py0 + pr, px = px0 + pc;
    if (py >= a.Hp || px >= a.Wp) continue;
    const unsigned char* w0 = lconv + ((2 * pr) * SNC + 2 * pc) * CPIX + cg * 16;
    PV m = *reinterpret_cast<const PV*>(w0);
#pragma unroll
    for (int t9 = 1; t9 < 9; ++t9) {
      const PV v = *reinterpret_cast<const PV*>(w0 + ((t9 / 3) * SNC + (t9 % 3)) * CPIX);
      m = __builtin_elementwise_max(m, v);
    }
    *reinterpret_cast<PV*>(y + (((size_t)n * a.Hp + py) * a.Wp + px) * a.y_cstride + a.y_coffset + cg * KG) = m;
  }
}

template <typename T, int KS, int NT>
static int launch_stem(const StemArgs& a, hipStream_t st) {
  constexpr int TR = 2 * (SNR - 1) + KS, TC = 2 * (SNC - 1) + KS;
  constexpr int IN_BYTES = (TR * TC * 3 * (int)sizeof(T) + 15) / 16 * 16;
  constexpr int CPIX = NT * 16 * (int)sizeof(T) + 16;
  const size_t lds = (size_t)IN_BYTES + (size_t)SNPIX * CPIX;
  static bool attr_done = false;
  if (lds > 65536 && !attr_done) {
    SQDET_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_pool<T, KS, NT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  hipLaunchKernelGGL((stem_conv_pool<T, KS, NT>), dim3((unsigned)(a.N * a.tiles_x * a.tiles_y)), dim3(256), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

// conv(k, stride 2, Cin 3) + bias + relu + maxpool(3, stride 2).  *handled=false: not eligible.
int stem_launch(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cout, int k,
                int conv_pad, int pool_pad, int dtype, int y_cstride, int y_coffset, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (!((k == 3 && cout == 64) || (k == 7 && (cout == 96 || cout == 64)))) return SQDET_OK;   // SqueezeDet, SqueezeDet+, ResNet50
  const ConvGeom g = conv_geom(k, 3, cout, dtype);
  if (!g.gather || g.ngroups != 1) return SQDET_OK;
  StemArgs a;
  a.x = x; a.wp = w_packed; a.bias = bias; a.y = y;
  a.N = n; a.H = h; a.W = w;
  a.Hc = out_size(h, k, 2, conv_pad); a.Wc = out_size(w, k, 2, conv_pad);
  a.Hp = out_size(a.Hc, 3, 2, pool_pad); a.Wp = out_size(a.Wc, 3, 2, pool_pad);
  a.ptc = pad_before(h, k, 2, conv_pad); a.plc = pad_before(w, k, 2, conv_pad);
  a.ptp = pad_before(a.Hc, 3, 2, pool_pad); a.plp = pad_before(a.Wc, 3, 2, pool_pad);
  a.Cout = cout; a.nchunk = g.nchunk; a.kdim = g.kdim;
  a.tiles_x = (a.Wp + SPW - 1) / SPW; a.tiles_y = (a.Hp + SPH - 1) / SPH;
  a.y_cstride = y_cstride; a.y_coffset = y_coffset;
  a.ws2 = nullptr; a.bs2 = nullptr; a.s_out = nullptr;
  if (a.Hp <= 0 || a.Wp <= 0) return SQDET_OK;
  if (tune(TUNE_STEM_ALGO) == 0) {  // default for the fp16 3x3 stem: the persistent kernel (stem3.hip)
    const int rc3 = stem_pers_launch(a, k, dtype, st, handled);
    if (rc3 != SQDET_OK || *handled) return rc3;
  }
  if (tune(TUNE_STEM_ALGO) != 1) {  // otherwise the in-register-pool strip kernel (stem2.hip)
    const int rc2 = stem_strip_launch(a, k, dtype, st, handled);
    if (rc2 != SQDET_OK || *handled) return rc2;
  }
  int rc;
  if (dtype == SQDET_F16)
    rc = k == 3 ? launch_stem<f16, 3, 4>(a, st) : cout == 96 ? launch_stem<f16, 7, 6>(a, st) : launch_stem<f16, 7, 4>(a, st);
  else
    rc = k == 3 ? launch_stem<float, 3, 4>(a, st) : cout == 96 ? launch_stem<float, 7, 6>(a, st) : launch_stem<float, 7, 4>(a, st);
  if (rc != SQDET_OK) return rc;
  *handled = true;
  return SQDET_OK;
}

// conv1 + pool1 + the next layer's squeeze1x1 (64 -> 16 couts) in one launch: only the squeeze tensor [n, Hp, Wp, 16] is written
// (persistent fp16 3x3 stem only).  *handled = false: not eligible.
int stem_squeeze_launch(const void* x, const void* w_packed, const float* bias, const void* ws2_packed, const float* bs2,
                        void* s_out, int n, int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype,
                        hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0 || tune(TUNE_STEM_ALGO) != 0 || k != 3 || cout != 64 || s2 != 16 || dtype != SQDET_F16) return SQDET_OK;
  const ConvGeom g = conv_geom(k, 3, cout, dtype), gs = conv_geom(1, cout, s2, dtype);
  if (!g.gather || g.ngroups != 1 || gs.gather || gs.nchunk != 2 || gs.nt != 1 || gs.ngroups != 1) return SQDET_OK;
  StemArgs a;
  a.x = x; a.wp = w_packed; a.bias = bias; a.y = nullptr;
  a.N = n; a.H = h; a.W = w;
  a.Hc = out_size(h, k, 2, conv_pad); a.Wc = out_size(w, k, 2, conv_pad);
  a.Hp = out_size(a.Hc, 3, 2, pool_pad); a.Wp = out_size(a.Wc, 3, 2, pool_pad);
  a.ptc = pad_before(h, k, 2, conv_pad); a.plc = pad_before(w, k, 2, conv_pad);
  a.ptp = pad_before(a.Hc, 3, 2, pool_pad); a.plp = pad_before(a.Wc, 3, 2, pool_pad);
  a.Cout = cout; a.nchunk = g.nchunk; a.kdim = g.kdim;
  a.tiles_x = a.tiles_y = 0;
  a.y_cstride = cout; a.y_coffset = 0;
  a.ws2 = ws2_packed; a.bs2 = bs2; a.s_out = s_out;
  if (a.Hp <= 0 || a.Wp <= 0) return SQDET_OK;
  return stem_pers_launch(a, k, dtype, st, handled);
}

bool stem_squeeze_eligible(int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype, int n) {
  if (conv_algo() != 0 || tune(TUNE_STEM_ALGO) != 0 || k != 3 || cout != 64 || s2 != 16 || dtype != SQDET_F16) return false;
  const int plc = pad_before(w, k, 2, conv_pad);
  const int hc = out_size(h, k, 2, conv_pad), wc = out_size(w, k, 2, conv_pad);
  const int hp = out_size(hc, 3, 2, pool_pad), wp = out_size(wc, 3, 2, pool_pad);
  if (hp <= 0 || wp <= 0 || w % 2 != 0 || plc % 2 != 0) return false;
  if ((size_t)h * w * 6 >= (1ull << 31) || w * 6 < 2 * 44 * 16) return false;          // as stem_pers_launch
  if ((size_t)n * hp * wp * cout * 2 >= (1ull << 31)) return false;
  const ConvGeom gs = conv_geom(1, cout, s2, dtype);
  return !gs.gather && gs.nchunk == 2 && gs.nt == 1 && gs.ngroups == 1;
}

}  // namespace sqdet

extern "C" int sqdet_stem_conv_pool_squeeze_fwd(const void* x, const void* w_packed, const float* bias, const void* w_next_s_packed,
                                                const float* b_next_s, void* sq_out, int n, int h, int w, int cout, int k,
                                                int conv_pad_mode, int pool_pad_mode, int next_s, int dtype, sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(x && w_packed && bias && w_next_s_packed && b_next_s && sq_out, "stem_squeeze: null pointer");
  bool handled = false;
  const int rc = stem_squeeze_launch(x, w_packed, bias, w_next_s_packed, b_next_s, sq_out, n, h, w, cout, k, conv_pad_mode,
                                     pool_pad_mode, next_s, dtype, as_stream(stream), &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "stem_squeeze: needs the float16 3x3 / 64-cout stem of even width followed by a 64 -> 16 squeeze1x1");
  return SQDET_OK;
}

extern "C" int sqdet_stem_conv_pool_squeeze_supported(int h, int w, int cout, int k, int conv_pad_mode, int pool_pad_mode,
                                                      int next_s, int dtype, int n) {
  return sqdet::stem_squeeze_eligible(h, w, cout, k, conv_pad_mode, pool_pad_mode, next_s, dtype, n) ? 1 : 0;
}

// C ABI: the fused stem as its own entry point (see include/sqdet.h).
extern "C" int sqdet_stem_conv_pool_fwd(const void* x, const void* w_packed, const float* bias, void* y, int n, int h,
                                        int w, int cout, int k, int conv_pad_mode, int pool_pad_mode, int dtype,
                                        sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(x && w_packed && bias && y, "stem: null pointer");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "stem: bad dtype");
  bool handled = false;
  const int saved = conv_algo();
  SQDET_UNSUPPORTED(saved != 0, "stem: fused kernel disabled by conv_algo=generic");
  int rc = stem_launch(x, w_packed, bias, y, n, h, w, cout, k, conv_pad_mode, pool_pad_mode, dtype, cout, 0,
                       as_stream(stream), &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "stem: only (k=3, cout=64) and (k=7, cout=96 or 64) stems are fused");
  return SQDET_OK;
}
