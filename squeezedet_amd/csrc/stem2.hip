// Fused stem, strip form: conv (KSxKS, stride 2, Cin = 3) + bias + ReLU + max-pool 3x3/s2 with the
// pooling done IN REGISTERS (reference src/nets/squeezeDet.py:40-44, src/nets/squeezeDetPlus.py:40-44).
//
// A 256-thread workgroup produces 8 pooled rows x 28 pooled columns; each of its 4 waves owns a strip
// of 16 conv columns (= 7 pooled columns) and walks the 17 conv rows under the tile:
//   * one MFMA block = the 16 conv columns of one conv row (K' = KS*KS*3 im2col patch gathered from
//     the LDS input patch, weights resident in registers for the 3x3 stem);
//   * the C/D layout puts conv column j in lane j of every 16-lane DPP row, so the HORIZONTAL
//     3-tap/stride-2 max is two DPP row shifts (row_shl:1, row_shl:2) + two max per register --
//     the strip starts at conv column 2*p0 - pad so windows never straddle strips;
//   * the VERTICAL max keeps one previous row in registers: out[q] = max(h[2q], h[2q+1], h[2q+2]);
//   * lanes j = 0,2,..,12 store the 7 pooled pixels, 16 channels (32 B in fp16) per lane.
//   * ReLU is applied once to the pooled value (it commutes with
//     max and with the monotonic fp16 rounding), the vertical max runs before the horizontal one: the
//     epilogue is ~1/3 of the VALU work of the straightforward order -- this kernel is VALU-bound.
// Conv activations never touch LDS or HBM; LDS holds only the 35 x 117 x 3 input patch (24.6 KB,
// 6 workgroups per CU).  Out-of-range conv pixels are -inf (TF SAME max-pool never picks padding).
#include "stem.h"

namespace sqdet {

constexpr int ZPR = 8;                    // pooled rows per workgroup
constexpr int ZSP = 7;                    // pooled columns per wave strip
constexpr int ZCR = 2 * ZPR + 1;          // conv rows under the tile (17)
constexpr int ZCC = 4 * 2 * ZSP + 2;      // conv columns under the tile (58)

template <typename T> struct Pk;          // 16 channels of one pixel as 32-bit registers
template <> struct Pk<f16> { static constexpr int R = 8; };
template <> struct Pk<float> { static constexpr int R = 16; };

template <typename T>
__device__ __forceinline__ unsigned int pkmax(unsigned int a, unsigned int b);
// Written as the instruction itself: through fmaxf / elementwise_max the compiler first canonicalises
// both operands (one extra v_pk_max_f16 x, x per input), which more than doubled the pooling VALU work.
template <>
__device__ __forceinline__ unsigned int pkmax<f16>(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <>
__device__ __forceinline__ unsigned int pkmax<float>(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// First chunk of a K loop: the accumulator input is the inline constant 0 (no register zeroing).
template <typename T>
__device__ __forceinline__ f32x4 mma16_first(const i32x4& a, const i32x4& b);
template <>
__device__ __forceinline__ f32x4 mma16_first<f16>(const i32x4& a, const i32x4& b) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b),
                                                f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma16_first<float>(const i32x4& a, const i32x4& b) {
  const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
  f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
  return acc;
}

// -DSQDET_FIRE_TIMING (experiments only): per-wave s_memtime totals of the kernel's three segments (tools/stem_timing.py).
// Measured (batch 32, 375x1242): staging the input patch 31 % of a workgroup's life (one memory round trip: 16-byte
// instead of 4-byte loads changed nothing), barrier 8 %, im2col gather + MFMA + pooling + stores 61 % -- of which the
// 68 MFMAs are 7 %: the kernel is bound by the VALU / LDS work of the gather and the pooled epilogue.
#ifdef SQDET_FIRE_TIMING
__device__ unsigned long long g_stem_timing[2048 * 8];
#define ST_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc[k] += now_ - st_last; st_last = now_; } while (0)
#else
#define ST_MARK(k) do {} while (0)
#endif

template <typename T, int KS, int NT, bool ALIGNED4>
__global__ __launch_bounds__(256) void stem_strip(StemArgs a) {
#ifdef SQDET_FIRE_TIMING
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  constexpr int TR = 2 * (ZCR - 1) + KS;            // staged input rows
  constexpr int TC = 2 * (ZCC - 1) + KS;            // staged input cols
  constexpr int LROW = (TC * 3 + 3) / 4 * 4;        // LDS row stride in elements (4-byte aligned rows)
  constexpr int NCHK = (KS * KS * 3 + KC - 1) / KC;
  constexpr bool PRE = NCHK * KG <= 16;
  static_assert(NT == 4 || NT == 6, "16*NT couts");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  T* lin = reinterpret_cast<T*>(lds);

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  // XCD-aware tile order: workgroup b runs on XCD b % 8 (own L2 each); every XCD gets a contiguous band of
  // tiles so the halos shared by neighbouring tiles are fetched into ONE L2 instead of up to eight.
  int b = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));   // gridDim.x is a multiple of 8
  if (b >= a.N * a.tiles_x * a.tiles_y) return;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int py0 = ty * ZPR, px0 = tx * (4 * ZSP);
  const int cy0 = 2 * py0 - a.ptp, cx0 = 2 * px0 - a.plp;
  const int iy0 = 2 * cy0 - a.ptc, ix0 = 2 * cx0 - a.plc;

  // ---- stage the input patch; all loads in flight before the LDS stores ----
  const T* x = reinterpret_cast<const T*>(a.x);
  if constexpr (ALIGNED4) {
    // rows start 4-byte aligned (W even, even left pad): one dword = 2 elements per load
    constexpr int EPL = 4 / (int)sizeof(T);            // elements per dword (2 for f16, 1 for f32)
    constexpr int DPR = (TC * 3 + EPL - 1) / EPL;      // dwords per staged row
    constexpr int NIT = (DPR + 255) / 256;
    unsigned int stg[TR][NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int d = it * 256 + threadIdx.x;
      const int e0 = d * EPL;                          // first element of the dword within the row
      const int c0 = e0 / 3, c1 = (e0 + EPL - 1) / 3;  // pixels the dword touches
      const bool lo_ok = d < DPR && ix0 + c0 >= 0 && ix0 + c0 < a.W;
      const bool hi_ok = d < DPR && ix0 + c1 >= 0 && ix0 + c1 < a.W;
#pragma unroll
      for (int r = 0; r < TR; ++r) {
        const int iy = iy0 + r;
        unsigned int v = 0;
        if ((lo_ok || hi_ok) && iy >= 0 && iy < a.H) {
          const long eoff = (((long)n * a.H + iy) * a.W + ix0) * 3 + e0;   // may be < 0 at the left edge
          if (lo_ok && hi_ok) {
            v = *reinterpret_cast<const unsigned int*>(x + eoff);
          } else if (EPL == 2) {                       // dword straddles the image edge: take the valid half
            const unsigned short h = *reinterpret_cast<const unsigned short*>(x + eoff + (lo_ok ? 0 : 1));
            v = lo_ok ? (unsigned int)h : ((unsigned int)h << 16);
          }
        }
        stg[r][it] = v;
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int d = it * 256 + threadIdx.x;
      if (d < DPR) {
#pragma unroll
        for (int r = 0; r < TR; ++r) *reinterpret_cast<unsigned int*>(lin + r * LROW + d * EPL) = stg[r][it];
      }
    }
  } else {
    constexpr int NIT = (TC * 3 + 255) / 256;
#pragma unroll 1
    for (int r0 = 0; r0 < TR; r0 += 8) {
      T stg[8][NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = it * 256 + threadIdx.x;
        const int c = e / 3;
        const int ix = ix0 + c;
        const bool col_ok = e < TC * 3 && ix >= 0 && ix < a.W;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int iy = iy0 + r0 + rr;
          stg[rr][it] = (col_ok && r0 + rr < TR && iy >= 0 && iy < a.H) ? x[(((long)n * a.H + iy) * a.W + ix) * 3 + (e - c * 3)] : (T)0;
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int e = it * 256 + threadIdx.x;
        if (e < TC * 3) {
#pragma unroll
          for (int rr = 0; rr < 8; ++rr)
            if (r0 + rr < TR) lin[(r0 + rr) * LROW + e] = stg[rr][it];
        }
      }
    }
  }
  ST_MARK(0);
  __syncthreads();
  ST_MARK(1);

  // ---- per-lane constants ----
  const int cb = g * 4 * NT;                              // this lane's 4*NT = 16 (or 24) consecutive couts
  f32x4 bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    bias[t] = cb + t * 4 < a.Cout ? *reinterpret_cast<const f32x4*>(a.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  auto slot_off = [&](int c, int e) -> int {
    const int kq = c * KC + g * KG + e;
    const int tap = kq / 3, ch = kq - tap * 3;
    const int dy = tap / KS, dx = tap - dy * KS;
    return kq < KS * KS * 3 ? dy * LROW + dx * 3 + ch : -1;
  };
  int offs[PRE ? NCHK * KG : 1];
  i32x4 afr[PRE ? NCHK * NT : 1];
  const i32x4* wp = reinterpret_cast<const i32x4*>(a.wp) + lane;
  if constexpr (PRE) {
#pragma unroll
    for (int c = 0; c < NCHK; ++c)
#pragma unroll
      for (int e = 0; e < KG; ++e) offs[c * KG + e] = slot_off(c, e);
#pragma unroll
    for (int i = 0; i < NCHK * NT; ++i) afr[i] = wp[i * 64];
  }
  typedef T TV __attribute__((ext_vector_type(KG)));
  constexpr int RT = NT * 4 * (int)sizeof(T) / 4;         // 32-bit registers per lane per conv pixel (8 f16 / 16 f32 at NT=4)
  const int cc = 2 * ZSP * wave + j;                      // conv column (relative to cx0) of this lane
  const int cx = cx0 + cc;
  const bool col_ok = cx >= 0 && cx < a.Wc;
  const unsigned int NEG = sizeof(T) == 2 ? 0xfc00fc00u : 0xff800000u;   // -inf (packed)

  // conv row rr (relative to cy0) -> conv + bias of this lane's column, packed to storage type, NOT yet
  // rectified (ReLU commutes with max and with the monotonic f16 rounding: it is applied once, to the
  // pooled value).  Rows outside the conv map are -inf without computing anything (wave-uniform).
  const bool edge_cols = cx0 < 0 || cx0 + ZCC > a.Wc;     // workgroup-uniform: some lanes sit outside the conv map
  auto conv_row = [&](int rr, unsigned int (&v)[RT]) {
    const int cy = cy0 + rr;
    if (cy < 0 || cy >= a.Hc) {
#pragma unroll
      for (int i = 0; i < RT; ++i) v[i] = NEG;
      return;
    }
    const T* patch = lin + (2 * rr) * LROW + (2 * cc) * 3;
    f32x4 acc[NT];
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
        if (c == 0) acc[t] = mma16_first<T>(af, bfrag);
        else mma16<T>(acc[t], af, bfrag);
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      acc[t] += bias[t];    // after the accumulation, like every other conv kernel here (bitwise-equal results)
      if constexpr (sizeof(T) == 2) {
        typedef f16 h2 __attribute__((ext_vector_type(2)));
        const h2 lo = {(f16)acc[t][0], (f16)acc[t][1]}, hi = {(f16)acc[t][2], (f16)acc[t][3]};
        v[2 * t] = __builtin_bit_cast(unsigned int, lo);
        v[2 * t + 1] = __builtin_bit_cast(unsigned int, hi);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * t + e] = __float_as_uint(acc[t][e]);
      }
    }
    if (edge_cols && !col_ok) {
#pragma unroll
      for (int i = 0; i < RT; ++i) v[i] = NEG;
    }
  };

  // ---- walk the conv rows: vertical 3-max first (plain registers), then ONE horizontal 3-tap max per
  // pooled row (lane j takes columns j, j+1, j+2 of its 16-lane row: DPP row_shl), then ReLU ----
  T* y = reinterpret_cast<T*>(a.y);
  const int pxl = ZSP * wave + (j >> 1);                  // pooled column within the tile
  const int px = px0 + pxl;
  const bool store_lane = (j & 1) == 0 && j < 2 * ZSP && px < a.Wp && cb < a.Cout;
  unsigned int prev[RT], va[RT], vb[RT];
  conv_row(0, prev);
#pragma unroll 1
  for (int q = 0; q < ZPR; ++q) {
    const int py = py0 + q;
    if (py >= a.Hp) break;                                // workgroup-uniform
    conv_row(2 * q + 1, va);
    conv_row(2 * q + 2, vb);
    unsigned int o[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
      const unsigned int m = pkmax<T>(prev[i], pkmax<T>(va[i], vb[i]));
      const unsigned int s1 = (unsigned int)__builtin_amdgcn_update_dpp((int)NEG, (int)m, 0x101, 0xf, 0xf, false);
      const unsigned int s2 = (unsigned int)__builtin_amdgcn_update_dpp((int)NEG, (int)m, 0x102, 0xf, 0xf, false);
      o[i] = pkmax<T>(pkmax<T>(m, pkmax<T>(s1, s2)), 0u);   // 0u = +0.0 (packed): the ReLU
      prev[i] = vb[i];
    }
    if (store_lane) {
      unsigned int* dst = reinterpret_cast<unsigned int*>(y + (((size_t)n * a.Hp + py) * a.Wp + px) * a.y_cstride + a.y_coffset + cb);
#pragma unroll
      for (int i = 0; i < RT; i += 4) {
        // whole 4-cout pieces beyond Cout do not exist (Cout % 4 == 0): skip them
        if (cb + i * 4 / (int)sizeof(T) < a.Cout)
          *reinterpret_cast<i32x4*>(dst + i) = i32x4{(int)o[i], (int)o[i + 1], (int)o[i + 2], (int)o[i + 3]};
      }
    }
  }
  ST_MARK(2);
#ifdef SQDET_FIRE_TIMING
  if ((threadIdx.x & 63) == 0 && blockIdx.x * 4 + (threadIdx.x >> 6) < 2048)
    for (int k = 0; k < 8; ++k) g_stem_timing[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + k] = st_acc[k];
#endif
}

template <typename T, int KS, int NT>
static int launch_strip(StemArgs a, bool aligned4, hipStream_t st) {
  constexpr int TR = 2 * (ZCR - 1) + KS, TC = 2 * (ZCC - 1) + KS;
  constexpr int LROW = (TC * 3 + 3) / 4 * 4;
  const size_t lds = (size_t)TR * LROW * sizeof(T);
  a.tiles_x = (a.Wp + 4 * ZSP - 1) / (4 * ZSP);
  a.tiles_y = (a.Hp + ZPR - 1) / ZPR;
  const dim3 grid((unsigned)((a.N * a.tiles_x * a.tiles_y + 7) / 8 * 8));
  if (aligned4)
    hipLaunchKernelGGL((stem_strip<T, KS, NT, true>), grid, dim3(256), lds, st, a);
  else
    hipLaunchKernelGGL((stem_strip<T, KS, NT, false>), grid, dim3(256), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

int stem_strip_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  // 16-byte stores of the 16 (24) channels a lane owns
  if (a.y_cstride % 8 != 0 || a.y_coffset % 8 != 0) return SQDET_OK;
  const bool aligned4 = dtype == SQDET_F32 || (a.W % 2 == 0 && a.plc % 2 == 0);
  int rc;
  if (dtype == SQDET_F16)
    rc = k == 3 ? launch_strip<f16, 3, 4>(a, aligned4, st)
                : a.Cout == 96 ? launch_strip<f16, 7, 6>(a, aligned4, st) : launch_strip<f16, 7, 4>(a, aligned4, st);
  else
    rc = k == 3 ? launch_strip<float, 3, 4>(a, aligned4, st)
                : a.Cout == 96 ? launch_strip<float, 7, 6>(a, aligned4, st) : launch_strip<float, 7, 4>(a, aligned4, st);
  if (rc != SQDET_OK) return rc;
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_FIRE_TIMING
extern "C" int sqdet_debug_stem_timing(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_stem_timing), sizeof(unsigned long long) * count);
}
#endif
