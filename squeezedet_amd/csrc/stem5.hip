// Fused 7x7 / stride-2 stem for gfx950, float16: conv (Cin = 3, 64 or 96 couts) + bias + ReLU + max-pool 3x3 / s2 VALID in one launch --
// conv1 + pool1 of SqueezeDet+ (reference src/nets/squeezeDetPlus.py:40-44: VALID conv) and of ResNet50 (src/nets/resnet50_convDet.py:41-45:
// SAME conv, frozen batch norm folded into kernel + bias).
//
// The strip kernel (stem2.hip) gathers the 147-element im2col column of every conv pixel from a 3-channel LDS patch element by element --
// a pixel is 6 bytes, nothing is aligned -- and spends 93 % of its time on that gather and on the pooled epilogue (25.8 GFLOP in 136 us
// for SqueezeDet+ at batch 8; ResNet50's training forward ran conv_gather + maxpool3: 164 us).  Here the K axis is re-cut so that NO
// gather is left:
//   * the wave's input rows live in LDS with FOUR channels per pixel (8 bytes, the fourth is zero): the 3 -> 4 expansion happens once, in
//     the staging (one 12-byte load = two pixels -> one 16-byte LDS store);
//   * one MFMA K chunk (32 elements) = ONE kernel row: 8 pixels x 4 channels, the eighth pixel and the fourth channel multiply zero
//     weights.  For conv column j the chunk is the 64 contiguous, 16-byte aligned bytes at input column 2 j of the row, so a B fragment is
//     ONE conflict-free ds_read_b128 (lane (j, g): byte 16 j + 16 g of the row) -- and the fragment of input row R serves every conv row
//     that touches R (conv row c uses it as kernel row R - 2 c): 13 reads feed the 28 (conv row, kernel row) pairs of a 4-row step;
//   * the weights are re-cut to that order inside the kernel (packed gather order -> [kernel row][cout tile] fragments in LDS, once per
//     workgroup: the C-ABI keeps taking sqdet_conv_pack_weights' format) and read as A fragments, one ds_read_b128 per 4 MFMAs;
//   * a WAVE is the unit of work: a strip of 16 conv columns (7 pooled ones) x 16 conv rows (7 pooled rows) in four steps of 4 conv rows
//     x all cout tiles (96 / 64 accumulators); its input rows sit in a 16-row ring of its own, the rows of the next step are requested a
//     step ahead (registers) and written behind the step's fragment reads -- no barrier anywhere after the weight conversion;
//     (stand-alone, batch 8: 55 us for SqueezeDet+'s stem -- of which the MFMA loop is ~40 and the pooled epilogue ~18, measured by
//     leaving each out -- against 91 for the strip kernel; 40 against 98 us for ResNet50's)
//   * pooling in registers as in the strip kernel: vertical v_max3 over the step's rows (+ the carry of the last two rows), horizontal
//     3-tap / stride-2 max by two DPP row shifts, then bias + ReLU + rounding ONCE on the pooled value (all three commute with max).
// K is 7 x 32 = 224 instead of 147 (1.52x the MFMAs), which is what the alignment costs; the matrix pipe has the room.
// Accumulation order: kernel rows ascending, inside a row the MFMA's own order -- not the gather kernels' (147 products in chunks of
// 32): results agree with conv -> pool to the float32 summation order, i.e. float16-identical but for rare 1-ulp flips.
#include "stem.h"

namespace sqdet {
namespace {

constexpr int S5_RING = 16;                 // input rows in a wave's ring
constexpr int S5_PITCH = 304;               // bytes per ring row: 38 pixels x 8 B (conv column 15's chunk ends at pixel 37)
constexpr int S5_PAIRS = 19;                // 16-byte pixel pairs per ring row
constexpr int S5_STEPS = 4;                 // steps per item: 16 conv rows -> 7 pooled rows
constexpr int S5_PROWS = 2 * S5_STEPS - 1;
constexpr int S5_WAVES = 4;                 // waves per workgroup (two workgroups per CU: 8-wave workgroups whose second four waves start half a
                                            // step late -- one wave's pooled epilogue under its SIMD partner's MFMAs -- measured 58.8 against 55.2 us)
// LDS behind the converted weights: the four rings (or, first, the packed weights as they arrive: 5 NT KiB), then the bias
template <int NT> constexpr int S5_BIAS_OFF = (S5_WAVES * S5_RING * S5_PITCH > 5 * NT * 1024 ? S5_WAVES * S5_RING * S5_PITCH : 5 * NT * 1024);

struct S5Args {
  StemArgs s;
  int nsx, nseg, nitems;
  unsigned x_bytes, y_bytes;
};

template <int NT>
__global__ __launch_bounds__(S5_WAVES * 64, 2) void stem_k7(S5Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int WL = 7 * NT * 1024;         // [kernel row][cout tile][64 lanes][16 B]
  unsigned char* const wl = lds;
  unsigned char* const rings = lds + WL;    // (first: the packed weights as they arrive, 5 NT KiB)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;

  // ---- weights: gather order (K' = (row * 7 + col) * 3 + ch, 5 chunks) -> one chunk per kernel row (K'' = col * 4 + ch)
  {
    const i32x4* src = reinterpret_cast<const i32x4*>(a.s.wp);
    for (int i = threadIdx.x; i < 5 * NT * 64; i += S5_WAVES * 64) reinterpret_cast<i32x4*>(rings)[i] = src[i];
    __syncthreads();
    const unsigned short* tmp = reinterpret_cast<const unsigned short*>(rings);
    for (int u = threadIdx.x; u < 7 * NT * 64; u += S5_WAVES * 64) {
      const int L = u & 63, rt = u >> 6;
      const int r = rt / NT, t = rt - r * NT;
      const int gg = L >> 4, ii = L & 15;
      unsigned short h[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kk = gg * 8 + e, s = kk >> 2, ch = kk & 3;
        const int ci = (r * 7 + s) * 3 + ch;
        h[e] = (s < 7 && ch < 3) ? tmp[(((ci >> 5) * NT + t) * 64 + (((ci & 31) >> 3) * 16 + ii)) * 8 + (ci & 7)] : (unsigned short)0;
      }
      const i32x4 v = {(int)(h[0] | (unsigned)h[1] << 16), (int)(h[2] | (unsigned)h[3] << 16), (int)(h[4] | (unsigned)h[5] << 16),
                       (int)(h[6] | (unsigned)h[7] << 16)};
      reinterpret_cast<i32x4*>(wl)[u] = v;
    }
    __syncthreads();
  }

  unsigned char* const ring = rings + wave * (S5_RING * S5_PITCH);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.s.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.s.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const int cb = g * 4 * NT;                               // this lane's 4 NT consecutive couts
  // (the bias lives in LDS behind the rings: 16 NT floats, read in the epilogue -- 4 NT registers the step loop needs elsewhere)
  float* const biasl = reinterpret_cast<float*>(rings + S5_BIAS_OFF<NT>);
  if (threadIdx.x < 16 * NT) biasl[threadIdx.x] = a.s.bias[threadIdx.x];
  __syncthreads();
  const unsigned char* const wlane = wl + lane * 16;
  const unsigned char* const rlane = ring + 16 * j + 16 * g;
  const float NEG = -__builtin_inff();

  // a staging task = one 16-byte pixel pair of one ring row: task id = lane + 64 q -> (row id / 19, pair id % 19)
  int trow[4], tpair[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int id = lane + 64 * q;
    trow[q] = (int)(__umul24((unsigned)id, 3450u) >> 16);   // id / 19 for id < 256
    tpair[q] = id - trow[q] * S5_PAIRS;
  }

  const int wid = (int)blockIdx.x * S5_WAVES + wave, nw = (int)gridDim.x * S5_WAVES;
  // (geometry of an item: image, first pooled row / column, first input row / column)
  auto decode = [&](int item, int& n, int& py0, int& px0, int& iy0, int& ix0) {
    const int sx = item % a.nsx;
    const int it2 = item / a.nsx;
    const int seg = it2 % a.nseg;
    n = it2 / a.nseg;
    py0 = seg * S5_PROWS; px0 = sx * 7;
    iy0 = 4 * py0 - a.s.ptc; ix0 = 4 * px0 - a.s.plc;         // conv (2 py0, 2 px0) -> input (4 py0 - pad, ..)
  };
  // requests the pixel pairs of input rows [rho0, rho0 + nrows) of an item (relative to its iy0): 12 bytes each, zeros outside the image
  auto request = [&](int n, int iy0, int ix0, int rho0, int nrows, int ntask, unsigned (&v)[4][3]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= ntask) continue;
      const int iy = iy0 + rho0 + trow[q], ix = ix0 + 2 * tpair[q];
      const bool ok = trow[q] < nrows && (unsigned)iy < (unsigned)a.s.H && (unsigned)ix < (unsigned)a.s.W;
      const unsigned off = ok ? (unsigned)(((n * a.s.H + iy) * a.s.W + ix) * 6) : OOB;
      const auto d = __builtin_amdgcn_raw_buffer_load_b96(rx, off, 0, 0);
      v[q][0] = d[0]; v[q][1] = d[1]; v[q][2] = d[2];
    }
  };
  // writes them: two 3-channel pixels -> two 4-channel pixels (16 bytes) at ring row (rho0 + row) % 16
  auto commit = [&](int rho0, int nrows, int ntask, const unsigned (&v)[4][3]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q >= ntask) continue;
      if (trow[q] < nrows) {
        const unsigned d0 = v[q][0], d1 = v[q][1], d2 = v[q][2];
        const i32x4 o = {(int)d0, (int)(d1 & 0xffffu), (int)((d1 >> 16) | (d2 << 16)), (int)(d2 >> 16)};
        *reinterpret_cast<i32x4*>(ring + ((rho0 + trow[q]) & (S5_RING - 1)) * S5_PITCH + tpair[q] * 16) = o;
      }
    }
  };

  // Every request has a whole step of MFMAs to land: a step's rows are requested two steps ahead and written one step ahead, behind
  // the step's own fragment reads (the LDS serves a wave in order); the first 13 rows of the NEXT item are requested in the third step
  // of the current one (the registers are free by then) and written when the item starts.
  unsigned pre[4][3];
  {
    int n, py0, px0, iy0, ix0;
    if (wid < a.nitems) {
      decode(wid, n, py0, px0, iy0, ix0);
      request(n, iy0, ix0, 0, 13, 4, pre);
    }
  }
#pragma unroll 1
  for (int item = wid; item < a.nitems; item += nw) {
    int n, py0, px0, iy0, ix0;
    decode(item, n, py0, px0, iy0, ix0);
    commit(0, 13, 4, pre);
    request(n, iy0, ix0, 13, 8, 3, pre);

    f32x4 carry[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) carry[t] = f32x4{NEG, NEG, NEG, NEG};

#pragma unroll 1
    for (int step = 0; step < S5_STEPS; ++step) {
      // ---- the step's 13 input rows as B fragments (ring rows (8 step + rho) % 16)
      const int par = step & 1;
      const unsigned char* b0 = rlane + par * 8 * S5_PITCH;          // rho 0..7
      const unsigned char* b1 = rlane + (par ^ 1) * 8 * S5_PITCH;    // rho 8..12
      i32x4 bf[13];
#pragma unroll
      for (int rho = 0; rho < 8; ++rho) bf[rho] = *reinterpret_cast<const i32x4*>(b0 + rho * S5_PITCH);
#pragma unroll
      for (int rho = 8; rho < 13; ++rho) bf[rho] = *reinterpret_cast<const i32x4*>(b1 + (rho - 8) * S5_PITCH);
      // ---- 4 conv rows x NT cout tiles x 7 kernel rows
      f32x4 acc[4][NT];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const i32x4 af = *reinterpret_cast<const i32x4*>(wlane + (r * NT + t) * 1024);
#pragma unroll
          for (int m = 0; m < 4; ++m) mma16<f16>(acc[m][t], af, bf[2 * m + r]);
        }
      // ---- the next step's rows (requested a step ago) behind this step's fragment reads; then the rows of the step after that,
      //      or -- in the third step -- the first rows of this wave's next item
      if (step + 1 < S5_STEPS) {
        commit(8 * step + 13, 8, 3, pre);
        if (step + 2 < S5_STEPS) {
          request(n, iy0, ix0, 8 * step + 21, 8, 3, pre);
        } else if (item + nw < a.nitems) {
          int n2, py2, px2, iy2, ix2;
          decode(item + nw, n2, py2, px2, iy2, ix2);
          request(n2, iy2, ix2, 0, 13, 4, pre);
        }
      }
      // ---- pooled rows 2 step - 1 (carry + row 0) and 2 step (rows 0..2); rows 2, 3 are carried
      const int pyA = py0 + 2 * step - 1, pyB = pyA + 1;
      const int px = px0 + (j >> 1);
      const bool lane_ok = (j & 1) == 0 && j < 14 && px < a.s.Wp;
      const unsigned rowA = (step > 0 && pyA < a.s.Hp && lane_ok) ? (unsigned)((((n * a.s.Hp + pyA) * a.s.Wp + px) * a.s.y_cstride + a.s.y_coffset + cb) * 2) : OOB;
      const unsigned rowB = (pyB < a.s.Hp && lane_ok) ? (unsigned)((((n * a.s.Hp + pyB) * a.s.Wp + px) * a.s.y_cstride + a.s.y_coffset + cb) * 2) : OOB;
      auto hpool = [&](float v) {
        const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, NEG), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
        const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, NEG), __builtin_bit_cast(int, v), 0x102, 0xf, 0xf, false));
        return __builtin_fmaxf(v, __builtin_fmaxf(s1, s2));
      };
      auto finish = [&](const f32x4 (&v)[NT], unsigned row) {
        f16x8 h[(NT + 1) / 2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) h[t >> 1][(t & 1) * 4 + e] = (f16)__builtin_fmaxf(hpool(v[t][e]) + biasl[cb + t * 4 + e], 0.f);
#pragma unroll
        for (int p = 0; p < NT / 2; ++p)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, h[p]), ry, row != OOB ? row + (unsigned)(p * 16) : OOB, 0, 0);
      };
      f32x4 va[NT], vb[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          va[t][e] = __builtin_fmaxf(carry[t][e], acc[0][t][e]);
          vb[t][e] = __builtin_fmaxf(acc[0][t][e], __builtin_fmaxf(acc[1][t][e], acc[2][t][e]));
          carry[t][e] = __builtin_fmaxf(acc[2][t][e], acc[3][t][e]);
        }
      finish(va, rowA);
      finish(vb, rowB);
    }
  }
}

template <int NT>
int launch_s5(S5Args& a, hipStream_t st) {
  const size_t lds = (size_t)7 * NT * 1024 + S5_BIAS_OFF<NT> + 16 * NT * 4;
  auto kern = &stem_k7<NT>;
  static PerDevice once;
  SQDET_CHECK_HIP(once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
  int wgs = (a.nitems + S5_WAVES - 1) / S5_WAVES;
  const int cap = cu_count() * (S5_WAVES == 8 ? 1 : 2);
  if (wgs > cap) wgs = cap;
  hipLaunchKernelGGL(kern, dim3((unsigned)wgs), dim3(S5_WAVES * 64), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

}  // namespace

// float16, k = 7, VALID 3x3 / s2 pool, even image width and even left padding (12-byte pixel pairs start 4-byte aligned), 64 or 96
// couts in 16-byte aligned rows.  *handled = false: the strip kernel / conv + pool take it.  ("stem_algo" 2 / 3: never)
int stem_k7_launch(StemArgs s, int k, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (k != 7 || dtype != SQDET_F16 || !(s.Cout == 64 || s.Cout == 96)) return SQDET_OK;
  if (s.ptp != 0 || s.plp != 0 || s.Hp != (s.Hc - 3) / 2 + 1 || s.Wp != (s.Wc - 3) / 2 + 1) return SQDET_OK;     // VALID pool only
  if (s.W % 2 != 0 || s.plc % 2 != 0 || s.y_cstride % 8 != 0 || s.y_coffset % 8 != 0) return SQDET_OK;
  const size_t xb = (size_t)s.N * s.H * s.W * 6, yb = (size_t)s.N * s.Hp * s.Wp * s.y_cstride * 2;
  if (xb >= (1ull << 31) || yb >= (1ull << 31)) return SQDET_OK;
  S5Args a;
  a.s = s;
  a.nsx = (s.Wp + 6) / 7;
  a.nseg = (s.Hp + S5_PROWS - 1) / S5_PROWS;
  a.nitems = s.N * a.nsx * a.nseg;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
  const int rc = s.Cout == 96 ? launch_s5<6>(a, st) : launch_s5<4>(a, st);
  if (rc != SQDET_OK) return rc;
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
