// 1x1 convolution (+bias +ReLU) as a persistent streaming GEMM for gfx950: the squeeze1x1 and
// expand1x1 layers of the fire modules (reference src/nets/squeezeDet.py:95-102 through
// src/nn_skeleton.py:471-563).  These layers are HBM-bound (13-82 FLOP/B in fp16), so the kernel
// is organised around memory traffic, not MFMA rate:
//   * every wave owns ONE cout group (<= 6 tiles of 16 couts) and keeps that group's weights --
//     all NCH K-chunks x NT tiles of A fragments -- in REGISTERS for its whole life;
//   * it then grid-strides over pixel tiles (MT blocks of 16 NHWC pixels): B fragments are
//     16-byte loads straight from the activation tensor (a 1x1 conv needs no halo, no LDS), the
//     NEXT tile's loads are issued before the current tile's MFMAs (register double buffer);
//   * the epilogue stores 4*NT consecutive channels per lane as 16-byte vectors.
// Waves of different cout groups read the same pixels; K << Cout for the expand layers, so that
// re-read (served by L2) is small next to the output stream.
#include "conv_common.h"

namespace sqdet {

struct C1Args {
  ConvArgs c;
  int ntiles;      // pixel tiles of MT*16
  int nstreams;    // waves per cout group
};

// PERM (every cout group full: Cout % (16*NT) == 0): the wave gathers its A-fragment rows from the standard packing in ANOTHER
// order, so that MFMA row i of tile n is cout 32*(n>>1) + 8*(i>>2) + 4*(n&1) + (i&3) of the group (float16; a leftover odd
// tile and float32 keep 16-cout blocks: 16*n + 4*(i>>2) + (i&3)).  A lane group then holds the 8 consecutive couts of a tile
// PAIR: one 16-byte store per pixel, the four lane groups of a pixel 64 CONTIGUOUS bytes per store instruction -- with the
// standard order (a lane owns 4*NT consecutive couts) the four 16-byte pieces of an instruction lie 8*NT bytes apart, which
// streams at about half the rate (tools/microbench/store_patterns.hip; fire2/expand1x1 stand-alone: 39 us for 150 MB).
// Same products in the same order per cout: bitwise the same results.
template <typename T, int NCH, int NT, int MT, bool PERM>
__global__ __launch_bounds__(256) void conv1x1_stream(C1Args a) {
  constexpr int KG = Tr<T>::KG;
  constexpr int KC = 4 * KG;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);   // global wave id
  const int group = gw % a.c.ngroups;
  const int stream = gw / a.c.ngroups;
  if (stream >= a.nstreams) return;
  const int j = lane & 15, g = lane >> 4;

  // this wave's weights: [NCH][NT] fragments, resident in registers
  i32x4 af[NCH][NT];
  {
    const i32x4* wp = reinterpret_cast<const i32x4*>(a.c.wp) + (size_t)group * NCH * NT * 64 + lane;
    if constexpr (PERM) {
      const i32x4* wg = reinterpret_cast<const i32x4*>(a.c.wp) + (size_t)group * NCH * NT * 64;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool pair = sizeof(T) == 2 && (t | 1) < NT;
        const int cd = pair ? 32 * (t >> 1) + 8 * (j >> 2) + 4 * (t & 1) + (j & 3) : 16 * t + 4 * (j >> 2) + (j & 3);   // the cout this row computes
        const int tn = (cd % (4 * NT)) >> 2, ti = 4 * (cd / (4 * NT)) + (cd & 3);                                         // where the packing keeps it
#pragma unroll
        for (int c = 0; c < NCH; ++c) af[c][t] = wg[(c * NT + tn) * 64 + 16 * g + ti];
      }
    } else {
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) af[c][t] = wp[(c * NT + t) * 64];
    }
  }
  const int cb = group * 16 * NT + (PERM ? 0 : g * 4 * NT);
  // channel offset (within the group) of this lane's 4 couts of tile t
  auto coff = [&](int t) {
    if constexpr (!PERM) return t * 4;
    else return (sizeof(T) == 2 && (t | 1) < NT) ? 32 * (t >> 1) + 8 * g + 4 * (t & 1) : 16 * t + 4 * g;
  };
  f32x4 bias[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    bias[t] = cb + coff(t) < a.c.Cout ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + coff(t)) : f32x4{0.f, 0.f, 0.f, 0.f};
  // couts this lane may store: Cout is a multiple of 4, whole tiles beyond Cout are skipped
  int nt_valid = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) nt_valid += cb + t * 4 < a.c.Cout ? 1 : 0;

  const T* x = reinterpret_cast<const T*>(a.c.x);
  T* y = reinterpret_cast<T*>(a.c.y);
  const i32x4 zero = {0, 0, 0, 0};
  bool k_ok[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) k_ok[c] = c * KC + g * KG < a.c.Cin;

  auto load_tile = [&](int tile, i32x4 (&bf)[MT][NCH]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int p = (tile * MT + m) * 16 + j;
      const bool ok = p < a.c.P;
      const T* src = x + (size_t)(ok ? p : 0) * a.c.Cin + g * KG;
#pragma unroll
      for (int c = 0; c < NCH; ++c) bf[m][c] = (ok && k_ok[c]) ? *reinterpret_cast<const i32x4*>(src + c * KC) : zero;
    }
  };

  i32x4 bcur[MT][NCH], bnext[MT][NCH];
  int tile = stream;
  if (tile < a.ntiles) load_tile(tile, bcur);
  for (; tile < a.ntiles; tile += a.nstreams) {
    const int nxt = tile + a.nstreams;
    if (nxt < a.ntiles) load_tile(nxt, bnext);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) mma16<T>(acc[m][t], af[c][t], bcur[m][c]);

#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int p = (tile * MT + m) * 16 + j;
      if (p < a.c.P) {
        f32x4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          v[t] = acc[m][t] + bias[t];
          if (a.c.relu) {
            v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
            v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
          }
        }
        T* dst = y + (size_t)p * a.c.y_cstride + a.c.y_coffset + cb;
        if constexpr (PERM) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (sizeof(T) == 2 && (t | 1) < NT) {
              if ((t & 1) == 0) {
                const f16x8 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3],
                                 (f16)v[t + 1 < NT ? t + 1 : t][0], (f16)v[t + 1 < NT ? t + 1 : t][1], (f16)v[t + 1 < NT ? t + 1 : t][2], (f16)v[t + 1 < NT ? t + 1 : t][3]};
                *reinterpret_cast<f16x8*>(dst + coff(t)) = h;
              }
            } else {
              store4<T>(dst + coff(t), v[t]);
            }
          }
        } else {
          store_couts<T, NT>(dst, v, nt_valid);
        }
      }
    }
    if (nxt < a.ntiles) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NCH; ++c) bcur[m][c] = bnext[m][c];
    }
  }
}

template <typename T, int NCH, int NT, int MT>
static void launch_c1(C1Args& a, hipStream_t st) {
  a.ntiles = (a.c.P + 16 * MT - 1) / (16 * MT);
  // persistent: ~8 waves per SIMD-slot budget -> 256 CUs x 16 waves, split over the cout groups
  const int budget = tune(TUNE_C1_WAVES) > 0 ? tune(TUNE_C1_WAVES) : cu_count() * 16;
  int streams = budget / a.c.ngroups;
  if (streams < 1) streams = 1;
  // a wave re-loads its weight fragments once: give it at least `min_tiles` pixel tiles to amortise them
  const int min_tiles = tune(TUNE_C1_MIN_TILES) > 0 ? tune(TUNE_C1_MIN_TILES) : 1;
  if (streams * min_tiles > a.ntiles) streams = (a.ntiles + min_tiles - 1) / min_tiles;
  if (streams > a.ntiles) streams = a.ntiles;
  if (streams < 1) streams = 1;
  a.nstreams = streams;
  const int waves = streams * a.c.ngroups;
  // PERM: whole cout groups and 16-byte aligned rows (float16 pairs store 16 bytes at 16-byte channel offsets)
  const bool perm = a.c.Cout % (16 * NT) == 0 && a.c.y_cstride % 8 == 0 && a.c.y_coffset % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.c.y) & 15) == 0 && tune(TUNE_DBG) != 50;
  if (perm) hipLaunchKernelGGL((conv1x1_stream<T, NCH, NT, MT, true>), dim3((waves + 3) / 4), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((conv1x1_stream<T, NCH, NT, MT, false>), dim3((waves + 3) / 4), dim3(256), 0, st, a);
}

template <typename T, int NCH, int NT>
static void dispatch_c1_mt(C1Args& a, int mt, hipStream_t st) {
  if (mt == 4) launch_c1<T, NCH, NT, 4>(a, st);
  else launch_c1<T, NCH, NT, 2>(a, st);
}

template <typename T, int NCH>
static bool dispatch_c1_nt(C1Args& a, int nt, int mt, hipStream_t st) {
  switch (nt) {
    case 1: dispatch_c1_mt<T, NCH, 1>(a, mt, st); return true;
    case 2: dispatch_c1_mt<T, NCH, 2>(a, mt, st); return true;
    case 3: dispatch_c1_mt<T, NCH, 3>(a, mt, st); return true;
    case 4: dispatch_c1_mt<T, NCH, 4>(a, mt, st); return true;
    case 5: dispatch_c1_mt<T, NCH, 5>(a, mt, st); return true;
    case 6: dispatch_c1_mt<T, NCH, 6>(a, mt, st); return true;
    default: return false;
  }
}

template <typename T>
static bool dispatch_c1(C1Args& a, int nch, int nt, int mt, hipStream_t st) {
  switch (nch) {
    case 1: return dispatch_c1_nt<T, 1>(a, nt, mt, st);
    case 2: return dispatch_c1_nt<T, 2>(a, nt, mt, st);
    case 3: return dispatch_c1_nt<T, 3>(a, nt, mt, st);
    case 4: return dispatch_c1_nt<T, 4>(a, nt, mt, st);
    default: return false;
  }
}

// *handled = false: not eligible (the generic kernel runs instead).
int conv1x1_stream_launch(const ConvArgs& c, const ConvGeom& g, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (c.k != 1 || c.stride != 1 || g.gather) return SQDET_OK;
  if (g.nchunk > 4 || g.nchunk * g.nt > 18) return SQDET_OK;   // weights must fit in registers
  C1Args a;
  a.c = c;
  // register budget: accumulators MT*NT*4 + double-buffered B 2*MT*NCH*4 + A NCH*NT*4
  int mt = 4;
  if (4 * g.nt * 4 + 2 * 4 * g.nchunk * 4 + g.nchunk * g.nt * 4 > 200) mt = 2;
  if (c.P < 16 * 4 * 1024) mt = 2;
  if (tune(TUNE_C1_MT) == 2 || tune(TUNE_C1_MT) == 4) mt = tune(TUNE_C1_MT);
  const bool ok = dtype == SQDET_F16 ? dispatch_c1<f16>(a, g.nchunk, g.nt, mt, st)
                                     : dispatch_c1<float>(a, g.nchunk, g.nt, mt, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
