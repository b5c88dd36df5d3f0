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
#include <type_traits>

#include "conv_common.h"

namespace sqdet {

struct C1Args {
  ConvArgs c;
  int ntiles;      // pixel tiles of MT*16
  int nstreams;    // waves per cout group
  unsigned x_bytes, y_bytes;   // extents of the two tensors (buffer resources: offsets beyond them read zeros / store nothing)
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
  // LIN (float16, NT = 4, PERM: the expand1x1 convs of fire2 .. fire5 -- 64 couts = 128 bytes of a pixel's row per wave): the rounded
  // tile goes through 2 KiB of LDS of the wave's own and leaves as LINEAR stores -- lane l writes bytes [16 (l & 7), +16) of pixel
  // l >> 3, a store instruction covers 8 pixels x 128 contiguous bytes (1 KiB in one piece when the conv has 64 couts) -- instead of
  // the MFMA D layout's 16 pixels x 64 bytes with ADJACENT LANES 128 BYTES APART.  Same bytes; with cold reads beside them the D-layout
  // stores of this stream run at 3.9 TB/s, the linear ones at 4.8 (tools/microbench/store_shape.hip: 38.4 against 31.4 us for fire2's
  // 150 MB).  No barrier: the wave reads back what it wrote itself (LDS executes a wave's instructions in order).
  constexpr bool LIN = PERM && sizeof(T) == 2 && NT == 4;
  // LIN6 (round 6; float16, NT = 6, PERM: the 192- / 384-cout expand1x1 convs of fire6 / fire7 / fire10 / fire11 -- 96 couts = 192 bytes of
  // a pixel's row per wave): the same bounce through 3 KiB of the wave's own LDS; a store instruction then covers 1 KiB of the tile in
  // pixel-major order -- runs of 192 contiguous bytes (5.3 pixels) instead of 16 x 64 bytes with adjacent lanes 128 bytes apart.
  // 16-byte piece q of pixel P sits in slot (q + P) mod 12 of its 192-byte row: the eight lanes a cycle serves hit eight different
  // bank quads on the way in.
  constexpr bool LIN6 = PERM && sizeof(T) == 2 && NT == 6;
  __shared__ __attribute__((aligned(16))) unsigned char lin_lds[LIN ? 4 * 2048 : (LIN6 ? 4 * 3072 : 16)];
  unsigned char* const lin = lin_lds + (LIN ? (threadIdx.x >> 6) * 2048 : (LIN6 ? (threadIdx.x >> 6) * 3072 : 0));
  // LIN6: store k of a block moves 16-byte unit u = 64 k + lane of the tile: pixel u / 12, piece u % 12
  [[maybe_unused]] int l6_pix[3], l6_lds[3];
  [[maybe_unused]] unsigned l6_rel[3];
  if constexpr (LIN6) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int u = 64 * k + lane;
      const int P = (int)(__umul24((unsigned)u, 5462u) >> 16);       // u / 12 for u < 192
      const int q = u - 12 * P;
      const int sl = q + P >= 24 ? q + P - 24 : (q + P >= 12 ? q + P - 12 : q + P);
      l6_pix[k] = P;
      l6_lds[k] = P * 192 + sl * 16;
      l6_rel[k] = (unsigned)q * 16u;
    }
  }

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

  // Everything inside the tile loop is branch-free: loads and stores are raw buffer operations whose offset is out of range (zeros /
  // dropped) for pixels past the end, for K groups past Cin and for the prefetch past the last tile.  With `if (p < P)` blocks around
  // them the loop was 39 basic blocks and hipcc's wait-count pass gave up at the joins: `s_waitcnt vmcnt(0)` at the top of every
  // step, i.e. every wave waited for its own STORES to be acknowledged before asking for the next pixels, and a launch took
  // stream time + compute time (fire2/expand1x1: 38-42 us where the same loads and stores without the arithmetic take 25,
  // tools/microbench/store_shape.hip).  One block: the pass counts (`vmcnt(stores + loads issued behind the fragment)`).
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.c.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.c.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  constexpr unsigned ES = sizeof(T);
  const unsigned xrow = (unsigned)a.c.Cin * ES, yrow = (unsigned)a.c.y_cstride * ES;
  unsigned kb[NCH];        // byte offset of this lane's K group inside a pixel row; OOB: the group lies past Cin (zeros)
#pragma unroll
  for (int c = 0; c < NCH; ++c) kb[c] = c * KC + g * KG < a.c.Cin ? (unsigned)(c * KC + g * KG) * ES : OOB;
  const unsigned yb = (unsigned)(a.c.y_coffset + cb) * ES;
  const unsigned ybl = (unsigned)(a.c.y_coffset + group * 16 * NT) * ES;   // (LIN: the group's segment of a row, no lane term)

  auto load_tile = [&](int tile, i32x4 (&bf)[MT][NCH]) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int p = (tile * MT + m) * 16 + j;
      const unsigned pb = (unsigned)p * xrow;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        bf[m][c] = __builtin_amdgcn_raw_buffer_load_b128(rx, (p < a.c.P && kb[c] != OOB) ? pb + kb[c] : OOB, 0, 0);
    }
  };

  auto run = [&](auto relu_t) {
    constexpr bool RELU = decltype(relu_t)::value;
    // one step: the NEXT tile's fragments are requested into `nxt`, then the tile in `cur` is computed and stored.  The loop runs two
    // steps per trip over two register sets (no copies: a copy placed behind the last MFMA of a fragment waits for the load that was
    // only just issued -- the prefetch would be a load that is waited for at once).
    auto step = [&](int tile, i32x4 (&cur)[MT][NCH], i32x4 (&nxt)[MT][NCH]) {
      load_tile(tile + a.nstreams, nxt);             // (past the last tile: every offset out of range, nothing is fetched)
      // (the fence keeps the scheduler from sinking these loads below the epilogue, into the registers the accumulators free)
      __builtin_amdgcn_sched_barrier(0);

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
          for (int t = 0; t < NT; ++t) mma16<T>(acc[m][t], af[c][t], cur[m][c]);

#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int p = (tile * MT + m) * 16 + j;
        const unsigned po = p < a.c.P ? (unsigned)p * yrow + yb : OOB;
        f32x4 v[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          v[t] = acc[m][t] + bias[t];
          if constexpr (RELU) {
            v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
            v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
          }
        }
        if constexpr (LIN) {
          // pair p = couts [32 p + 8 g, +8) of pixel j = piece 4 p + g of its 128-byte segment; slot = piece ^ (pixel & 7): the eight lanes
          // a cycle serves write / read eight different 16-byte slots of 128-byte rows -- conflict-free both ways
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {
            const f16x8 h = {(f16)v[2 * pr][0], (f16)v[2 * pr][1], (f16)v[2 * pr][2], (f16)v[2 * pr][3],
                             (f16)v[2 * pr + 1][0], (f16)v[2 * pr + 1][1], (f16)v[2 * pr + 1][2], (f16)v[2 * pr + 1][3]};
            *reinterpret_cast<f16x8*>(lin + j * 128 + (((4 * pr + g) ^ (j & 7)) << 4)) = h;
          }
          asm volatile("" ::: "memory");               // (the reads below are not hoisted above the writes)
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            const int P = sx * 8 + (lane >> 3), q = lane & 7;
            const i32x4 o = *reinterpret_cast<const i32x4*>(lin + P * 128 + ((q ^ (P & 7)) << 4));
            const int pp = (tile * MT + m) * 16 + P;
            __builtin_amdgcn_raw_buffer_store_b128(o, ry, pp < a.c.P ? (unsigned)pp * yrow + ybl + (unsigned)q * 16u : OOB, 0, 0);
          }
          asm volatile("" ::: "memory");               // (nor the next block's writes above these reads)
        } else if constexpr (LIN6) {
#pragma unroll
          for (int pr = 0; pr < 3; ++pr) {
            const f16x8 h = {(f16)v[2 * pr][0], (f16)v[2 * pr][1], (f16)v[2 * pr][2], (f16)v[2 * pr][3],
                             (f16)v[2 * pr + 1][0], (f16)v[2 * pr + 1][1], (f16)v[2 * pr + 1][2], (f16)v[2 * pr + 1][3]};
            const int sl = 4 * pr + g + j;             // slot (piece + pixel) mod 12, piece = 4 pr + g, pixel = j (< 27: at most two wraps)
            *reinterpret_cast<f16x8*>(lin + j * 192 + ((sl >= 24 ? sl - 24 : (sl >= 12 ? sl - 12 : sl)) << 4)) = h;
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const i32x4 o = *reinterpret_cast<const i32x4*>(lin + l6_lds[k]);
            const int pp = (tile * MT + m) * 16 + l6_pix[k];
            __builtin_amdgcn_raw_buffer_store_b128(o, ry, pp < a.c.P ? (unsigned)pp * yrow + ybl + l6_rel[k] : OOB, 0, 0);
          }
          asm volatile("" ::: "memory");
        } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const bool pair = PERM && sizeof(T) == 2 && (t | 1) < NT;
          if (pair && (t & 1)) continue;               // (stored with its even partner)
          // non-PERM: whole 4-cout pieces beyond Cout do not exist
          const unsigned off = (po != OOB && (PERM || t < nt_valid)) ? po + (unsigned)coff(t) * ES : OOB;
          if constexpr (sizeof(T) == 2) {
            if (pair) {
              const int t1 = t + 1 < NT ? t + 1 : t;
              const f16x8 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3],
                               (f16)v[t1][0], (f16)v[t1][1], (f16)v[t1][2], (f16)v[t1][3]};
              __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, h), ry, off, 0, 0);
            } else {
              const f16x4 h = {(f16)v[t][0], (f16)v[t][1], (f16)v[t][2], (f16)v[t][3]};
              __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, h), ry, off, 0, 0);
            }
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v[t]), ry, off, 0, 0);
          }
        }
        }
      }
    };
    i32x4 b0[MT][NCH], b1[MT][NCH];
    int tile = stream;
    load_tile(tile, b0);
    // One step's worth of stores that go nowhere (every offset out of range), so that the memory queue looks the same on both ways
    // into the loop -- [fragments][stores] -- : hipcc's wait counts at a loop header are the minimum over the incoming paths, and
    // coming from here with the fragments alone it made every first step wait until all but two of the PREVIOUS step's stores were
    // acknowledged (vmcnt(6) where the back edge needs vmcnt(15)).
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bool pair = PERM && sizeof(T) == 2 && (t | 1) < NT;
        if (pair && (t & 1)) continue;
        if (sizeof(T) == 2 && !pair) __builtin_amdgcn_raw_buffer_store_b64(i32x2{0, 0}, ry, OOB, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(i32x4{0, 0, 0, 0}, ry, OOB, 0, 0);
      }
    // (a trip's second step past the last tile computes on zeros and stores nothing)
    for (; tile < a.ntiles; tile += 2 * a.nstreams) {
      step(tile, b0, b1);
      step(tile + a.nstreams, b1, b0);
    }
  };
  if (a.c.relu) run(std::true_type{});
  else run(std::false_type{});
}

// resident 256-thread workgroups per CU of one instantiation (registers decide), asked once per device
template <typename T, int NCH, int NT, int MT, bool PERM>
static int c1_blocks_per_cu() {
  static std::atomic<int> cached[64];      // (relaxed: racing first launches of two host threads both ask and store the same number)
  const int d = current_device() & 63;
  int v = cached[d].load(std::memory_order_relaxed);
  if (v == 0) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(&conv1x1_stream<T, NCH, NT, MT, PERM>), 256, 0) != hipSuccess || n < 1) {
      (void)hipGetLastError();
      n = 4;
    }
    v = n > 8 ? 8 : n;
    cached[d].store(v, std::memory_order_relaxed);
  }
  return v;
}

template <typename T, int NCH, int NT, int MT>
static void launch_c1(C1Args& a, hipStream_t st) {
  a.ntiles = (a.c.P + 16 * MT - 1) / (16 * MT);
  // PERM: whole cout groups and 16-byte aligned rows (float16 pairs store 16 bytes at 16-byte channel offsets)
  const bool perm = a.c.Cout % (16 * NT) == 0 && a.c.y_cstride % 8 == 0 && a.c.y_coffset % 8 == 0 &&
                    (reinterpret_cast<uintptr_t>(a.c.y) & 15) == 0 && tune(TUNE_DBG) != 50;
  // persistent: exactly the waves that are RESIDENT together (a wave that starts when another one ends repeats the ramp: with the
  // fixed 16 waves per CU of rounds 1-4 a 159-register instantiation ran 1.33 rounds), split over the cout groups
  const int per_cu = perm ? c1_blocks_per_cu<T, NCH, NT, MT, true>() : c1_blocks_per_cu<T, NCH, NT, MT, false>();
  const int budget = tune(TUNE_C1_WAVES) > 0 ? tune(TUNE_C1_WAVES) : cu_count() * per_cu * 4;
  int streams = budget / a.c.ngroups;
  if (streams < 1) streams = 1;
  // a wave re-loads its weight fragments once: give it at least `min_tiles` pixel tiles to amortise them
  const int min_tiles = tune(TUNE_C1_MIN_TILES) > 0 ? tune(TUNE_C1_MIN_TILES) : 1;
  if (streams * min_tiles > a.ntiles) streams = (a.ntiles + min_tiles - 1) / min_tiles;
  if (streams > a.ntiles) streams = a.ntiles;
  if (streams < 1) streams = 1;
  a.nstreams = streams;
  const int waves = streams * a.c.ngroups;
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
  const size_t es = dtype == SQDET_F16 ? 2 : 4;
  const size_t xb = (size_t)c.P * c.Cin * es, yb = (size_t)c.P * c.y_cstride * es;
  if (xb >= (1ull << 31) || yb >= (1ull << 31)) return SQDET_OK;   // 32-bit buffer offsets (the generic kernel takes over)
  C1Args a;
  a.c = c;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
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
