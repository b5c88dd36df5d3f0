// The expand half of a fire module from its SQUEEZE tensor (+ the 3x3 / stride-2 SAME max-pool behind it) + the NEXT module's
// squeeze1x1, in one launch -- SqueezeDet's fire2/expand+fire3/squeeze, fire3/expand+pool3+fire4/squeeze, fire4/expand+fire5/squeeze,
// fire5/expand+pool5+fire6/squeeze (reference src/nets/squeezeDet.py:46-57, 81-106).  Successor of fire_stream's SQIN / NTS2 forms
// (fire2.hip), built for FOUR waves per SIMD instead of two: those launches moved 21-60 MB, ran the matrix pipe 25-29 % of the time and
// issued one instruction per wave every ~11 cycles -- latency-bound with nothing to hide it behind (profiles/r03_sq_counters.txt).
//
//   * the squeeze halo tile comes in by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no ds_write pass, and an
//     out-of-range offset (pixels outside the image, the pitch columns, a tile past the band) lands as ZEROS in LDS -- exactly the SAME
//     padding of the squeeze tensor (tools/microbench/dma_oob_probe.hip).  The tile of step t+1 is requested at the top of step t into the
//     other buffer and waited for with a counted `s_waitcnt vmcnt(stores issued behind it)` at the end of step t.
//   * S = 16 (float16): 32 bytes per pixel in LDS, unswizzled -- 16 consecutive pixels x pieces {0,1} are conflict-free as they lie
//     (ds_read_b128 serves lanes in groups of 8 + 8 across two lane groups) -- with a 19-pixel pitch whose 19th column is never
//     fetched, i.e. always zero: the K slots that have no tap (second half of the fifth tap PAIR, the expand1x1's upper half) read it.
//     S = 32: 64 bytes per pixel, 24-pixel pitch, piece slot ^ ((pixel >> 1) & 3) as in conv3x3_tile.h; the swizzle is applied on the
//     SOURCE side of the DMA (lane l fetches the piece whose slot it fills).
//   * ROW-OUTER accumulation: a wave walks its tile rows one at a time (all K-steps of a row, then its epilogue), so the live state is
//     one row's accumulators + B fragments next to the register-resident expand weights -- 100-125 registers where the K-outer form held
//     150-210 -- and the epilogue VALU work of row m overlaps the MFMAs of row m + 1 of the other waves of the SIMD.
//   * NTW = 1 forms (S = 32): a wave owns ONE 16-cout tile (36 + 4 weight registers instead of 72 + 8) and all rows of the tile.
// Accumulation order per output = fire_stream's (K-steps ascending; the S = 16 forms pair two taps per MFMA in the same K slots; the
// next squeeze walks the concat chunks in ascending order): results are bitwise those of fire2.hip's kernels.
#include <type_traits>

#include "conv_common.h"

namespace sqdet {
namespace {

struct FireXArgs {
  const void* sq_in;
  void* s_out;
  const void *w1, *w3, *ws2;
  const float *b1, *b3, *bs2;
  int N, H, W, S, E, S2;
  int tiles_x, tiles_y, ntiles;
  unsigned in_bytes, out_bytes;
  int Hp, Wp, ptp, plp;        // POOL: pooled dims and the SAME pads (top / left) of the 3x3/s2 pool
};

constexpr int XCOLS = 16;      // module-output columns per tile

template <bool S16, bool POOL, int ROWS_T> struct GeoX {
  static constexpr int ROWS = POOL ? 9 : ROWS_T;          // module rows per tile (POOL: 9 -> 4 pooled rows)
  static constexpr int HR = ROWS + 2;                     // halo rows
  static constexpr int PXB = S16 ? 32 : 64;               // bytes per pixel in the LDS tile
  static constexpr int LW = S16 ? 19 : 24;                // row pitch in pixels (columns >= 18 are never fetched: zeros)
  static constexpr int NB = (HR * LW * PXB + 1023) / 1024;   // 1-KiB DMA blocks per tile
  static constexpr int STILE = NB * 1024;
  static constexpr int RSTEP = POOL ? 8 : ROWS_T, CSTEP = POOL ? 14 : 16;
};

// One 1-KiB block straight into LDS: global address = buffer resource + per-lane byte offset (out of range -> zeros), LDS address =
// M0 + lane * 16.  Hidden from hipcc's wait-count pass on purpose (it would wait vmcnt(0) at the next LDS read).
__device__ __forceinline__ void dma16(unsigned voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void vm_wait_x() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned int pkmax(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// S16: squeeze depth 16 (tap-paired expand3x3), else 32.  NG: 64-cout groups per expand conv (E = 64 NG).  NTW: cout tiles per wave.
// RS: row split.  ROWS_T: tile rows of the unpooled forms (8 or 4).  NTS2: cout tiles of the next squeeze.  WPS: waves per SIMD the
// launch is sized for (register budget 512 / WPS).
template <bool S16, bool POOL, int NG, int NTW, int RS, int ROWS_T, int NTS2, int WPS>
__global__ __launch_bounds__((4 * NG / NTW) * RS * 64, WPS) void fire_dma(FireXArgs a) {
  using G = GeoX<S16, POOL, ROWS_T>;
  constexpr int NWAVES = (4 * NG / NTW) * RS;
  constexpr int NT3 = S16 ? 5 : 9;
  constexpr int LW = G::LW, PXB = G::PXB, ROWP = LW * PXB, NB = G::NB, STILE = G::STILE;
  constexpr int NBW = (NB + NWAVES - 1) / NWAVES;     // DMA blocks per wave
  constexpr int NQ = POOL ? 4 / RS : 0;               // pooled rows per wave
  constexpr int MT = POOL ? 2 * NQ + 1 : G::ROWS / RS;   // module rows per wave
  constexpr int NQC = NG * 4;                         // 64-byte K chunks of the concat tensor
  constexpr int CPIX = POOL ? 32 : G::ROWS * 16;      // pixels of the concat tile
  constexpr int NBLK = CPIX / 16;
  constexpr int NIT = NBLK * NTS2, IPW = (NIT + NWAVES - 1) / NWAVES;   // next-squeeze work items (16 pixels x 16 couts)
  static_assert(!POOL || RS == 1 || RS == 2, "pooled forms: 4 or 2 pooled rows per wave");
  static_assert(IPW <= 2, "wait counts cover at most two stores per wave and step");
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* sq = lds;                                      // [2][STILE]
  unsigned char* ctile = sq + 2 * STILE;                        // [NQC][CPIX][4 x 16 B swizzled]
  unsigned char* ws2l = ctile + NQC * CPIX * 64;                // [NQC][NTS2][64 lanes][16 B]
  float* bl = reinterpret_cast<float*>(ws2l + NQC * NTS2 * 1024);   // [b1 E | b3 E | bs2 S2]
  const unsigned sq_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)sq;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;

  // tiles of this workgroup: XCD x (= blockIdx % 8) owns the contiguous band [x * per, (x + 1) * per)
  const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3, nl = gridDim.x >> 3;
  const int per = (a.ntiles + 7) >> 3;
  const int band_end = min(a.ntiles, (xcd + 1) * per);
  int tile = xcd * per + lid;
  if (tile >= band_end) return;                       // (whole workgroup: no barrier has been reached)

  struct TileXY { int tx, ty, n; };
  auto decode = [&](int t) {
    TileXY c;
    c.tx = t % a.tiles_x; t /= a.tiles_x;
    c.ty = t % a.tiles_y;
    c.n = t / a.tiles_y;
    return c;
  };
  const TileXY dstride = decode(nl);
  auto advance = [&](TileXY& c) {
    c.tx += dstride.tx;
    const int cy = c.tx >= a.tiles_x ? 1 : 0;
    c.tx -= cy ? a.tiles_x : 0;
    c.ty += dstride.ty + cy;
    const int cn = c.ty >= a.tiles_y ? 1 : 0;
    c.ty -= cn ? a.tiles_y : 0;
    c.n += dstride.n + cn;
  };
  TileXY cur = decode(tile), nxt = cur;
  advance(nxt);

  // ---- DMA: per-lane, tile-invariant parts ----
  // buffer resource of the squeeze tensor as four SGPRs (raw buffer, stride 0: offsets >= in_bytes are out of range)
  const unsigned long long xaddr = (unsigned long long)(uintptr_t)a.sq_in;
  const i32x4 rin = {(int)(unsigned)xaddr, (int)(unsigned)((xaddr >> 32) & 0xffffu), (int)a.in_bytes, 0x00020000};
  constexpr unsigned OOBL = 0x80000000u;              // (+ any tensor offset stays out of range: tensors are < 2 GiB)
  unsigned relx[NBW], rcp[NBW];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int b = wave + NWAVES * i;
    int PL, piece;
    if constexpr (S16) { PL = b * 32 + (lane >> 1); piece = lane & 1; }
    else { PL = b * 16 + (lane >> 2); piece = (lane & 3) ^ ((PL >> 1) & 3); }
    const int r = PL / LW, c = PL - r * LW;
    const bool exists = b < NB && r < G::HR && c < XCOLS + 2;
    relx[i] = exists ? (unsigned)(((r * a.W + c) * a.S + piece * 8) * 2) : OOBL;
    rcp[i] = ((unsigned)r << 16) | (unsigned)c;
  }
  auto issue_dma = [&](const TileXY& tc, int buf) {
    const int hy0 = tc.ty * G::RSTEP - (POOL ? a.ptp : 0) - 1, hx0 = tc.tx * G::CSTEP - (POOL ? a.plp : 0) - 1;
    const bool allin = hy0 >= 0 && hy0 + G::HR <= a.H && hx0 >= 0 && hx0 + XCOLS + 2 <= a.W;   // wave-uniform
    const unsigned base = (unsigned)((((tc.n * a.H + hy0) * a.W + hx0) * a.S) * 2);   // (modular when hy0 / hx0 = -1)
#pragma unroll
    for (int i = 0; i < NBW; ++i) {
      const int b = wave + NWAVES * i;
      if (b < NB) {                                   // wave-uniform
        unsigned off = base + relx[i];
        if (!allin) {
          const int iy = hy0 + (int)(rcp[i] >> 16), ix = hx0 + (int)(rcp[i] & 0xffffu);
          if (!(iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)) off = OOBL;
        }
        dma16(off, rin, sq_addr + (unsigned)(buf * STILE + b * 1024));
      }
    }
  };
  issue_dma(cur, 0);

  // ---- one-time set-up: next squeeze's weights and the biases -> LDS; this wave's expand weights -> registers ----
  {
    const i32x4* src2 = reinterpret_cast<const i32x4*>(a.ws2);
    for (int i = threadIdx.x; i < NQC * NTS2 * 64; i += NWAVES * 64) reinterpret_cast<i32x4*>(ws2l)[i] = src2[i];
    for (int i = threadIdx.x; i < 2 * a.E + a.S2; i += NWAVES * 64)
      bl[i] = i < a.E ? a.b1[i] : (i < 2 * a.E ? a.b3[i - a.E] : a.bs2[i - 2 * a.E]);
  }
  const int cw = wave / RS, rq = wave % RS;           // cout wave, row part
  const int cp = NTW == 2 ? cw : cw >> 1;             // cout pair: couts [cp * 32, cp * 32 + 32) of both expand convs
  const int t0 = NTW == 2 ? 0 : (cw & 1);             // NTW = 1: which tile of the pair
  const int m0 = POOL ? rq * 2 * NQ : rq * MT;        // first module row of this wave
  // tile t of the pair: tile row i <-> cout cp*32 + 8*(i>>2) + 4*t + (i&3), so lane group g ends up with the consecutive couts
  // cp*32 + 8g + 4t + [0,4) (fire2.hip has the derivation); the packed weights hold cout c of a 64-cout group in tile (c%16)/4,
  // row 4*(c/16) + c%4
  int wsrc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int c = (cp & 1) * 32 + 8 * (j >> 2) + 4 * (t0 + t) + (j & 3);
    wsrc[t] = ((c & 15) >> 2) * 64 + (4 * (c >> 4) + (c & 3)) + 16 * g;
  }
  const int group = cp >> 1;
  i32x4 w3r[NT3][NTW], w1r[NTW];
  {
    const i32x4* p3 = reinterpret_cast<const i32x4*>(a.w3) + (size_t)group * 9 * 4 * 64;
#pragma unroll
    for (int p = 0; p < NT3; ++p) {
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        if constexpr (S16) {
          // k-group g of the paired fragment = k-group (g & 1) of tap 2p + (g >> 1); tap 9 does not exist: zeros
          const int tap = 2 * p + (g >> 1);
          w3r[p][t] = tap < 9 ? p3[tap * 4 * 64 + wsrc[t] - 16 * g + 16 * (g & 1)] : i32x4{0, 0, 0, 0};
        } else {
          w3r[p][t] = p3[p * 4 * 64 + wsrc[t]];
        }
      }
    }
    const i32x4* p1 = reinterpret_cast<const i32x4*>(a.w1) + (size_t)group * 4 * 64;
#pragma unroll
    for (int t = 0; t < NTW; ++t) w1r[t] = p1[wsrc[t]];
  }
  const int cb = cp * 32 + g * 8 + 4 * t0;            // this lane's first cout (of both expand convs): 4 * NTW consecutive ones

  // LDS byte offsets (inside a squeeze tile, at this wave's first row) of this lane's B fragments
  //   S16: one per tap pair (lane groups 0,1: pieces 0,1 of tap 2p's pixel; groups 2,3: of tap 2p+1's; no such tap: the zero column)
  //        + the expand1x1's (groups 0,1: the centre pixel; groups 2,3: the zero column)
  //   S32: one per dx (the swizzle term is row-independent with a 24-pixel pitch); dy and the row are immediates
  constexpr int NOFF = S16 ? 6 : 3;
  unsigned boff[NOFF];
  if constexpr (S16) {
#pragma unroll
    for (int p = 0; p < 5; ++p) {
      const int tap = 2 * p + (g >> 1);
      const int dy = tap / 3, dx = tap - dy * 3;
      const int pix = tap < 9 ? (m0 + dy) * LW + j + dx : m0 * LW + XCOLS + 2;
      boff[p] = (unsigned)(pix * PXB + (g & 1) * 16);
    }
    const int pix1 = g < 2 ? (m0 + 1) * LW + j + 1 : m0 * LW + XCOLS + 2;
    boff[5] = (unsigned)(pix1 * PXB + (g & 1) * 16);
  } else {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int P0 = m0 * LW + j + dx;
      boff[dx] = (unsigned)(P0 * 64 + ((g ^ ((P0 >> 1) & 3)) << 4));
    }
  }
  // this lane's slot in a row of the concat tile (the swizzle term (pixel >> 1) & 3 = (j >> 1) & 3 for every row / pooled index base)
  const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.s_out, 0, a.out_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  // next-squeeze items of this wave: item = wave + NWAVES * i -> (16-pixel block, cout tile)
  const int nitems = wave < NIT - (IPW - 1) * NWAVES ? IPW : IPW - 1;   // wave-uniform

  // the first tile has landed (and every set-up load) -- as a wait the COMPILER sees: behind an asm wait hipcc's wait-count pass still
  // believed the last expand-weight loads pending and put `s_waitcnt vmcnt(1)` / `vmcnt(0)` in front of their first MFMAs INSIDE the
  // tile loop (the fire5 + pool5 form: every step then waited, in the middle of its MFMAs, for the next tile's DMA to land)
  __builtin_amdgcn_s_waitcnt(0x0f70);                 // vmcnt(0)
  __syncthreads();

  int buf = 0;
  while (true) {
    const bool next_ok = tile + nl < band_end;
    if (next_ok) issue_dma(nxt, buf ^ 1);
    unsigned char* sqb = sq + buf * STILE;
    const int tx = cur.tx, ty = cur.ty, n = cur.n;
    const int oy0 = ty * G::RSTEP - (POOL ? a.ptp : 0), ox0 = tx * G::CSTEP - (POOL ? a.plp : 0);
    const int ox = ox0 + j;
    // ---------------- phase B: expand3x3, then expand1x1, on rows [m0, m0 + MT), row by row ----------------
    // INS (wave-uniform, POOL only): every module position this wave produces lies inside the image (out-of-image ones must be -inf
    // under the max; the unpooled forms never store them)
    const bool inside = oy0 + m0 >= 0 && oy0 + m0 + MT <= a.H && ox0 >= 0 && ox0 + XCOLS <= a.W;
    auto conv_pass = [&](auto is3_t, auto ins_t) {
      constexpr bool IS3 = decltype(is3_t)::value;
      constexpr bool INS = decltype(ins_t)::value;
      const float* bias_lds = bl + (IS3 ? a.E : 0) + cb;
      const int chunk = (IS3 ? NQC / 2 : 0) + cp;     // K chunk of the concat tile this wave's couts belong to
      unsigned char* cbase = ctile + chunk * (CPIX * 64) + (NTW == 1 ? t0 * 8 : 0);
      f32x4 run[NTW];                                 // POOL: running vertical maximum
      const float NEGF = __uint_as_float(0xff800000u);
      const bool col_ok = ox >= 0 && ox < a.W;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 acc[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (IS3 && S16) {
          i32x4 bf[NT3];
#pragma unroll
          for (int k = 0; k < NT3; ++k) bf[k] = *reinterpret_cast<const i32x4*>(sqb + boff[k] + m * ROWP);
#pragma unroll
          for (int k = 0; k < NT3; ++k)
#pragma unroll
            for (int t = 0; t < NTW; ++t) mma16<f16>(acc[t], w3r[k][t], bf[k]);
        } else if constexpr (IS3) {
          // one kernel row (3 taps) of B fragments at a time: 12 registers in flight instead of 36
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            i32x4 bf[3];
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) bf[dx] = *reinterpret_cast<const i32x4*>(sqb + boff[dx] + (m + dy) * ROWP);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int t = 0; t < NTW; ++t) mma16<f16>(acc[t], w3r[dy * 3 + dx][t], bf[dx]);
          }
        } else {
          i32x4 bf1;
          if constexpr (S16) bf1 = *reinterpret_cast<const i32x4*>(sqb + boff[5] + m * ROWP);
          else bf1 = *reinterpret_cast<const i32x4*>(sqb + boff[1] + (m + 1) * ROWP);
#pragma unroll
          for (int t = 0; t < NTW; ++t) mma16<f16>(acc[t], w1r[t], bf1);
        }
        if constexpr (!POOL) {
          unsigned int h[2 * NTW];
#pragma unroll
          for (int t = 0; t < NTW; ++t) {
            f32x4 v = acc[t] + *reinterpret_cast<const f32x4*>(bias_lds + t * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
            const i32x2 hi = __builtin_bit_cast(i32x2, hv);
            h[2 * t] = (unsigned)hi[0]; h[2 * t + 1] = (unsigned)hi[1];
          }
          const int PLc = (m0 + m) * 16 + j;
          unsigned char* dst = cbase + PLc * 64 + ((g ^ ((PLc >> 1) & 3)) << 4);
          if constexpr (NTW == 2) *reinterpret_cast<i32x4*>(dst) = i32x4{(int)h[0], (int)h[1], (int)h[2], (int)h[3]};
          else *reinterpret_cast<i32x2*>(dst) = i32x2{(int)h[0], (int)h[1]};
        } else {
          if constexpr (!INS) {
            const int oy = oy0 + m0 + m;
            if (!(col_ok && oy >= 0 && oy < a.H)) {
#pragma unroll
              for (int t = 0; t < NTW; ++t) acc[t] = f32x4{NEGF, NEGF, NEGF, NEGF};
            }
          }
          if (m == 0) {
#pragma unroll
            for (int t = 0; t < NTW; ++t) run[t] = acc[t];
          } else {
#pragma unroll
            for (int t = 0; t < NTW; ++t)
#pragma unroll
              for (int e = 0; e < 4; ++e) run[t][e] = __builtin_fmaxf(run[t][e], acc[t][e]);
          }
          if (m >= 2 && (m & 1) == 0) {
            // pooled row q = m / 2 - 1 of this wave: bias, ONE conversion, horizontal 3-max by two DPP row shifts, ReLU (max
            // commutes with the monotonic x -> fl(x + b) and with the rounding: the values are those of bias -> convert -> max)
            const int q = m / 2 - 1;
            unsigned int vr[2 * NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
              const f32x4 v = run[t] + *reinterpret_cast<const f32x4*>(bias_lds + t * 4);
              const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
              const i32x2 hi = __builtin_bit_cast(i32x2, hv);
              vr[2 * t] = (unsigned)hi[0]; vr[2 * t + 1] = (unsigned)hi[1];
            }
            unsigned int o[2 * NTW];
#pragma unroll
            for (int r = 0; r < 2 * NTW; ++r) {
              // lanes j+1, j+2 of the 16-lane row; lanes 14, 15 read zeros past the row end (bound_ctrl) and store nothing
              const unsigned int s1 = (unsigned int)__builtin_amdgcn_mov_dpp((int)vr[r], 0x101, 0xf, 0xf, true);
              const unsigned int s2 = (unsigned int)__builtin_amdgcn_mov_dpp((int)vr[r], 0x102, 0xf, 0xf, true);
              o[r] = pkmax(pkmax(vr[r], pkmax(s1, s2)), 0u);          // 0u = +0.0 (packed): the ReLU
            }
            if ((j & 1) == 0 && j <= 12) {            // pooled pixel (row rq*NQ + q, column j/2) of the tile's 4 x 7
              const int PLc = (rq * NQ + q) * 7 + (j >> 1);
              unsigned char* dst = cbase + PLc * 64 + ((g ^ ((PLc >> 1) & 3)) << 4);
              if constexpr (NTW == 2) *reinterpret_cast<i32x4*>(dst) = i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
              else *reinterpret_cast<i32x2*>(dst) = i32x2{(int)o[0], (int)o[1]};
            }
#pragma unroll
            for (int t = 0; t < NTW; ++t) run[t] = acc[t];   // row 2q + 2 is also the first row of pooled row q + 1
          }
        }
      }
    };
    if (!POOL || inside) { conv_pass(std::true_type{}, std::true_type{}); conv_pass(std::false_type{}, std::true_type{}); }
    else { conv_pass(std::true_type{}, std::false_type{}); conv_pass(std::false_type{}, std::false_type{}); }
    lds_barrier();                                    // the whole concat tile is in LDS
    // ---------------- phase C: the next module's squeeze1x1 on the tile, one (16-pixel block, 16-cout tile) item at a time ----------------
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int it = wave + NWAVES * i;
      if (it < NIT) {                                 // wave-uniform
        const int blk = it / NTS2, t = it - blk * NTS2;
        f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
        const int PLc = blk * 16 + j;
        const unsigned char* cb0 = ctile + PLc * 64 + ((g ^ ((PLc >> 1) & 3)) << 4);
        // item t computes the 16 CONSECUTIVE couts [16t, 16t + 16): MFMA row i <-> cout 16t + i, so a pixel's four lane groups store 32
        // contiguous bytes (whole 32-byte sectors; with the packing's own order -- a lane owning couts 4*NTS2*g + 4t.. -- the items of a
        // pixel interleave 8-byte pieces written by different waves at different times, and the fabric saw the 96-byte rows of
        // fire6's squeeze tensor twice).  The packed kernel keeps cout c in tile (c % (4 NTS2)) / 4, row 4 (c / (4 NTS2)) + c % 4.
        const int cw = 16 * t + j;
        const unsigned char* wb0 = ws2l + ((((cw % (4 * NTS2)) >> 2) * 64 + 4 * (cw / (4 * NTS2)) + (cw & 3) + 16 * g) * 16);
#pragma unroll
        for (int q = 0; q < NQC; ++q)
          mma16<f16>(acc2, *reinterpret_cast<const i32x4*>(wb0 + q * NTS2 * 1024), *reinterpret_cast<const i32x4*>(cb0 + q * (CPIX * 64)));
        int orow, ocol, OH, OW;
        bool ok;
        if constexpr (POOL) {
          const int prow = PLc / 7, pcol = PLc - prow * 7;
          orow = ty * 4 + prow; ocol = tx * 7 + pcol; OH = a.Hp; OW = a.Wp;
          ok = PLc < 28 && orow < OH && ocol < OW;
        } else {
          orow = oy0 + blk; ocol = ox; OH = a.H; OW = a.W;
          ok = orow < OH && ocol < OW;
        }
        const int ch = 16 * t + 4 * g;
        f32x4 v = acc2 + *reinterpret_cast<const f32x4*>(bl + 2 * a.E + ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        const f16x4 hv = {(f16)v[0], (f16)v[1], (f16)v[2], (f16)v[3]};
        const unsigned off = ok ? (unsigned)((((n * OH + orow) * OW + ocol) * a.S2 + ch) * 2) : OOB;   // out of range = dropped
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hv), rout, off, 0, 0);
      }
    }
    if (!next_ok) break;
    // the next tile has landed once at most this step's stores (issued behind its DMA) are outstanding
    if (nitems == 2) vm_wait_x<2>();
    else if (nitems == 1) vm_wait_x<1>();
    else vm_wait_x<0>();
    lds_barrier();                                    // everybody's part of it; everybody is done with the concat tile and this buffer
    tile += nl;
    buf ^= 1;
    cur = nxt;
    advance(nxt);
  }
}

template <bool S16, bool POOL, int NG, int NTW, int RS, int ROWS_T, int NTS2, int WPS>
int launch_dma(const FireXArgs& a, hipStream_t st) {
  using G = GeoX<S16, POOL, ROWS_T>;
  constexpr int NWAVES = (4 * NG / NTW) * RS;
  constexpr int NQC = NG * 4, CPIX = POOL ? 32 : G::ROWS * 16;
  const size_t lds = 2 * (size_t)G::STILE + (size_t)NQC * CPIX * 64 + (size_t)NQC * NTS2 * 1024 + (size_t)(2 * a.E + a.S2) * 4;
  auto kern = &fire_dma<S16, POOL, NG, NTW, RS, ROWS_T, NTS2, WPS>;
  static PerDevice once;
  SQDET_CHECK_HIP(once.run([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
  constexpr int WGPC = WPS * 4 / NWAVES;              // workgroups per CU
  int grid = cu_count() * WGPC;
  if (grid > (a.ntiles + 7) / 8 * 8) grid = (a.ntiles + 7) / 8 * 8;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NWAVES * 64), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

}  // namespace

// SqueezeDet's four shapes (fire2.hip: fire_expand_squeeze_next_eligible); *handled = false -> fire_stream's forms take over.
// "dbg" 70 = always the fire_stream forms (A/B); 71..74 = alternative geometries of one form (A/B).
int fire_dma_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3, const void* ws2,
                    const float* bs2, void* s_out, int n, int h, int w, int s, int e1, int e3, int s2, int pool, int dtype,
                    hipStream_t st, bool* handled) {
  *handled = false;
  const int d = tune(TUNE_DBG);
  if (d == 70 || dtype != SQDET_F16 || e1 != e3) return SQDET_OK;
  const bool f2 = s == 16 && e1 == 64 && s2 == 16 && !pool, f3 = s == 16 && e1 == 64 && s2 == 32 && pool;
  const bool f4 = s == 32 && e1 == 128 && s2 == 32 && !pool, f5 = s == 32 && e1 == 128 && s2 == 48 && pool;
  if (!(f2 || f3 || f4 || f5)) return SQDET_OK;
  FireXArgs a;
  a.sq_in = sq_in; a.s_out = s_out; a.w1 = w1; a.w3 = w3; a.ws2 = ws2; a.b1 = b1; a.b3 = b3; a.bs2 = bs2;
  a.N = n; a.H = h; a.W = w; a.S = s; a.E = e1; a.S2 = s2;
  a.Hp = out_size(h, 3, 2, SQDET_PAD_SAME); a.Wp = out_size(w, 3, 2, SQDET_PAD_SAME);
  a.ptp = pad_before(h, 3, 2, SQDET_PAD_SAME); a.plp = pad_before(w, 3, 2, SQDET_PAD_SAME);
  const int rows_t = (f4 && d != 74 && d != 76 && d != 77) ? 4 : 8;
  if (pool) { a.tiles_x = (a.Wp + 6) / 7; a.tiles_y = (a.Hp + 3) / 4; }
  else { a.tiles_x = (w + XCOLS - 1) / XCOLS; a.tiles_y = (h + rows_t - 1) / rows_t; }
  const long nt = (long)n * a.tiles_x * a.tiles_y;
  const long xb = (long)n * h * w * s * 2, yb = (pool ? (long)n * a.Hp * a.Wp : (long)n * h * w) * s2 * 2;
  if (nt > 0x3fffffffL || xb >= (1L << 31) || yb >= (1L << 31)) return SQDET_OK;   // 32-bit buffer offsets
  a.ntiles = (int)nt;
  a.in_bytes = (unsigned)xb; a.out_bytes = (unsigned)yb;
  int rc;
  //                        S16   POOL  NG NTW RS ROWS NTS2 WPS
  if (f2) rc = d == 71 ? launch_dma<true, false, 1, 2, 2, 8, 1, 3>(a, st) : launch_dma<true, false, 1, 2, 4, 8, 1, 4>(a, st);
  else if (f3) rc = launch_dma<true, true, 1, 2, 2, 8, 2, 4>(a, st);
  else if (f4) {
    // (same box, us: fire_stream's form 28.5; 8 waves x 2 tiles x 4-row tile, two workgroups per CU 27.2 -- the default; one tile per
    // wave 29.2; 8-row tiles: one 8-wave workgroup 30.9, one 16-wave workgroup 29.9 / 31.3)
    if (d == 74) rc = launch_dma<false, false, 2, 2, 2, 8, 2, 2>(a, st);
    else if (d == 75) rc = launch_dma<false, false, 2, 1, 1, 4, 2, 4>(a, st);
    else if (d == 76) rc = launch_dma<false, false, 2, 2, 4, 8, 2, 4>(a, st);
    else if (d == 77) rc = launch_dma<false, false, 2, 1, 2, 8, 2, 4>(a, st);
    else rc = launch_dma<false, false, 2, 2, 2, 4, 2, 4>(a, st);
  } else {
    if (d == 78) rc = launch_dma<false, true, 2, 2, 2, 8, 3, 2>(a, st);
    else if (d == 79) rc = launch_dma<false, true, 2, 1, 2, 8, 3, 4>(a, st);
    else rc = launch_dma<false, true, 2, 1, 1, 8, 3, 4>(a, st);
  }
  if (rc != SQDET_OK) return rc;
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
