// Fused fire module, persistent streaming form, for the LARGE feature maps with FEW channels
// (fire2..fire5 of SqueezeDet: squeeze <= 32 fp16 channels, expand 64/128; reference
// src/nets/squeezeDet.py:81-106).  These modules are HBM-bound (fire2: 58 FLOP/B): the kernel is built
// so that the only thing a workgroup ever waits for is the input stream.
//
//   * PERSISTENT workgroups (one or two per CU) walk 8 x 16 output tiles; every XCD owns a contiguous
//     band of tiles so the halo re-reads of neighbouring tiles hit that XCD's L2.
//   * A wave owns one PAIR of 16-cout MFMA tiles of expand3x3 and of expand1x1 (8 consecutive output
//     channels per lane -> 16-byte stores) for HALF of the tile rows, and keeps ALL its weights -- 18
//     expand3x3 + 2 expand1x1 fragments -- in REGISTERS for its whole life: the steady state reads no
//     weights at all.  (The squeeze weights live in LDS, copied once.)
//   * Software pipeline per tile:   A  squeeze MFMAs on the 10 x 18 halo from the input fragments
//     PREFETCHED during the previous tile, bias + ReLU -> LDS squeeze tile (double-buffered);
//     P  issue the 16-byte loads of the NEXT tile's halo into registers;   barrier;
//     B  expand3x3 (9 taps) + expand1x1 (centre tap) from LDS, bias + ReLU, stores.
//     One barrier per tile; the input loads have a whole phase B (+ the other workgroup) to land.
// Accumulation order equals conv3x3_tile / conv1x1 (chunk-major, taps 0..8): results are bitwise
// those of the unfused kernels -- except the float16 S = 16 modules (fire2 / fire3), whose expand3x3 pairs two
// taps per MFMA (PAIR below): equal up to float32 summation order.
// Where a tile's time goes was measured per segment with -DSQDET_FIRE_TIMING + tools/fire_timing.py (s_memtime
// deltas): of ~10 k cycles per tile of fire3+pool3 the largest single item was every wave WAITING TO ISSUE its 16
// prefetch loads (~2800 cycles: eight waves reach that point together and the CU has one address pipe); the pooled
// kernels therefore trickle the loads through the expand3x3's K-steps (SPREAD below).
//
// POOL variant (fire3 + pool3, fire5 + pool5 of SqueezeDet, nets/squeezeDet.py:49-57): the 3x3 / stride-2 SAME
// max-pool that follows the module is taken IN REGISTERS and only the pooled tensor is written -- the
// module's full-resolution output (the largest write of the net) and the pool kernel's read of it never
// exist.  A tile is 9 x 16 module outputs -> 4 x 7 pooled pixels (tiles overlap by one row / two columns, like
// the stem's strips); a wave holds 5 rows: vertical 3-max on plain registers, horizontal 3-max by two DPP row
// shifts (the C/D layout puts column j in lane j of a 16-lane DPP row), ReLU once on the pooled value,
// out-of-image positions are -inf (TF SAME pooling never picks padding).
#include <type_traits>

#include "conv_common.h"

namespace sqdet {

constexpr int SCOLS = 16;                        // module-output columns per tile (one MFMA pixel block per row)
template <bool POOL> struct Geo {
  static constexpr int ROWS = POOL ? 9 : 8;      // module-output rows per tile
  static constexpr int HP = (ROWS + 2) * (SCOLS + 2);   // halo pixels of the squeeze tile (198 / 180)
  static constexpr int BLK = (HP + 15) / 16;     // 16-pixel blocks of the halo
  // LDS row pitch of the squeeze tile: 24 pixels, not 18.  With a pitch that is a multiple of 8 the swizzle term
  // ((pixel >> 1) & 3) of a fragment read is the same for every tile row, so the MT reads of one tap are ONE address
  // register + immediate row offsets (with pitch 18 every (tap, row) needed its own address: 180 VALU per tile).
  static constexpr int LW = 24;
  static constexpr int LPIX = (ROWS + 2) * LW;   // pixels of the LDS tile (incl. the unused pitch columns)
  static constexpr int TILE = LPIX * 64;         // bytes of one squeeze tile (one 64-byte chunk per pixel)
  static constexpr int RSTEP = POOL ? 8 : 8;     // module-output rows between tile origins
  static constexpr int CSTEP = POOL ? 14 : 16;   // columns between tile origins
};

// packed max written as the instruction (the generic max canonicalises both inputs first)
template <typename T> __device__ __forceinline__ unsigned int pmax(unsigned int a, unsigned int b);
template <> __device__ __forceinline__ unsigned int pmax<f16>(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <> __device__ __forceinline__ unsigned int pmax<float>(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// -DSQDET_FIRE_TIMING (experiments only): per-wave s_memtime totals of the tile loop's segments, read back with
// sqdet_debug_fire_timing (tools/fire_timing.py).  Compiled out otherwise.
#ifdef SQDET_FIRE_TIMING
__device__ unsigned long long g_fire_timing[2048 * 8];
#define FT_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ft_acc[k] += now_ - ft_last; ft_last = now_; } while (0)
#else
#define FT_MARK(k) do {} while (0)
#endif

struct FireSArgs {
  const void* x;
  void* y;
  const void *ws, *w1, *w3;
  const float *bs, *b1, *b3;
  int N, H, W, Cin, S, E;      // E = expand1x1 = expand3x3 filters
  int tiles_x, tiles_y, ntiles;
  int x_pieces;                // Cin*sizeof(T)/16
  unsigned x_bytes, y_bytes;   // tensor sizes (< 2^31: 32-bit buffer offsets)
  int Hp, Wp, ptp, plp;        // POOL: pooled output dims and the SAME pads (top / left) of the 3x3/s2 pool
  // NTS2 > 0 (squeeze-out form): the NEXT module's squeeze1x1 (packed kernel, bias, S2 channels) and its output tensor
  const void* ws2;
  const float* bs2;
  void* s_out;
  int S2;
  // not NULL (whole-module, unpooled form only): the module's OWN squeeze tensor [N,H,W,S] is also written (training)
  void* sq_keep;
};

// RS = row split: the tile rows are divided among RS waves per cout pair (NWAVES = cout pairs x RS).  POOL: a wave
// owns 4/RS pooled rows and computes the 2*(4/RS)+1 module rows under them (neighbouring waves both compute the row
// they share); otherwise 8/RS rows.
// PAIR (squeeze depth = half a 64-byte chunk, i.e. S = 16 in float16): two taps of the expand3x3 share one MFMA --
// lane groups 0,1 carry tap 2p's 16 channels and groups 2,3 tap 2p+1's (the last pair's second half is zero) -- so
// the 3x3 takes 5 K-steps instead of 9 half-empty ones, 25 instead of 45 B-fragment reads, and its resident weights
// 40 instead of 72 registers.  (The two taps' products are then summed inside one MFMA instead of two: equal to the
// three-conv path up to float32 summation order, no longer bitwise.)
// SQIN: `x` IS the module's squeeze tensor [N,H,W,S] (produced by the chain kernel of the previous module, chain.hip):
// phase A is a copy -- the prefetched 16-byte pieces go straight into the LDS squeeze tile (out-of-image pieces and the
// channel padding arrive as zeros: exactly the SAME padding of the squeeze tensor) -- and the module reads 32-64 bytes
// per pixel instead of 128-256 (NCHX = 1, NTS unused).
// NTS2 > 0: the module's concat tensor (POOL: its pooled form) is NOT written; the tile's rounded float16 results go to an LDS
// tile in the B-fragment layout of the NEXT module's squeeze1x1 (lane group g of cout pair cp holds piece g of K chunk
// coff/32 + cp), and after one more barrier the waves run that squeeze on the tile's 128 pixels (16-pixel block x NTS2
// cout tiles per wave, chunks in ascending concat-channel order = the canonical accumulation order) and write the
// S2-channel squeeze tensor: 32-96 bytes per pixel leave the kernel instead of 256-512.  (POOL: the tile's 4 x 7 pooled
// pixels fill two 16-pixel blocks, the last four positions unused.)
template <typename T, int NCHX, int NTS, int NWAVES, int PF, bool POOL, int RS, bool PAIR = false, bool SQIN = false, int NTS2 = 0>
__global__ __launch_bounds__(NWAVES * 64, 8 / NWAVES) void fire_stream(FireSArgs a) {   // 2 waves per SIMD: 256 VGPRs
  static_assert(NTS2 == 0 || sizeof(T) == 2, "the squeeze-out form is float16 only");
  constexpr int KG = Tr<T>::KG;
  constexpr int NT3 = PAIR ? 5 : 9;                  // K-steps of the expand3x3
  constexpr int KC = 4 * KG;
  constexpr int SHP = Geo<POOL>::HP, SBLK = Geo<POOL>::BLK, STILE = Geo<POOL>::TILE, LW = Geo<POOL>::LW;
  constexpr int MB = (SBLK + NWAVES - 1) / NWAVES;   // halo pixel blocks per wave in phase A
  constexpr int NQ = 4 / RS;                         // POOL: pooled rows per wave
  constexpr int MT = POOL ? 2 * NQ + 1 : 8 / RS;     // module rows per wave in phase B
  constexpr int NG = NWAVES / RS / 2;                // 64-cout groups of each expand conv
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* sq = lds;                           // [2][STILE]
  unsigned char* wsl = lds + 2 * STILE;              // squeeze weights [NCHX][NTS][64 lanes][16 B]
  unsigned char* w1l = wsl + NCHX * NTS * 1024;      // expand1x1 weights [E/16 tiles][64 lanes][16 B]
  constexpr int NQC = NG * 4;                         // 64-byte K chunks of the concat tensor (2E / 32 channels)
  unsigned char* ctile = w1l + NG * 4 * 1024;         // NTS2: [NQC chunks][128 pixels][4 x 16 B swizzled]
  constexpr int CPIX = POOL ? 32 : 128;               // pixels of the concat tile
  unsigned char* ws2l = ctile + (NTS2 ? NQC * CPIX * 64 : 0);   // NTS2: next squeeze weights [NQC][NTS2][64 lanes][16 B]
  float* bl = reinterpret_cast<float*>(ws2l + (NTS2 ? NQC * NTS2 * 1024 : 0));   // biases [b1 | b3 | bs | bs2]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // in an SGPR: what follows from it is wave-uniform
  const int j = lane & 15, g = lane >> 4;

  // ---- one-time set-up ------------------------------------------------------------------------
  {
    if constexpr (!SQIN) {
      const i32x4* src = reinterpret_cast<const i32x4*>(a.ws);
      for (int i = threadIdx.x; i < NCHX * NTS * 64; i += NWAVES * 64) reinterpret_cast<i32x4*>(wsl)[i] = src[i];
    }
    const i32x4* src1 = reinterpret_cast<const i32x4*>(a.w1);
    for (int i = threadIdx.x; i < NG * 4 * 64; i += NWAVES * 64) reinterpret_cast<i32x4*>(w1l)[i] = src1[i];
    // biases -> LDS [b1 (E) | b3 (E) | bs (S)]: read back with ds_read (lgkmcnt).  A global bias load inside
    // the tile loop would sit behind the prefetched input loads and the stores in the in-order vmcnt queue
    // and drain them every time it is waited for.
    for (int i = threadIdx.x; i < 2 * a.E + (SQIN ? 0 : a.S); i += NWAVES * 64)
      bl[i] = i < a.E ? a.b1[i] : (i < 2 * a.E ? a.b3[i - a.E] : a.bs[i - 2 * a.E]);
    if constexpr (NTS2 > 0) {
      const i32x4* src2 = reinterpret_cast<const i32x4*>(a.ws2);
      for (int i = threadIdx.x; i < NQC * NTS2 * 64; i += NWAVES * 64) reinterpret_cast<i32x4*>(ws2l)[i] = src2[i];
      for (int i = threadIdx.x; i < a.S2; i += NWAVES * 64) bl[2 * a.E + a.S + i] = a.bs2[i];   // (behind the unused bs slot when SQIN)
    }
    // channel padding of the squeeze tile (S*sizeof(T) < 64 bytes) is zero in both buffers, forever
    const int s_pieces = a.S * (int)sizeof(T) / 16;
    const int pad = 4 - s_pieces;
    constexpr int LPIX = Geo<POOL>::LPIX;
    for (int idx = threadIdx.x; idx < 2 * LPIX * pad; idx += NWAVES * 64) {
      const int bsel = idx / (LPIX * pad), r = idx - bsel * LPIX * pad;
      const int P = r / pad, q = s_pieces + (r - P * pad);
      *reinterpret_cast<i32x4*>(sq + bsel * STILE + P * 64 + ((q ^ ((P >> 1) & 3)) << 4)) = i32x4{0, 0, 0, 0};
    }
  }
  const int cp = wave / RS, rq = wave % RS;          // cout pair, row part
  const int m0 = POOL ? rq * 2 * NQ : rq * MT;       // first module row of this wave (POOL: boundary rows are shared)
  // This wave's two MFMA tiles t = 0,1 cover the 32 consecutive couts [cp*32, cp*32+32) with
  //   tile row i  <->  cout cp*32 + 8*(i>>2) + 4*t + (i&3),
  // so lane group g ends up with the 8 consecutive couts cp*32 + 8g + [0,8): one 16-byte store per pixel,
  // and the four lane groups write 64 contiguous bytes per pixel per store instruction (measured
  // 6.2-6.8 TB/s against 3.4-5.7 TB/s for lane-owned 32/64-byte runs, tools/microbench/store_patterns.hip).
  // The packed weights hold cout c of a 64-cout group in tile (c%16)/4, row 4*(c/16) + c%4 (conv.hip
  // pack_weights_kernel): every lane gathers its A-fragment row from there -- once, weights are resident.
  int wsrc[2];   // i32x4 index of this lane's fragment data inside one (group, tap) block of 4 tiles, for t = 0,1
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int c = (cp & 1) * 32 + 8 * (j >> 2) + 4 * t + (j & 3);     // cout within the 64-cout group
    wsrc[t] = ((c & 15) >> 2) * 64 + (4 * (c >> 4) + (c & 3)) + 16 * g;
  }
  const int group = cp >> 1;
  i32x4 w3r[NT3][2];
  {
    const i32x4* p3 = reinterpret_cast<const i32x4*>(a.w3) + (size_t)group * 9 * 4 * 64;
#pragma unroll
    for (int p = 0; p < NT3; ++p) {
      if constexpr (PAIR) {
        // k-group g of the paired fragment = k-group (g & 1) of tap 2p + (g >> 1); tap 9 does not exist: zeros
        const int tap = 2 * p + (g >> 1);
#pragma unroll
        for (int t = 0; t < 2; ++t)
          w3r[p][t] = tap < 9 ? p3[tap * 4 * 64 + wsrc[t] - 16 * g + 16 * (g & 1)] : i32x4{0, 0, 0, 0};
      } else {
        w3r[p][0] = p3[p * 4 * 64 + wsrc[0]];
        w3r[p][1] = p3[p * 4 * 64 + wsrc[1]];
      }
    }
  }
  const int cb = cp * 32 + g * 8;                     // this lane's 8 consecutive couts (of both expand convs)

  // tiles of this workgroup: XCD x (= blockIdx % 8) owns the contiguous band [x*per, (x+1)*per)
  const int xcd = blockIdx.x & 7, lid = blockIdx.x >> 3, nl = gridDim.x >> 3;   // gridDim.x is a multiple of 8
  const int per = (a.ntiles + 7) >> 3;
  const int band_end = min(a.ntiles, (xcd + 1) * per);
  int tile = xcd * per + lid;

  // Raw buffer resources: an out-of-range offset makes a load return 0 (exactly the zero padding) and
  // drops a store, so neither needs a branch -- and with branch-free memory instructions the compiler
  // knows how many are in flight: its s_waitcnt for the prefetched input lets the stores issued after it
  // stay outstanding instead of draining the whole in-order queue.
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.y_bytes, 0x00020000);
  constexpr unsigned OOB = 0xfffffff0u;
  const int ctot = 2 * a.E;

  // PF tiles of input are in flight (registers): xr[q] / inimg[q] belong to the tile processed q steps ahead.
  // Loads are issued unconditionally (an invalid tile turns every offset out of range: no traffic, zeros)
  // so the number of memory instructions per step is fixed and the waits stay exact.
  // The prefetch is split in two: prep_loads computes one byte offset per 16-pixel block (out of range for pixels outside
  // the image / the halo / a tile past the band: such loads return zeros, exactly the SAME padding, and move no data),
  // issue_loads sends the block's NCHX 16-byte loads.  SPREAD (the pooled float16 kernels, whose epilogues store
  // little): inside the tile loop the loads go out a few per K-step of the expand3x3 instead of all at once -- eight
  // waves pushing 16 loads each through the CU's one address pipe at the same moment made every wave wait ~2800
  // cycles of a 10 k-cycle tile (measured with -DSQDET_FIRE_TIMING): fire3+pool3 103 -> 97 us, fire5+pool5 68 -> 64 us.
  // The unpooled kernels (whose phase B already carries 16 stores per wave) were 5 % SLOWER that way: they keep the
  // loads in one batch ahead of the barrier; so do the float32 forms (no registers for the offsets).
  constexpr bool SPREAD = POOL && sizeof(T) == 2;
  constexpr unsigned OOBL = 0x80000000u;            // (+ c2 * 64 stays out of range: tensors are < 2 GiB)
  constexpr int NLOADS = MB * NCHX, LPP = (NLOADS + NT3 - 1) / NT3;   // loads per K-step of the expand3x3
  i32x4 xr[PF][MB][NCHX];
  bool inimgs[PF][MB];
  unsigned offs[MB];
  // Tile coordinates are carried, not decoded: a workgroup's tiles are `nl` apart, so (tx, ty, n) advance by the mixed-radix
  // digits of nl with two carries (~10 scalar instructions; the two divisions of a decode were ~50, twice per tile: the
  // tile loop's instruction stream -- SQ_INSTS per wave and tile, profiles/r03_sq_counters.txt -- was 30 % scalar).
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
  TileXY cur = decode(tile), pre = cur;     // of `tile` and of the tile PF steps ahead (whose halo prep_loads fetches)
  // Per-lane, tile-invariant parts of the halo addressing, once per workgroup: relx[mb] = byte offset of this lane's piece of
  // halo pixel P relative to the halo origin (0x80000000 for a piece that does not exist: base + relx is then out of range,
  // tensors being < 2 GiB), rcp[mb] = (halo row << 16 | halo column), and -- SQIN -- the LDS address phase A copies the piece to
  // (a never-read pitch column for pieces that do not exist: the four stores of phase A are unconditional).
  // (the forms that start from the squeeze tensor only -- the forward plan's: the whole-module forms have no registers to spare)
  constexpr bool PRE = SQIN;
  unsigned relx[MB], rcp[MB], ldsoff[MB];
#pragma unroll
  for (int mb = 0; mb < (PRE ? MB : 0); ++mb) {
    const int P = (wave + NWAVES * mb) * 16 + j;
    const int r = P / (SCOLS + 2), c = P - r * (SCOLS + 2);
    const bool exists = P < SHP && (!SQIN || g * KG < a.Cin);
    relx[mb] = exists ? (unsigned)(((r * a.W + c) * a.Cin + g * KG) * (int)sizeof(T)) : OOBL;
    rcp[mb] = ((unsigned)r << 16) | (unsigned)c;
    const int PL = P < SHP ? r * LW + c : SCOLS + 4;          // (column 20 of row 0: the LDS pitch is 24, columns 18.. are never read)
    ldsoff[mb] = (unsigned)(PL * 64 + ((g ^ ((PL >> 1) & 3)) << 4));
  }
  // (PRE) the LDS byte offset, inside a squeeze tile, of this lane's B fragment of expand3x3 K-step `tap` at its first row
  // (load_tap below has the derivation): 5 or 9 registers instead of ~6 VALU instructions per K-step and tile
  constexpr int NT3P = PAIR ? 5 : 9;
  unsigned tapoff[NT3P];
  if constexpr (PRE) {
    const int m0p = POOL ? (wave % RS) * 2 * (4 / RS) : (wave % RS) * (8 / RS);
#pragma unroll
    for (int tap = 0; tap < NT3P; ++tap) {
      int P0, piece;
      if constexpr (PAIR) {
        const int ta = 2 * tap, tb = 2 * tap + 1 < 9 ? 2 * tap + 1 : 8;
        const int ca = (ta / 3) * LW + ta % 3, cbb = (tb / 3) * LW + tb % 3;
        P0 = j + (g >= 2 ? cbb : ca);
        piece = (2 * tap + 1 < 9) ? (g & 1) : g;
      } else {
        const int dy = tap / 3, dx = tap - dy * 3;
        P0 = dy * LW + j + dx;
        piece = g;
      }
      tapoff[tap] = (unsigned)((P0 + LW * m0p) * 64 + ((piece ^ ((P0 >> 1) & 3)) << 4));
    }
  }
  auto prep_loads = [&](const TileXY& tc, bool tile_ok, bool (&inimg)[MB]) {
    const int hy0 = tc.ty * Geo<POOL>::RSTEP - (POOL ? a.ptp : 0) - 1, hx0 = tc.tx * Geo<POOL>::CSTEP - (POOL ? a.plp : 0) - 1;
    // common case (wave-uniform): a real tile whose halo lies inside the image -- no per-pixel bounds tests
    const bool allin = tile_ok && hy0 >= 0 && hy0 + Geo<POOL>::ROWS + 2 <= a.H && hx0 >= 0 && hx0 + SCOLS + 2 <= a.W;
    if constexpr (!PRE) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int P = (wave + NWAVES * mb) * 16 + j;
        const int r = P / (SCOLS + 2), c = P - r * (SCOLS + 2);
        const int iy = hy0 + r, ix = hx0 + c;
        inimg[mb] = allin || (tile_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W);
        const unsigned base = (unsigned)((((tc.n * a.H + iy) * a.W + ix) * a.Cin + g * KG) * (int)sizeof(T));
        offs[mb] = (inimg[mb] && P < SHP && (!SQIN || g * KG < a.Cin)) ? base : OOBL;
      }
      return;
    }
    const unsigned base = (unsigned)((((tc.n * a.H + hy0) * a.W + hx0) * a.Cin) * (int)sizeof(T));   // (modular when hy0 / hx0 = -1)
    if (allin) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) { inimg[mb] = true; offs[mb] = base + relx[mb]; }
    } else {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int iy = hy0 + (int)(rcp[mb] >> 16), ix = hx0 + (int)(rcp[mb] & 0xffffu);
        inimg[mb] = tile_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        offs[mb] = inimg[mb] ? base + relx[mb] : OOBL;
      }
    }
  };
  auto issue_loads = [&](int first, int last, i32x4 (&xq)[MB][NCHX]) {   // (Cin fills whole 64-byte chunks: stream_shape)
#pragma unroll
    for (int l = 0; l < NLOADS; ++l)
      if (l >= first && l < last) xq[l / NCHX][l % NCHX] = __builtin_amdgcn_raw_buffer_load_b128(rx, offs[l / NCHX] + (l % NCHX) * 64, 0, 0);
  };

#ifdef SQDET_FIRE_TIMING
  unsigned long long ft_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ft_last = __builtin_amdgcn_s_memtime();
#endif
  auto step = [&](int tile, i32x4 (&xq)[MB][NCHX], bool (&inimg)[MB], unsigned char* sqb) {
    FT_MARK(7);
    // coordinates of THIS tile (all wave-uniform)
    if constexpr (!PRE) cur = decode(tile);
    const int tx = cur.tx, ty = cur.ty, n = cur.n;
    const int oy0 = ty * Geo<POOL>::RSTEP - (POOL ? a.ptp : 0), ox0 = tx * Geo<POOL>::CSTEP - (POOL ? a.plp : 0);
    // the whole halo inside the image (3 tiles out of 4): no SAME-padding select on the squeeze tile
    const bool haloin = oy0 >= 1 && oy0 + Geo<POOL>::ROWS + 1 <= a.H && ox0 >= 1 && ox0 + SCOLS + 1 <= a.W;
    // ---------------- phase A: squeeze on the halo, from the prefetched fragments ----------------
    if constexpr (SQIN) {
      // the prefetched pieces ARE the squeeze tile: lane (j, g) holds piece g of halo pixel P
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) *reinterpret_cast<i32x4*>(sqb + ldsoff[mb]) = xq[mb][0];
      FT_MARK(0);
    } else {
      f32x4 acc[MB][NTS];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int t = 0; t < NTS; ++t) acc[mb][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NCHX; ++c) {
        i32x4 af[NTS];
#pragma unroll
        for (int t = 0; t < NTS; ++t) af[t] = *reinterpret_cast<const i32x4*>(wsl + ((c * NTS + t) * 64 + lane) * 16);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int t = 0; t < NTS; ++t) mma16<T>(acc[mb][t], af[t], xq[mb][c]);
      }
      FT_MARK(0);
      auto store_squeeze = [&](auto pad_t) {
#pragma unroll
        for (int t = 0; t < NTS; ++t) {
          const int ch0 = g * 4 * NTS + 4 * t;
          if (ch0 < a.S) {
            const f32x4 biass = *reinterpret_cast<const f32x4*>(bl + 2 * a.E + ch0);
            const int q = ch0 / KG;
            const int sub = (ch0 - q * KG) * (int)sizeof(T);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
              const int P = (wave + NWAVES * mb) * 16 + j;
              if (P < SHP) {
                f32x4 v = acc[mb][t] + biass;
                v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
                if constexpr (decltype(pad_t)::value)
                  if (!inimg[mb]) v = f32x4{0.f, 0.f, 0.f, 0.f};   // SAME padding of the squeeze tensor
                const int hr = P / (SCOLS + 2);
                const int PL = hr * LW + (P - hr * (SCOLS + 2));   // position in the LDS tile (row pitch LW)
                store4<T>(reinterpret_cast<T*>(sqb + PL * 64 + ((q ^ ((PL >> 1) & 3)) << 4) + sub), v);
                if constexpr (!POOL && NTS2 == 0) {
                  if (a.sq_keep) {    // the tile's own pixels (not the halo ring) inside the image
                    const int hc = P - hr * (SCOLS + 2);
                    if (inimg[mb] && hr >= 1 && hr <= Geo<POOL>::ROWS && hc >= 1 && hc <= SCOLS)
                      store4<T>(reinterpret_cast<T*>(a.sq_keep) + ((size_t)(n * a.H + oy0 + hr - 1) * a.W + ox0 + hc - 1) * a.S + ch0, v);
                  }
                }
              }
            }
          }
        }
      };
      if (haloin) store_squeeze(std::false_type{});
      else store_squeeze(std::true_type{});
    }
    // ---------------- prefetch: the halo of the tile PF steps ahead, into the registers just consumed ----------------
    FT_MARK(1);
    // ---------------- prefetch: addresses of the halo of the tile PF steps ahead (its loads go out during phase B,
    // into the registers phase A has just consumed) ----------------
    if constexpr (PRE) { advance(cur); advance(pre); }
    else pre = decode(tile + PF * nl);
    prep_loads(pre, tile + PF * nl < band_end, inimg);
    if constexpr (!SPREAD) issue_loads(0, NLOADS, xq);
    FT_MARK(2);
    __syncthreads();
    FT_MARK(3);
    // ---------------- phase B: expand3x3, then expand1x1, on rows [m0, m0 + MT) ----------------
    const int ox = ox0 + j;
    // INS (wave-uniform): every module row and column this wave produces lies inside the image -- the common case --
    // so the per-element edge selects (a third of the epilogue's VALU work) are compiled out of that path
    const bool inside = oy0 + m0 >= 0 && oy0 + m0 + MT <= a.H && ox0 >= 0 && ox0 + SCOLS <= a.W;
    auto epilogue = [&](f32x4 (&acc)[MT][2], const float* bias_lds, int coff, auto ins_t) {
      constexpr bool INS = decltype(ins_t)::value;
      f32x4 bias[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) bias[t] = *reinterpret_cast<const f32x4*>(bias_lds + cb + t * 4);
      if constexpr (!POOL) {
        const unsigned off0 = (unsigned)(((((n * a.H + oy0 + m0) * a.W + ox) * ctot) + coff + cb) * (int)sizeof(T));
        const unsigned yrow = (unsigned)(a.W * ctot * (int)sizeof(T));
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const unsigned off = (INS || (ox < a.W && oy0 + m0 + m < a.H)) ? off0 + m * yrow : OOB;   // OOB stores are dropped
          f32x4 v[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            v[t] = acc[m][t] + bias[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[t][e] = fmaxf(v[t][e], 0.f);
          }
          if constexpr (NTS2 > 0) {
            const f16x8 h = {(f16)v[0][0], (f16)v[0][1], (f16)v[0][2], (f16)v[0][3], (f16)v[1][0], (f16)v[1][1], (f16)v[1][2], (f16)v[1][3]};
            const int PLc = (m0 + m) * SCOLS + j;
            *reinterpret_cast<i32x4*>(ctile + (coff / 32 + cp) * (CPIX * 64) + PLc * 64 + ((g ^ ((PLc >> 1) & 3)) << 4)) = __builtin_bit_cast(i32x4, h);
          } else if constexpr (sizeof(T) == 2) {
            const f16x8 h = {(f16)v[0][0], (f16)v[0][1], (f16)v[0][2], (f16)v[0][3], (f16)v[1][0], (f16)v[1][1], (f16)v[1][2], (f16)v[1][3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, h), ry, off, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v[0]), ry, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, v[1]), ry, off == OOB ? OOB : off + 16, 0, 0);
          }
        }
      } else {
        constexpr int NR = sizeof(T) == 2 ? 4 : 8;        // 32-bit registers holding this lane's 8 couts of one pixel
        // Vertical 3-max on the RAW float32 accumulators (v_max3_f32), then bias, then ONE conversion per pooled row:
        // max commutes with the monotonic x -> fl(x + b) and with the float16 rounding, so the values are those of
        // bias -> convert -> max, at 36 instead of 66 VALU instructions per pooled row.  Rows / columns outside the
        // map (edge tiles only) are -inf before the max: they never win.
        const float NEGF = __uint_as_float(0xff800000u);
        if constexpr (!INS) {
          const bool col_ok = ox >= 0 && ox < a.W;
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int oy = oy0 + m0 + m;
            if (!(col_ok && oy >= 0 && oy < a.H)) {
              acc[m][0] = f32x4{NEGF, NEGF, NEGF, NEGF};
              acc[m][1] = f32x4{NEGF, NEGF, NEGF, NEGF};
            }
          }
        }
        const int pc = tx * 7 + (j >> 1);                 // pooled column of the even lanes j = 0, 2, .., 12
        const bool lane_ok = (j & 1) == 0 && j <= 12 && pc < a.Wp;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {                    // this wave's pooled rows
          const int pr = ty * 4 + rq * NQ + q;
          f32x4 vm[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              vm[t][e] = __builtin_fmaxf(__builtin_fmaxf(acc[2 * q][t][e], acc[2 * q + 1][t][e]), acc[2 * q + 2][t][e]);
            vm[t] += bias[t];
          }
          unsigned int vr[NR];
          if constexpr (sizeof(T) == 2) {
            const f16x8 h = {(f16)vm[0][0], (f16)vm[0][1], (f16)vm[0][2], (f16)vm[0][3], (f16)vm[1][0], (f16)vm[1][1], (f16)vm[1][2], (f16)vm[1][3]};
            const i32x4 hi = __builtin_bit_cast(i32x4, h);
#pragma unroll
            for (int r = 0; r < 4; ++r) vr[r] = (unsigned)hi[r];
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) { vr[r] = __float_as_uint(vm[0][r]); vr[4 + r] = __float_as_uint(vm[1][r]); }
          }
          unsigned int o[NR];
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            // lanes j+1, j+2 of the 16-lane row; lanes 14, 15 read zeros past the row end (bound_ctrl) and store nothing
            const unsigned int s1 = (unsigned int)__builtin_amdgcn_mov_dpp((int)vr[r], 0x101, 0xf, 0xf, true);
            const unsigned int s2 = (unsigned int)__builtin_amdgcn_mov_dpp((int)vr[r], 0x102, 0xf, 0xf, true);
            o[r] = pmax<T>(pmax<T>(vr[r], pmax<T>(s1, s2)), 0u);        // 0u = +0.0 (packed): the ReLU
          }
          if constexpr (NTS2 > 0) {
            if ((j & 1) == 0 && j <= 12) {          // pooled pixel (row rq*NQ + q, column j/2) of the tile's 4 x 7
              const int PLc = (rq * NQ + q) * 7 + (j >> 1);
              *reinterpret_cast<i32x4*>(ctile + (coff / 32 + cp) * (CPIX * 64) + PLc * 64 + ((g ^ ((PLc >> 1) & 3)) << 4)) =
                  i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]};
            }
            continue;
          }
          const unsigned off = (lane_ok && pr < a.Hp)
              ? (unsigned)(((((n * a.Hp + pr) * a.Wp + pc) * ctot) + coff + cb) * (int)sizeof(T)) : OOB;
          __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, ry, off, 0, 0);
          if constexpr (sizeof(T) == 4)
            __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}, ry, off == OOB ? OOB : off + 16, 0, 0);
        }
      }
    };
    {
      f32x4 acc3[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc3[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
      // B fragments double-buffered by hand, one tap ahead; the scheduling barriers keep the compiler
      // from hoisting all 36 LDS reads to the top (that spilled registers into the tile loop, and every
      // scratch reload drains the in-order vmcnt queue, i.e. waits for the prefetched input)
      // jo / go: j and g behind an empty asm, so the 36 per-(tap,row) LDS addresses are recomputed per tile
      // (a handful of VALU ops) instead of being hoisted out of the tile loop into 36 live registers
      int jo = j, go = g;
      asm volatile("" : "+v"(jo), "+v"(go));
      auto load_tap = [&](int tap, i32x4 (&bf)[MT]) {
        int P0, piece;
        if constexpr (PAIR) {
          // `tap` is the pair index: lane groups 0,1 read pieces 0,1 of tap 2p's pixel, groups 2,3 those of tap 2p+1's;
          // past the ninth tap they read the (always zero) padding pieces 2,3 of tap 8's pixel
          const int ta = 2 * tap, tb = 2 * tap + 1 < 9 ? 2 * tap + 1 : 8;
          const int ca = (ta / 3) * LW + ta % 3, cbb = (tb / 3) * LW + tb % 3;
          const bool hi = go >= 2;
          P0 = jo + (hi ? cbb : ca);
          piece = (2 * tap + 1 < 9) ? (go & 1) : go;
        } else {
          const int dy = tap / 3, dx = tap - dy * 3;
          P0 = dy * LW + jo + dx;
          piece = go;
        }
        const unsigned char* base = PRE ? sqb + tapoff[tap] : sqb + (P0 + LW * m0) * 64 + ((piece ^ ((P0 >> 1) & 3)) << 4);   // LW*(m0+m)/2 = 0 mod 4
#pragma unroll
        for (int m = 0; m < MT; ++m) bf[m] = *reinterpret_cast<const i32x4*>(base + m * (LW * 64));
      };
      constexpr bool DB = PAIR || MB * NCHX * PF < 16 || !POOL;   // (the unpaired pooled fire3 shape has no registers left for the second buffer)
      if constexpr (DB) {
        i32x4 bfa[MT], bfb[MT];
        load_tap(0, bfa);
#pragma unroll
        for (int tap = 0; tap < NT3; ++tap) {
          i32x4 (&cur)[MT] = (tap & 1) ? bfb : bfa;
          i32x4 (&nxt)[MT] = (tap & 1) ? bfa : bfb;
          if (tap + 1 < NT3) load_tap(tap + 1, nxt);
          if constexpr (SPREAD) issue_loads(tap * LPP, (tap + 1) * LPP, xq);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) mma16<T>(acc3[m][t], w3r[tap][t], cur[m]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int tap = 0; tap < NT3; ++tap) {
          i32x4 bf[MT];
          load_tap(tap, bf);
          if constexpr (SPREAD) issue_loads(tap * LPP, (tap + 1) * LPP, xq);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < 2; ++t) mma16<T>(acc3[m][t], w3r[tap][t], bf[m]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      FT_MARK(4);
      if (inside) epilogue(acc3, bl + a.E, a.E, std::true_type{});   // expand3x3 -> channels [E, 2E)
      else epilogue(acc3, bl + a.E, a.E, std::false_type{});
      FT_MARK(5);
    }
    {
      f32x4 acc1[MT][2];
      i32x4 w1f[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) w1f[t] = *reinterpret_cast<const i32x4*>(w1l + (group * 4 * 64 + wsrc[t]) * 16);
      const int P0 = LW + j + 1;                      // centre tap
      const unsigned char* base1 = sqb + (P0 + LW * m0) * 64 + ((g ^ ((P0 >> 1) & 3)) << 4);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const i32x4 bf = *reinterpret_cast<const i32x4*>(base1 + m * (LW * 64));
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          acc1[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
          mma16<T>(acc1[m][t], w1f[t], bf);
        }
      }
      if (inside) epilogue(acc1, bl, 0, std::true_type{});            // expand1x1 -> channels [0, E)
      else epilogue(acc1, bl, 0, std::false_type{});
      FT_MARK(6);
    }
    if constexpr (NTS2 > 0) {
      // ---------------- phase C: the next module's squeeze1x1 on the tile (tile row = one 16-pixel block) ----------------
      __syncthreads();                                    // the whole concat tile is in LDS
      constexpr int NBLK = POOL ? 2 : 8;                  // 16-pixel blocks of the concat tile (POOL: 28 pooled pixels)
      T* so = reinterpret_cast<T*>(a.s_out);
#pragma unroll
      for (int bb = 0; bb < (NBLK + NWAVES - 1) / NWAVES; ++bb) {
        const int blk = wave + bb * NWAVES;
        if (blk >= NBLK) break;
        f32x4 acc2[NTS2];
#pragma unroll
        for (int t = 0; t < NTS2; ++t) acc2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int PLc = blk * SCOLS + j;
        const unsigned char* cb0 = ctile + PLc * 64 + ((g ^ ((PLc >> 1) & 3)) << 4);
#pragma unroll
        for (int q = 0; q < NQC; ++q) {
          const i32x4 bfq = *reinterpret_cast<const i32x4*>(cb0 + q * (CPIX * 64));
#pragma unroll
          for (int t = 0; t < NTS2; ++t)
            mma16<T>(acc2[t], *reinterpret_cast<const i32x4*>(ws2l + ((q * NTS2 + t) * 64 + lane) * 16), bfq);
        }
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
        if (ok) {
          T* dst = so + ((size_t)(n * OH + orow) * OW + ocol) * a.S2 + g * 4 * NTS2;
#pragma unroll
          for (int t = 0; t < NTS2; ++t) {
            f32x4 v = acc2[t] + *reinterpret_cast<const f32x4*>(bl + 2 * a.E + a.S + g * 4 * NTS2 + t * 4);
            v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
            store4<T>(dst + t * 4, v);
          }
        }
      }
    }
  };

  prep_loads(pre, tile < band_end, inimgs[0]);
  issue_loads(0, NLOADS, xr[0]);
  if constexpr (PF == 2) {
    if constexpr (PRE) advance(pre);
    else pre = decode(tile + nl);
    prep_loads(pre, tile + nl < band_end, inimgs[1]);
    issue_loads(0, NLOADS, xr[1]);
  }
  // (from here on `pre` is PF - 1 tiles ahead of `cur` at the top of a step and PF ahead once the step has advanced both)
  __syncthreads();                                    // squeeze weights / padding / biases visible
  if constexpr (PF == 1) {
    int buf = 0;
    for (; tile < band_end; tile += nl, buf ^= 1) step(tile, xr[0], inimgs[0], sq + buf * STILE);
  } else {
    while (tile < band_end) {
      step(tile, xr[0], inimgs[0], sq);
      tile += nl;
      if (tile >= band_end) break;
      step(tile, xr[PF - 1], inimgs[PF - 1], sq + STILE);
      tile += nl;
    }
  }
#ifdef SQDET_FIRE_TIMING
  if (lane == 0 && blockIdx.x * NWAVES + wave < 2048)
    for (int k = 0; k < 8; ++k) g_fire_timing[(blockIdx.x * NWAVES + wave) * 8 + k] = ft_acc[k];
#endif
}

static bool stream_shape(int cin, int s, int e1, int e3, int dtype, int* nchx, int* nts, int* nwaves) {
  if (conv_algo() != 0 || e1 != e3) return false;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  const ConvGeom gs = conv_geom(1, cin, s, dtype), g1 = conv_geom(1, s, e1, dtype), g3 = conv_geom(3, s, e3, dtype);
  if (gs.gather || g1.gather || g3.gather || gs.ngroups != 1) return false;
  if (g1.nchunk != 1 || g3.nchunk != 1 || g1.nt != 4 || g3.nt != 4) return false;   // squeeze fits one 64-byte chunk; E % 64 == 0
  if (e1 % 64 != 0 || (g1.ngroups != 1 && g1.ngroups != 2)) return false;
  if (!(gs.nt == 1 || gs.nt == 2)) return false;
  // input fragments stay in registers: 2 or 4 chunks with two tiles in flight, 8 chunks (8-wave form only) with one
  if (!(gs.nchunk == 2 || gs.nchunk == 4 || (gs.nchunk == 8 && g1.ngroups == 2))) return false;
  if ((cin * esz) % 64 != 0 || (s * esz) % 16 != 0 || s % 4 != 0) return false;   // the input fills whole 64-byte chunks
  *nchx = gs.nchunk; *nts = gs.nt; *nwaves = 4 * g1.ngroups;
  return true;
}

bool fire_stream_eligible(int cin, int s, int e1, int e3, int dtype) {
  int a, b, c;
  return stream_shape(cin, s, e1, e3, dtype, &a, &b, &c);
}

template <typename T, int NCHX, int NTS, int NWAVES, bool POOL, int RS, bool SQIN = false>
static void launch_stream(const FireSArgs& a, hipStream_t st) {
  const size_t lds = 2 * (size_t)Geo<POOL>::TILE + (size_t)NCHX * NTS * 1024 + (size_t)(NWAVES / RS / 2) * 4 * 1024 + (size_t)(2 * a.E + a.S) * 4;
  // persistent: 8 waves per CU (the register-resident weights + prefetched input allow 2 per SIMD)
  int grid = cu_count() * (8 / NWAVES);
  if (grid > (a.ntiles + 7) / 8 * 8) grid = (a.ntiles + 7) / 8 * 8;
  // two tiles of input in flight when their fragments fit the register budget next to the resident weights
  constexpr int MBH = (Geo<POOL>::BLK + NWAVES - 1) / NWAVES;
  // squeeze depth of half a chunk (S = 16 in float16): the tap-paired expand3x3 ("dbg" 16 keeps the unpaired form for A/B)
  if constexpr (sizeof(T) == 2 && NTS == 1 && RS == 2) {
    if (a.S * (int)sizeof(T) == 32 && tune(TUNE_DBG) != 16) {
      if constexpr (2 * MBH * NCHX <= 16) {
        if (tune(TUNE_DBG) != 8) {
          hipLaunchKernelGGL((fire_stream<T, NCHX, NTS, NWAVES, 2, POOL, RS, true, SQIN>), dim3(grid), dim3(NWAVES * 64), lds, st, a);
          return;
        }
      }
      hipLaunchKernelGGL((fire_stream<T, NCHX, NTS, NWAVES, 1, POOL, RS, true, SQIN>), dim3(grid), dim3(NWAVES * 64), lds, st, a);
      return;
    }
  }
  if constexpr (2 * MBH * NCHX <= 16) {   // (these compile without spills; a spill in the tile loop drains vmcnt)
    if (tune(TUNE_DBG) != 8) {
      hipLaunchKernelGGL((fire_stream<T, NCHX, NTS, NWAVES, 2, POOL, RS, false, SQIN>), dim3(grid), dim3(NWAVES * 64), lds, st, a);
      return;
    }
  }
  hipLaunchKernelGGL((fire_stream<T, NCHX, NTS, NWAVES, 1, POOL, RS, false, SQIN>), dim3(grid), dim3(NWAVES * 64), lds, st, a);
}

template <typename T, bool POOL>
static bool dispatch_stream(const FireSArgs& a, int nchx, int nts, int nwaves, hipStream_t st) {
  // nwaves = 4 per 64-cout group at row split 2.  One group (E = 64) can also run as 8 waves at row split 4 (half the
  // input fragments per wave, two tiles in flight): measured 5-12 % SLOWER on MI355X, kept behind "dbg" & 64.
  const bool rs4 = nwaves == 4 && (tune(TUNE_DBG) & 64);
#define SQDET_FS(NC, NS, NW) \
  if (nchx == NC && nts == NS && nwaves == NW) { launch_stream<T, NC, NS, NW, POOL, 2>(a, st); return true; }
#define SQDET_FS4(NC, NS) \
  if (rs4 && nchx == NC && nts == NS) { launch_stream<T, NC, NS, 8, POOL, 4>(a, st); return true; }
  SQDET_FS4(2, 1) SQDET_FS4(4, 1) SQDET_FS4(2, 2) SQDET_FS4(4, 2)
  SQDET_FS(2, 1, 4) SQDET_FS(4, 1, 4) SQDET_FS(2, 2, 4) SQDET_FS(4, 2, 4)
  SQDET_FS(2, 1, 8) SQDET_FS(4, 1, 8) SQDET_FS(2, 2, 8) SQDET_FS(4, 2, 8) SQDET_FS(8, 1, 8) SQDET_FS(8, 2, 8)
#undef SQDET_FS
#undef SQDET_FS4
  return false;
}

// *handled = false: shape not covered (fire_fused_launch / the three separate convs take over).
int fire_stream_launch_ex(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                          const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                          int pool, hipStream_t st, bool* handled);

int fire_stream_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                       const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                       hipStream_t st, bool* handled) {
  return fire_stream_launch_ex(x, ws, bs, w1, b1, w3, b3, y, n, h, w, cin, s, e1, e3, dtype, 0, st, handled);
}

static int fire_stream_launch_full(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                                   const float* b3, void* sq_keep, void* y, int n, int h, int w, int cin, int s, int e1, int e3,
                                   int dtype, int pool, hipStream_t st, bool* handled);

// the whole module with its squeeze tensor written as well (sq_out != NULL; the unpooled form)
int fire_stream_launch_keep(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                            const float* b3, void* sq_out, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                            hipStream_t st, bool* handled) {
  return fire_stream_launch_full(x, ws, bs, w1, b1, w3, b3, sq_out, y, n, h, w, cin, s, e1, e3, dtype, 0, st, handled);
}

// pool != 0: the module is followed by max_pool 3x3 / stride 2 / SAME and y is the POOLED tensor [n, ceil(h/2), ceil(w/2), e1+e3]
int fire_stream_launch_ex(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                          const float* b3, void* y, int n, int h, int w, int cin, int s, int e1, int e3, int dtype,
                          int pool, hipStream_t st, bool* handled) {
  return fire_stream_launch_full(x, ws, bs, w1, b1, w3, b3, nullptr, y, n, h, w, cin, s, e1, e3, dtype, pool, st, handled);
}

static int fire_stream_launch_full(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                                   const float* b3, void* sq_keep, void* y, int n, int h, int w, int cin, int s, int e1, int e3,
                                   int dtype, int pool, hipStream_t st, bool* handled) {
  *handled = false;
  int nchx, nts, nwaves;
  if (!stream_shape(cin, s, e1, e3, dtype, &nchx, &nts, &nwaves)) return SQDET_OK;
  FireSArgs a;
  a.x = x; a.y = y; a.ws = ws; a.w1 = w1; a.w3 = w3; a.bs = bs; a.b1 = b1; a.b3 = b3;
  a.ws2 = nullptr; a.bs2 = nullptr; a.s_out = nullptr; a.S2 = 0; a.sq_keep = pool ? nullptr : sq_keep;
  a.N = n; a.H = h; a.W = w; a.Cin = cin; a.S = s; a.E = e1;
  a.Hp = out_size(h, 3, 2, SQDET_PAD_SAME); a.Wp = out_size(w, 3, 2, SQDET_PAD_SAME);
  a.ptp = pad_before(h, 3, 2, SQDET_PAD_SAME); a.plp = pad_before(w, 3, 2, SQDET_PAD_SAME);
  if (pool) { a.tiles_x = (a.Wp + 6) / 7; a.tiles_y = (a.Hp + 3) / 4; }
  else { a.tiles_x = (w + SCOLS - 1) / SCOLS; a.tiles_y = (h + 7) / 8; }
  const long nt = (long)n * a.tiles_x * a.tiles_y;
  if (nt > 0x3fffffffL) return SQDET_OK;
  a.ntiles = (int)nt;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  a.x_pieces = cin * esz / 16;
  const long xb = (long)n * h * w * cin * esz, yb = pool ? (long)n * a.Hp * a.Wp * 2 * e1 * esz : (long)n * h * w * 2 * e1 * esz;
  if (xb >= (1L << 31) || yb >= (1L << 31)) return SQDET_OK;   // 32-bit buffer offsets
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
  if (tune(TUNE_DBG) == 40) a.x_bytes = 16;   // EXPERIMENT: every input load out of range (no read traffic)
  if (tune(TUNE_DBG) == 41) a.y_bytes = 16;   // EXPERIMENT: every store dropped (no write traffic)
  bool ok;
  if (pool) ok = dtype == SQDET_F16 ? dispatch_stream<f16, true>(a, nchx, nts, nwaves, st) : dispatch_stream<float, true>(a, nchx, nts, nwaves, st);
  else ok = dtype == SQDET_F16 ? dispatch_stream<f16, false>(a, nchx, nts, nwaves, st) : dispatch_stream<float, false>(a, nchx, nts, nwaves, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

// ---- whole fire module from x + the NEXT module's squeeze (float16): fire2 -> fire3's squeeze, fire4 -> fire5's ----
// (instantiated for SqueezeDet's two such pairs: Cin 64 / S 16 / E 64 / S2 16 and Cin 128 / S 32 / E 128 / S2 32)
bool fire_squeeze_next_eligible(int cin, int s, int e1, int e3, int s2, int dtype) {
  if (dtype != SQDET_F16) return false;
  int nchx, nts, nwaves;
  if (!stream_shape(cin, s, e1, e3, dtype, &nchx, &nts, &nwaves)) return false;
  return (nchx == 2 && nts == 1 && nwaves == 4 && s == 16 && s2 == 16) || (nchx == 4 && nts == 2 && nwaves == 8 && s2 == 32);
}

template <int NCHX, int NTS, int NWAVES, bool PAIR, int NTS2, bool POOL = false, bool SQIN = false>
static int launch_stream_sq(const FireSArgs& a, hipStream_t st) {
  constexpr int NQ = (NWAVES / 2 / 2) * 4;
  const size_t lds = 2 * (size_t)Geo<POOL>::TILE + (size_t)NCHX * NTS * 1024 + (size_t)(NWAVES / 2 / 2) * 4 * 1024 +
                     (size_t)NQ * (POOL ? 32 : 128) * 64 + (size_t)NQ * NTS2 * 1024 + (size_t)(2 * a.E + a.S + a.S2) * 4;
  static PerDevice once;
  SQDET_CHECK_HIP(once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&fire_stream<f16, NCHX, NTS, NWAVES, 2, POOL, 2, PAIR, SQIN, NTS2>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }));
  int grid = cu_count() * (8 / NWAVES);
  if (grid > (a.ntiles + 7) / 8 * 8) grid = (a.ntiles + 7) / 8 * 8;
  hipLaunchKernelGGL((fire_stream<f16, NCHX, NTS, NWAVES, 2, POOL, 2, PAIR, SQIN, NTS2>), dim3(grid), dim3(NWAVES * 64), lds, st, a);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

// the expand half of a module from its squeeze tensor (+ its pool) emitting the NEXT module's squeeze tensor -- SqueezeDet's
// fire2 -> fire3's squeeze (when the stem launch already produced fire2's squeeze tensor), fire3+pool3 -> fire4's squeeze,
// fire4 -> fire5's, fire5+pool5 -> fire6's
bool fire_expand_squeeze_next_eligible(int s, int e1, int e3, int s2, int pool, int dtype) {
  if (!fire_expand_stream_eligible(s, e1, e3, dtype)) return false;
  return (s == 16 && e1 == 64 && s2 == 32 && pool) || (s == 16 && e1 == 64 && s2 == 16 && !pool) ||
         (s == 32 && e1 == 128 && s2 == 32 && !pool) || (s == 32 && e1 == 128 && s2 == 48 && pool);
}

int fire_expand_squeeze_next_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3,
                                    const void* ws2, const float* bs2, void* s_out, int n, int h, int w, int s, int e1, int e3,
                                    int s2, int pool, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (!fire_expand_squeeze_next_eligible(s, e1, e3, s2, pool, dtype)) return SQDET_OK;
  {   // the DMA-fed four-waves-per-SIMD kernel (fire3.hip) takes SqueezeDet's four shapes ("dbg" 70: the forms below)
    const int rc = fire_dma_launch(sq_in, w1, b1, w3, b3, ws2, bs2, s_out, n, h, w, s, e1, e3, s2, pool, dtype, st, handled);
    if (rc != SQDET_OK || *handled) return rc;
  }
  FireSArgs a;
  a.x = sq_in; a.y = nullptr; a.ws = nullptr; a.w1 = w1; a.w3 = w3; a.bs = nullptr; a.b1 = b1; a.b3 = b3;
  a.ws2 = ws2; a.bs2 = bs2; a.s_out = s_out; a.S2 = s2; a.sq_keep = nullptr;
  a.N = n; a.H = h; a.W = w; a.Cin = s; a.S = s; a.E = e1;
  a.Hp = out_size(h, 3, 2, SQDET_PAD_SAME); a.Wp = out_size(w, 3, 2, SQDET_PAD_SAME);
  a.ptp = pad_before(h, 3, 2, SQDET_PAD_SAME); a.plp = pad_before(w, 3, 2, SQDET_PAD_SAME);
  if (pool) { a.tiles_x = (a.Wp + 6) / 7; a.tiles_y = (a.Hp + 3) / 4; }
  else { a.tiles_x = (w + SCOLS - 1) / SCOLS; a.tiles_y = (h + 7) / 8; }
  const long nt = (long)n * a.tiles_x * a.tiles_y;
  if (nt > 0x3fffffffL) return SQDET_OK;
  a.ntiles = (int)nt;
  a.x_pieces = s * 2 / 16;
  const long xb = (long)n * h * w * s * 2;
  if (xb >= (1L << 31)) return SQDET_OK;
  a.x_bytes = (unsigned)xb; a.y_bytes = 0;
  int rc;
  if (s == 16 && !pool) rc = launch_stream_sq<1, 1, 4, true, 1, false, true>(a, st);   // fire2 (from the stem's squeeze tensor) -> fire3's squeeze
  else if (s == 16) rc = launch_stream_sq<1, 1, 4, true, 2, true, true>(a, st);
  else if (!pool) rc = launch_stream_sq<1, 1, 8, false, 2, false, true>(a, st);
  else rc = launch_stream_sq<1, 1, 8, false, 3, true, true>(a, st);
  if (rc != SQDET_OK) return rc;
  *handled = true;
  return SQDET_OK;
}

int fire_squeeze_next_launch(const void* x, const void* ws, const float* bs, const void* w1, const float* b1, const void* w3,
                             const float* b3, const void* ws2, const float* bs2, void* s_out, int n, int h, int w, int cin,
                             int s, int e1, int e3, int s2, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (!fire_squeeze_next_eligible(cin, s, e1, e3, s2, dtype)) return SQDET_OK;
  FireSArgs a;
  a.x = x; a.y = nullptr; a.ws = ws; a.w1 = w1; a.w3 = w3; a.bs = bs; a.b1 = b1; a.b3 = b3;
  a.ws2 = ws2; a.bs2 = bs2; a.s_out = s_out; a.S2 = s2; a.sq_keep = nullptr;
  a.N = n; a.H = h; a.W = w; a.Cin = cin; a.S = s; a.E = e1;
  a.Hp = a.Wp = a.ptp = a.plp = 0;
  a.tiles_x = (w + SCOLS - 1) / SCOLS; a.tiles_y = (h + 7) / 8;
  const long nt = (long)n * a.tiles_x * a.tiles_y;
  if (nt > 0x3fffffffL) return SQDET_OK;
  a.ntiles = (int)nt;
  a.x_pieces = cin * 2 / 16;
  const long xb = (long)n * h * w * cin * 2;
  if (xb >= (1L << 31)) return SQDET_OK;
  a.x_bytes = (unsigned)xb; a.y_bytes = 0;
  const int rc = s == 16 ? launch_stream_sq<2, 1, 4, true, 1>(a, st) : launch_stream_sq<4, 2, 8, false, 2>(a, st);
  if (rc != SQDET_OK) return rc;
  *handled = true;
  return SQDET_OK;
}

// ---- the expand half of a fire module from its SQUEEZE tensor (float16; the chain kernels of chain.hip produce it) ----
bool fire_expand_stream_eligible(int s, int e1, int e3, int dtype) {
  if (conv_algo() != 0 || e1 != e3 || dtype != SQDET_F16) return false;
  const ConvGeom g1 = conv_geom(1, s, e1, dtype), g3 = conv_geom(3, s, e3, dtype);
  if (g1.gather || g3.gather || g1.nchunk != 1 || g3.nchunk != 1 || g1.nt != 4 || g3.nt != 4) return false;
  if (e1 % 64 != 0 || (g1.ngroups != 1 && g1.ngroups != 2)) return false;
  return (s * 2) % 16 == 0 && s % 4 == 0;
}

// y = concat(relu(conv1x1(sq_in)), relu(conv3x3(sq_in))), or its 3x3/s2 SAME max-pool when pool != 0
int fire_expand_stream_launch(const void* sq_in, const void* w1, const float* b1, const void* w3, const float* b3, void* y,
                              int n, int h, int w, int s, int e1, int e3, int dtype, int pool, hipStream_t st, bool* handled) {
  *handled = false;
  if (!fire_expand_stream_eligible(s, e1, e3, dtype)) return SQDET_OK;
  FireSArgs a;
  a.x = sq_in; a.y = y; a.ws = nullptr; a.w1 = w1; a.w3 = w3; a.bs = nullptr; a.b1 = b1; a.b3 = b3;
  a.ws2 = nullptr; a.bs2 = nullptr; a.s_out = nullptr; a.S2 = 0; a.sq_keep = nullptr;
  a.N = n; a.H = h; a.W = w; a.Cin = s; a.S = s; a.E = e1;
  a.Hp = out_size(h, 3, 2, SQDET_PAD_SAME); a.Wp = out_size(w, 3, 2, SQDET_PAD_SAME);
  a.ptp = pad_before(h, 3, 2, SQDET_PAD_SAME); a.plp = pad_before(w, 3, 2, SQDET_PAD_SAME);
  if (pool) { a.tiles_x = (a.Wp + 6) / 7; a.tiles_y = (a.Hp + 3) / 4; }
  else { a.tiles_x = (w + SCOLS - 1) / SCOLS; a.tiles_y = (h + 7) / 8; }
  const long nt = (long)n * a.tiles_x * a.tiles_y;
  if (nt > 0x3fffffffL) return SQDET_OK;
  a.ntiles = (int)nt;
  a.x_pieces = s * 2 / 16;
  const long xb = (long)n * h * w * s * 2, yb = pool ? (long)n * a.Hp * a.Wp * 2 * e1 * 2 : (long)n * h * w * 2 * e1 * 2;
  if (xb >= (1L << 31) || yb >= (1L << 31)) return SQDET_OK;
  a.x_bytes = (unsigned)xb; a.y_bytes = (unsigned)yb;
  const int nwaves = 4 * (e1 / 64);
  if (pool) {
    if (nwaves == 4) launch_stream<f16, 1, 1, 4, true, 2, true>(a, st);
    else launch_stream<f16, 1, 1, 8, true, 2, true>(a, st);
  } else {
    if (nwaves == 4) launch_stream<f16, 1, 1, 4, false, 2, true>(a, st);
    else launch_stream<f16, 1, 1, 8, false, 2, true>(a, st);
  }
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_FIRE_TIMING
extern "C" int sqdet_debug_fire_timing(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_fire_timing), sizeof(unsigned long long) * count);
}
#endif
