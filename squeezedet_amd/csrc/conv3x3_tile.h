// 3x3 / stride-1 / SAME convolution with an LDS-staged halo tile (the expand3x3 layers of the
// fire modules and the ConvDet head; reference src/nn_skeleton.py:471-563 via
// nets/squeezeDet.py:103-105 and :76-79).
//
// One 256-thread workgroup owns an 8-row x 16-column output tile of one image.
//   * The (8+2) x (16+2) input halo tile is staged ONCE into LDS (zero-filled outside the image =
//     TF SAME padding), laid out [64-byte K-chunk][pixel][4 x 16 B] with the 16-byte slot of lane
//     group g stored at  g ^ ((pixel>>1)&3).  With that XOR every ds_read_b128 of a B fragment
//     (16 consecutive pixels of one row, 16 B each) is bank-conflict free for any alignment of the
//     first pixel (checked exhaustively against the ds_read_b128 lane-group model of
//     MI355X_MICROARCH.md section LDS).  All 9 taps then read LDS: no re-read of the input through L1/L2.
//   * Weights stay the MFMA A operand, read as 1-KiB fragments straight from global (L1/L2).
//   * COUT-split mode (expand3x3): a wave owns MT tile rows x one whole packed cout group
//     (NTW = 4..6 tiles -> 32..48 contiguous output bytes per lane); the 4 waves are laid out
//     WR = 8/MT along the rows x WC = 4/WR along the cout groups.
//   * The ConvDet head (Cin = 768, 72 couts = 5 tiles) uses the same tile and LDS layout with K split over the waves:
//     convdet.hip.
#pragma once
#include <type_traits>

#include "conv_common.h"

namespace sqdet {

constexpr int TROWS = 8;                      // output rows per tile
constexpr int TCOLS = 16;                     // output cols per tile (one MFMA pixel block per row)
constexpr int HP = (TROWS + 2) * (TCOLS + 2); // halo pixels = 180
constexpr int CHUNK_BYTES = HP * 64;          // one 64-byte K-chunk of the halo tile = 11520 B

struct TileArgs {
  ConvArgs c;
  int tiles_x, tiles_y;
  int nt_pack;      // tiles per packed group
  int total_tiles;  // cout tiles over all groups
  int nchunk;       // 64-byte K chunks per pixel (ceil)
  int pieces;       // 16-byte pieces per pixel actually present = Cin*sizeof(T)/16
  unsigned x_bytes; // size of the input tensor (split-K mode: 32-bit buffer offsets)
  int stage_chunks; // cout-split mode: K-chunks of the halo tile resident in LDS at a time (deep K is walked in stages)
  // PAIR form (expand1x1 || expand3x3 of a fire module from ONE staged squeeze tile, nets/squeezeDetPlus.py:81-106): the packed 1x1
  // kernel / bias of the same Cin -> Cout shape and the channel offset of its slice of y's rows; the whole tile is resident
  const void* wp1;
  const float* bias1;
  int y_coffset1;
  // staging: dma != 0: the halo tile arrives by LDS-DMA (stage_tile_dma), chunk pitch 12288 B = 192 pixels (whole 1-KiB DMA blocks);
  // else through registers (stage_tile), chunk pitch CHUNK_BYTES
  int dma, chunk_pitch;
};

// Stage `nload` K-chunks starting at chunk c0 of the halo tile into LDS.
template <typename T>
__device__ __forceinline__ void stage_tile(const TileArgs& a, unsigned char* lds, int n, int oy0, int ox0, int c0,
                                           int nload) {
  const int CP = a.chunk_pitch;
  // the input may be a channel slice [x_coffset, x_coffset + Cin) of rows x_cstride channels wide (backward-data of a fire
  // module reads the expand3x3 half of dY)
  const unsigned char* x = reinterpret_cast<const unsigned char*>(a.c.x) + (size_t)a.c.x_coffset * sizeof(T);
  const int row_bytes = a.c.x_cstride * (int)sizeof(T);  // bytes per pixel in global memory
  const int ppc = nload * 4;            // pieces per pixel in this stage
  const int total = HP * ppc;
  // batches of 4 loads in flight per thread before the LDS stores (a load->store loop would pay the
  // global latency once per iteration)
  for (int base = threadIdx.x; base < total; base += 256 * 4) {
    i32x4 v[4];
    int off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * 256;
      const int P = idx / ppc;
      const int q = idx - P * ppc;          // piece within the staged range
      const int gq = c0 * 4 + q;            // piece within the pixel
      const int r = P / (TCOLS + 2), cc = P - r * (TCOLS + 2);
      const int iy = oy0 - 1 + r, ix = ox0 - 1 + cc;
      v[u] = i32x4{0, 0, 0, 0};
      if (idx < total && gq < a.pieces && iy >= 0 && iy < a.c.H && ix >= 0 && ix < a.c.W)
        v[u] = *reinterpret_cast<const i32x4*>(x + (((size_t)n * a.c.H + iy) * a.c.W + ix) * row_bytes + gq * 16);
      const int c = q >> 2, g = q & 3;
      off[u] = idx < total ? c * CP + P * 64 + ((g ^ ((P >> 1) & 3)) << 4) : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (off[u] >= 0) *reinterpret_cast<i32x4*>(lds + off[u]) = v[u];
  }
}

// The same stage by LDS-DMA (round 6): `buffer_load_dwordx4 ... lds` moves 16 pixels x 64 B = one 1-KiB block of a chunk straight into
// LDS -- no staging registers, no ds_write pass, and ALL of a stage's blocks are in flight at once: the register path fetched 4 pieces
// per thread at a time, i.e. a 6-chunk stage (69 KB) was four to five serialized global round trips, and two co-resident workgroups
// that start together stage together (nothing computes meanwhile).  Block (chunk q, pixel block i): lane l = (pixel 16 i + l / 4,
// LDS slot l & 3) requests the 16-byte piece whose slot it fills (the XOR of stage_tile applied on the SOURCE side); pixels outside the
// image, pieces beyond Cin and the 12 pitch pixels are out-of-range offsets (bit 31) = zeros.  Wave w takes blocks w, w + 4, ...: three
// distinct pixel blocks (12 = 3 x 4), whose pixel offsets are computed once per stage.  The caller waits (vmcnt(0)) before the barrier.
__device__ __forceinline__ void tile_dma16(unsigned voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
template <typename T>
__device__ __forceinline__ void stage_tile_dma(const TileArgs& a, unsigned lds_addr, const i32x4& rx, int n, int oy0, int ox0, int c0,
                                               int nload, int wave, int lane) {
  const int row_bytes = a.c.x_cstride * (int)sizeof(T);
  unsigned rel[3];
  int pc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int P = (wave + 4 * k) * 16 + (lane >> 2);
    const int r = P / (TCOLS + 2), cc = P - r * (TCOLS + 2);
    const int iy = oy0 - 1 + r, ix = ox0 - 1 + cc;
    const bool ok = P < HP && iy >= 0 && iy < a.c.H && ix >= 0 && ix < a.c.W;
    rel[k] = ok ? (unsigned)(((n * a.c.H + iy) * a.c.W + ix) * row_bytes) : 0x80000000u;
    pc[k] = (lane & 3) ^ ((P >> 1) & 3);
  }
  for (int q = 0; q < nload; ++q) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int gq = (c0 + q) * 4 + pc[k];
      const unsigned off = gq < a.pieces ? rel[k] + (unsigned)(gq * 16) : 0x80000000u;    // (rel's bit 31 survives the add: offsets < 2^31)
      tile_dma16(off | (rel[k] & 0x80000000u), rx, lds_addr + (unsigned)(q * 12288 + (wave + 4 * k) * 1024));
    }
  }
}

// PAIR (round 6): behind the 3x3 conv the workgroup runs the module's expand1x1 on the SAME resident tile -- the centre tap's B
// fragments, nchunk steps instead of 9 nchunk, the accumulators and fragment registers the 3x3 pass has just left -- and writes it
// into its slice of the concat rows: one read and one staging of the squeeze tile for both expands, one launch instead of two.
// The 1x1 weights (6 chunks per group of loads) are requested before the 3x3 epilogue's stores.  Accumulation order = chunk
// ascending, as every 1x1 kernel: bitwise the separate conv.
// -DSQDET_C3_TIMELINE (experiments only, tools/c3_timeline.py): 100 MHz s_memrealtime stamps per workgroup -- entry, first stage requested,
// first stage landed, K loop done, epilogue issued -- plus the cumulated wait for the later stages' tiles
#ifdef SQDET_C3_TIMELINE
__device__ unsigned long long g_c3_tl[16384 * 8];
#define C3TL(k) do { c3tl[k] = wall_clock64(); } while (0)
#else
#define C3TL(k) do {} while (0)
#endif

// (the <8, 3> float16 form -- SqueezeDet+ fire6 / fire7 -- runs three workgroups per CU: its register budget is 168; checked spill-free)
template <typename T, int MT, int NTW, bool PAIR = false>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && MT == 8 && NTW == 3) ? 3 : 1) void conv3x3_tile(TileArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
#ifdef SQDET_C3_TIMELINE
  unsigned long long c3tl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  C3TL(0);
  constexpr int WR = TROWS / MT;   // waves along the tile rows
  constexpr int WC = 4 / WR;       // waves along the cout groups
  static_assert(WR * MT == TROWS && WR * WC == 4, "bad wave layout");
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;

  // XCD-aware tile order: workgroup b runs on XCD b % 8 (own L2 each); every XCD gets a contiguous band of
  // tiles so the halos shared by neighbouring tiles are fetched into ONE L2 instead of up to eight.
  int b = (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3));   // gridDim.x is a multiple of 8
  if (b >= a.c.N * a.tiles_x * a.tiles_y) return;
  const int tx = b % a.tiles_x; b /= a.tiles_x;
  const int ty = b % a.tiles_y;
  const int n = b / a.tiles_y;
  const int oy0 = ty * TROWS, ox0 = tx * TCOLS;

  // rows and cout tiles this wave owns
  const int m0 = (wave % WR) * MT;
  const int tile0 = (blockIdx.y * WC + wave / WR) * NTW;
  const bool active = tile0 < a.total_tiles;
  const int group = tile0 / a.nt_pack;
  const int n0 = tile0 - group * a.nt_pack;

  f32x4 acc[MT][NTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const i32x4* wbase = reinterpret_cast<const i32x4*>(a.c.wp) + ((size_t)group * a.c.steps * a.nt_pack + n0) * 64 + lane;
  const int nstages = (a.nchunk + a.stage_chunks - 1) / a.stage_chunks;
  const int CP = a.chunk_pitch;
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // (DMA staging: the input as a buffer resource -- channel slice offset folded into the base, 32-bit offsets: checked by the launcher)
  const unsigned long long xaddr = (unsigned long long)(uintptr_t)a.c.x + (unsigned long long)a.c.x_coffset * sizeof(T);
  // (readfirstlane: the descriptor must sit in SGPRs for the DMA's asm operand whatever the compiler thinks of its uniformity)
  const i32x4 rx = {__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr), __builtin_amdgcn_readfirstlane((int)(unsigned)((xaddr >> 32) & 0xffffu)),
                    __builtin_amdgcn_readfirstlane((int)a.x_bytes), 0x00020000};

  // The bias (both biases in the PAIR form) lives in LDS behind the tile, [total_tiles * 16] floats, zero beyond Cout (and all zero without
  // a bias): 4-byte LDS-DMA pieces (256 B per wave instruction, out-of-range = zeros) issued with the first stage's blocks -- no
  // registers.  (Loaded in the epilogue it cost every workgroup an exposed global round trip: tools/c3_timeline.py.)
  const int ncb = a.total_tiles * 16;
  float* const lbias = reinterpret_cast<float*>(lds + a.stage_chunks * CP);
  {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    auto bias_dma = [&](const float* bp, unsigned lds_off) {
      const unsigned long long ba = (unsigned long long)(uintptr_t)bp;
      const i32x4 rb = {__builtin_amdgcn_readfirstlane((int)(unsigned)ba), __builtin_amdgcn_readfirstlane((int)(unsigned)((ba >> 32) & 0xffffu)),
                        __builtin_amdgcn_readfirstlane(bp ? a.c.Cout * 4 : 0), 0x00020000};
      for (int i0 = wv * 64; i0 < ncb; i0 += 256)      // (wave-uniform)
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds" ::"v"((unsigned)(i0 + lane) * 4u), "s"(rb),
                     "s"(__builtin_amdgcn_readfirstlane((int)(lds_addr + (unsigned)(a.stage_chunks * CP) + lds_off + (unsigned)i0 * 4u))) : "memory", "m0");
    };
    bias_dma(a.c.bias, 0u);
    if constexpr (PAIR) bias_dma(a.bias1, (unsigned)ncb * 4u);
  }

  i32x4 wq[3][NTW];   // cout-split mode: weight fragments of three consecutive steps
  for (int stage = 0; stage < nstages; ++stage) {
    const int sc = a.stage_chunks;
    const int c0 = stage * sc;
    const int nload = a.nchunk - c0 < sc ? a.nchunk - c0 : sc;
    if (stage > 0) __syncthreads();  // everyone done reading the previous stage
#ifdef SQDET_C3_TIMELINE
    const unsigned long long st0 = wall_clock64();
#endif
    if (a.dma) {
      stage_tile_dma<T>(a, lds_addr, rx, n, oy0, ox0, c0, nload, __builtin_amdgcn_readfirstlane(wave), lane);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's blocks (and the weight steps in flight) have landed
    } else {
      stage_tile<T>(a, lds, n, oy0, ox0, c0, nload);
      if (stage == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the bias pieces)
    }
    if (stage == 0) C3TL(1);
    __syncthreads();
#ifdef SQDET_C3_TIMELINE
    if (stage == 0) c3tl[2] = wall_clock64(); else c3tl[5] += wall_clock64() - st0;
#endif
    if (!active) continue;

    // Flattened (chunk, tap) steps of this stage.  The weight fragments run TWO steps ahead over three statically named
    // register sets (step s uses set s % 3; a stage is a multiple of 9 = 3 x 3 steps, so the names line up across stages
    // and nothing is copied), and the look-ahead crosses the staging barriers: with a one-step look-ahead a workgroup that
    // is alone on its CU (the 22x76 / 24x78 maps at batch 8: 240 workgroups) paid the L2 latency at every step.
    const int nsteps = nload * 9;
    const int gs0 = c0 * 9;                       // global step index of this stage's first step
    const int gsl = a.nchunk * 9 - 1;             // last step of the conv (loads past it are clamped: harmless re-reads)
    auto wptr = [&](int gs) {
      const int gq = gs < gsl ? gs : gsl;
      const int ch = gq / 9, t9 = gq - ch * 9;
      return wbase + (size_t)(t9 * a.nchunk + ch) * a.nt_pack * 64;
    };
    if (stage == 0) {
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const i32x4* wp = wptr(pp);
#pragma unroll
        for (int t = 0; t < NTW; ++t) wq[pp][t] = wp[t * 64];
      }
    }
    // B fragments one step ahead as well (three sets, same rotation): the LDS latency of step s+1 hides under step s's MFMAs
    auto bread = [&](int ss, i32x4 (&bf)[MT]) {
      const int cl = ss / 9, t9 = ss - cl * 9;
      const int dy = t9 / 3, dx = t9 - dy * 3;
      // (16-byte pieces beyond Cin were zero-filled by stage_tile, so every read is unconditional)
      const unsigned char* lchunk = lds + cl * CP;
      const int P0 = (dy + m0) * (TCOLS + 2) + j + dx;   // halo pixel of this wave's first row at this tap
      const int h0 = P0 >> 1;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        // pixel P = P0 + 18*m ; (P>>1)&3 == (h0 + 9m)&3 == (h0 + m)&3
        const int slot = g ^ ((h0 + m) & 3);
        bf[m] = *reinterpret_cast<const i32x4*>(lchunk + (P0 + (TCOLS + 2) * m) * 64 + (slot << 4));
      }
    };
    // (whole-tile waves, MT = 8: 96 more registers would cost the third resident workgroup -- measured slower -- so those
    // read their fragments in the step itself)
    constexpr bool BAHEAD = MT <= 4;
    i32x4 bq[BAHEAD ? 3 : 1][MT];
    if constexpr (BAHEAD) bread(0, bq[0]);
    // One trip = three steps over the three statically named sets.  The conv's LAST trip requests no fragments beyond the final step (PF1 /
    // PF2 false): those re-reads were harmless but still in flight when the loop ended, and the epilogue -- which reuses their registers --
    // began with `s_waitcnt vmcnt(0)`: an L2 round trip in every workgroup's epilogue (tools/c3_timeline.py).
    auto trip = [&](int s, auto pf1, auto pf2) __attribute__((always_inline)) {
      constexpr bool PF[3] = {true, decltype(pf1)::value, decltype(pf2)::value};
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        if (PF[u]) {
          const i32x4* wp = wptr(gs0 + s + u + 2);
#pragma unroll
          for (int t = 0; t < NTW; ++t) wq[(u + 2) % 3][t] = wp[t * 64];
        }
        if constexpr (BAHEAD) {
          if (s + u + 1 < nsteps) bread(s + u + 1, bq[(u + 1) % 3]);
        } else {
          bread(s + u, bq[0]);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], wq[u][t], bq[BAHEAD ? u : 0][m]);
      }
    };
    const bool last_stage = stage == nstages - 1;
    const int nfull = last_stage ? nsteps - 3 : nsteps;
#pragma unroll 1
    for (int s = 0; s < nfull; s += 3) trip(s, std::true_type{}, std::true_type{});
    if (last_stage) trip(nsteps - 3, std::false_type{}, std::false_type{});
  }

  C3TL(3);
  // epilogue: lane = pixel (row m0+m, col j); couts group*16*NTp + g*4*NTp + (n0+t)*4 .. +4 : 4*NTW consecutive
  T* y = reinterpret_cast<T*>(a.c.y);
  const int ox = ox0 + j;
  const int cb = group * 16 * a.nt_pack + g * 4 * a.nt_pack + n0 * 4;
  // (PLAIN: no `y +=`, no ReLU mask -- the forward convs: one wave-uniform test instead of two exec-masked blocks per row and tile)
  // (always_inline: a lambda the inliner leaves out-of-line takes `acc` by reference, i.e. through SCRATCH -- the PAIR form's four copies did)
  auto epilogue_t = [&](const float* bias_l, int y_coffset, auto plain_t) __attribute__((always_inline)) {
  constexpr bool PLAIN = decltype(plain_t)::value;
  f32x4 bias[NTW];
  int nt_valid = 0;   // Cout is a multiple of 4: whole 4-cout pieces beyond Cout are skipped
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const bool ok = cb + t * 4 < a.c.Cout;
    bias[t] = active ? *reinterpret_cast<const f32x4*>(bias_l + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};   // (LDS: zeros beyond Cout)
    nt_valid += ok ? 1 : 0;
  }

  if (!active) return;

  if (ox < a.c.W) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int oy = oy0 + m0 + m;
      if (oy >= a.c.H) break;
      T* dst = y + (((size_t)n * a.c.H + oy) * a.c.W + ox) * a.c.y_cstride + y_coffset + cb;
      f32x4 v[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        v[t] = acc[m][t] + bias[t];
        if (!PLAIN && a.c.accum && t < nt_valid) {   // y += conv(x): d(squeeze) = dgrad_1x1(dY[:, :e1]) + dgrad_3x3(dY[:, e1:])
          v[t][0] += (float)dst[t * 4 + 0]; v[t][1] += (float)dst[t * 4 + 1];
          v[t][2] += (float)dst[t * 4 + 2]; v[t][3] += (float)dst[t * 4 + 3];
        }
        if (a.c.relu) {
          v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
          v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
        }
        if (!PLAIN && a.c.relu_of && t < nt_valid) {   // ReLU backward of the layer below (see ConvArgs)
          const T* r = reinterpret_cast<const T*>(a.c.relu_of) + (dst - y) + t * 4;
          v[t][0] = (float)r[0] > 0.f ? v[t][0] : 0.f; v[t][1] = (float)r[1] > 0.f ? v[t][1] : 0.f;
          v[t][2] = (float)r[2] > 0.f ? v[t][2] : 0.f; v[t][3] = (float)r[3] > 0.f ? v[t][3] : 0.f;
        }
      }
      store_couts<T, NTW>(dst, v, nt_valid);
    }
  }
  };
  auto epilogue = [&](const float* bias_l, int y_coffset) __attribute__((always_inline)) {
    if (!a.c.accum && !a.c.relu_of) epilogue_t(bias_l, y_coffset, std::true_type{});
    else epilogue_t(bias_l, y_coffset, std::false_type{});
  };
  if constexpr (!PAIR) {
    epilogue(lbias, a.c.y_coffset);
#ifdef SQDET_C3_TIMELINE
    C3TL(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    c3tl[6] = wall_clock64();
    if (threadIdx.x == 0) {
      const unsigned fid = blockIdx.y * gridDim.x + blockIdx.x;
      if (fid < 16384)
        for (int k = 0; k < 8; ++k) g_c3_tl[(size_t)fid * 8 + k] = c3tl[k];
    }
#endif
  } else {
    // ---- expand1x1 on the resident tile (a.stage_chunks == a.nchunk: one stage, the whole halo tile is still in LDS) ----
    constexpr int CG = 6;                    // chunks per group of weight loads (6 x NTW fragments in flight)
    const i32x4* w1base = reinterpret_cast<const i32x4*>(a.wp1) + ((size_t)group * a.nchunk * a.nt_pack + n0) * 64 + lane;
    i32x4 w1[CG][NTW];
    auto load_w1 = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < CG; ++u) {
        const int ch = c0 + u < a.nchunk ? c0 + u : a.nchunk - 1;      // (past the last chunk: a harmless re-read, never multiplied)
#pragma unroll
        for (int t = 0; t < NTW; ++t) w1[u][t] = w1base[(size_t)(ch * a.nt_pack + t) * 64];
      }
    };
    if (active) load_w1(0);                  // in flight under the 3x3 epilogue's stores
    epilogue(lbias, a.c.y_coffset);
    if (!active) return;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int P0 = (1 + m0) * (TCOLS + 2) + j + 1;   // halo pixel of this wave's first row at the centre tap
    const int h0 = P0 >> 1;
    for (int c0 = 0; c0 < a.nchunk; c0 += CG) {
      if (c0 > 0) load_w1(c0);
#pragma unroll
      for (int u = 0; u < CG; ++u) {
        if (c0 + u < a.nchunk) {             // wave-uniform
          const unsigned char* lchunk = lds + (c0 + u) * CP;
          i32x4 bf[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m)
            bf[m] = *reinterpret_cast<const i32x4*>(lchunk + (P0 + (TCOLS + 2) * m) * 64 + ((g ^ ((h0 + m) & 3)) << 4));
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], w1[u][t], bf[m]);
        }
      }
    }
    epilogue(lbias + ncb, a.y_coffset1);
  }
}

template <typename T, int MT, int NTW, bool PAIR = false>
static void launch_tile(const TileArgs& a, int grid_y, size_t lds, hipStream_t st) {
  static PerDevice once;   // > 64 KiB of dynamic LDS has to be allowed once per kernel and device
  if (lds > 65536)
    (void)once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_tile<T, MT, NTW, PAIR>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  const dim3 grid((unsigned)((a.c.N * a.tiles_x * a.tiles_y + 7) / 8 * 8), (unsigned)grid_y);
  hipLaunchKernelGGL((conv3x3_tile<T, MT, NTW, PAIR>), grid, dim3(256), lds, st, a);
}

// the split-K (ConvDet) kernel lives in convdet.hip: that file is compiled with the accumulators in AGPRs
int convdet_tile_launch(const TileArgs& a, int dtype, hipStream_t st);

}  // namespace sqdet
