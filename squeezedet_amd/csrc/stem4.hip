// Fused stem, PHASE form, fp16 / 3x3 / 64 couts (reference src/nets/squeezeDet.py:40-44: conv1 3x3/s2 + bias + ReLU, then
// pool1 3x3/s2 SAME; with fire2's squeeze1x1 behind it in the squeeze form).  Same persistent frame as stem3.hip's stem_pers
// (XCD-banded tiles, the next tile's input patch prefetched into registers, dword im2col gather, one MFMA per 16 pixels x 16
// couts, vertical max on the raw float32 accumulators) with the pooling re-thought:
//
//   stem_pers maps the 16 lanes of an MFMA pixel block to 16 ADJACENT conv columns, so a pooled pixel's three columns sit in
//   three lanes: two DPP row shifts + two packed max per register, and only 7 of the 16 lanes end up holding a pooled pixel
//   (strips overlap by two conv columns).  Counted per pooled pixel that is ~12 VALU instructions against ~1.1 MFMA: the
//   launch is bound by its VALU stream (4 waves per SIMD issue ~450 VALU + 44 MFMA per tile: 58 us, 0.23 of HBM peak).
//
//   Here lane j IS pooled column px0 + 16*wave + j, and the conv is evaluated in three column PHASES dx = 0, 1, 2: phase dx
//   computes conv column 2*p + dx - pad of every lane's pooling window (a gather with a 24-byte lane stride instead of 12;
//   the phases differ by an immediate offset).  The 3 x 3 window maximum is then lane-local: v_max3 over the three conv rows
//   of a phase, a running max over the phases -- no cross-lane traffic, all 16 lanes useful, no strip overlap (tile = 4 x 64
//   pooled pixels).  Price: window columns shared by neighbouring pooled pixels are computed twice, 1.5 MFMA per pooled
//   pixel instead of 1.1 -- the matrix pipe was 17 % busy.  Per pooled pixel: ~7 VALU + 1.6 MFMA.
//
// Arithmetic per element is stem_pers's (same K order inside the MFMA, max before bias before the one float16 rounding):
// results are bitwise equal to stem_pers's (tools/exp_stem4.py, tests/test_gpu_ops.py).
// THE DEFAULT on images >= 523 wide since the end of round 3 ("stem_algo" 3 = stem_pers).  When it was written the 32-image step
// gained nothing from it (0.5402 against 0.5397 ms, the launch alone 59.1 us against 62.5); with the post-processing riding in
// the chain launches and the leaner streaming launches behind it, six alternating runs on two boxes read 0.5254 against
// 0.5305 ms (-1 %).
// Measured (batch 32, 375x1242, squeeze form, same box, the launch alone): 56.5 us against stem_pers's 61.1; ladder: no stores 52.3, no loads
// 46.4, neither 43.0 (stem_pers 53.0 / 49.1 / 46.7).  Two workgroups per CU: the three phases' "previous conv row"
// accumulators (48 registers) + one set in flight + the running maximum + 32 prefetch registers need ~250; the 168-register
// build (three per CU) spills the PREFETCHED PATCH to scratch and takes 140 us, a 2-pooled-row tile (20 prefetch registers)
// still spills 72 and is slower at two per CU (62.8 us); dropping the canonicalising v_max x,x,x in front of every fmaxf
// (-fno-honor-nans: 148 -> 84 max instructions per pooled row) changed nothing -- at 8 waves per CU the kernel is bound by the
// latency of its gather -> MFMA -> max chains, not by instruction count.  Next step if this launch is revisited: the patch
// prefetch as LDS-DMA into a second LDS buffer (no prefetch registers) with 2-row tiles, which fits three workgroups per CU.
// INPUT CONTRACT: finite pixels.  This file is compiled with -fno-honor-nans (build.py: without it every two-operand max on an MFMA
// result is preceded by a canonicalising v_max x, x, x -- 148 instead of 84 max instructions per pooled row), so the pooling chain is
// undefined for NaN inputs and may return a finite value where conv -> pool (and stem_pers / stem_strip, stem_algo = 3 / 2) would
// propagate the NaN.  Images are uint8 - mean in every caller (sqdet_preprocess_bgr, demo.py:186-190); a caller that can produce NaN /
// Inf pixels selects stem_algo = 3.
#include <type_traits>
#include "stem.h"

namespace sqdet {
namespace {

#ifndef SQDET_STEM4_PPR
#define SQDET_STEM4_PPR 4
#endif
constexpr int PPR = SQDET_STEM4_PPR;      // pooled rows per tile
constexpr int PPC = 64;                   // pooled columns per tile (16 per wave, one per lane)
constexpr int PCR = 2 * PPR + 1;          // conv rows under the tile (9)
constexpr int PTR = 2 * (PCR - 1) + 3;    // staged input rows (19)
constexpr int PIC = 4 * PPC + 3;          // input columns under the tile (259)
constexpr int PRP = (PIC * 6 + 15) / 16;  // 16-byte pieces fetched per row (98)
constexpr int PPB = PRP * 16;             // LDS row pitch in bytes (1568: the gather reads are at most 3-way, on average 2.25-way
                                          // bank-conflicted -- brute-forced; pitches that are not multiples of 16 reach 2.0)
constexpr int PPCS = PTR * PRP;           // pieces per tile (1862)
constexpr int PNIT = (PPCS + 255) / 256;  // fetch rounds per thread (8)
constexpr int PBIAS = PTR * PPB;          // LDS offset of the 64 float32 biases
constexpr int PSQW = PBIAS + 256;         // squeeze form: the next layer's 2 KiB of squeeze1x1 fragments + 16 biases
constexpr int PLDS = PSQW + 2048 + 64;
constexpr int POOB = (int)0x80000000;

__device__ __forceinline__ unsigned int pkmax16(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// (followed by two idle cycles where the result feeds an MFMA directly: the wait states between a VALU write and a matrix
// instruction reading it are only inserted for instructions the compiler emits itself)
__device__ __forceinline__ unsigned int pkmax16_then_idle(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// NOT inline asm: the operands are MFMA results (see stem3.hip)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

// WPC = workgroups per CU the register budget is set for (2: 256 registers, nothing spilled)
template <bool SQ, int WPC>
__global__ __launch_bounds__(256, WPC) void stem_phase(StemArgs a, int ntiles, int per_xcd, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;

  // ---- once per workgroup: A fragments in the dword-gather K order (exactly stem_pers's), biases to LDS ----
  i32x4 af[4];
  {
    const f16* wsrc = reinterpret_cast<const f16*>(a.wp);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f16x8 v;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int i = s >> 1, h = s & 1;
        const int dy = g < 3 ? g : i;
        const int e = g < 3 ? (i < 3 ? 2 * i : 8) + h : 6 + h;
        const bool ok = g < 3 ? !(i == 3 && h == 1) : i < 3;
        const int kq = ok ? dy * 9 + e : 0;
        const int co = (t >> 1) * 32 + 8 * (j >> 2) + (t & 1) * 4 + (j & 3);
        const int pt = (co >> 2) & 3, pm = ((co >> 4) << 2) | (co & 3);
        const f16 wv = wsrc[((pt * 64) + (kq >> 3) * 16 + pm) * 8 + (kq & 7)];
        v[s] = ok ? wv : (f16)0;
      }
      af[t] = __builtin_bit_cast(i32x4, v);
    }
  }
  if (tid < 64) reinterpret_cast<float*>(lds + PBIAS)[tid] = a.bias[tid];
  if constexpr (SQ) {
    if (tid < 128) reinterpret_cast<i32x4*>(lds + PSQW)[tid] = reinterpret_cast<const i32x4*>(a.ws2)[tid];
    if (tid < 16) reinterpret_cast<float*>(lds + PSQW + 2048)[tid] = a.bs2[tid];
  }
  const int pl = 16 * wave + j;                                       // pooled column of this lane within the tile
  // gather addresses (bytes, patch row 0, phase 0): the lane's window starts at conv column 2*pl = byte 24*pl; groups 0..2
  // read dwords 0,1,2,4 of patch row dy = g, group 3 dword 3 of rows 0,1,2 (its 4th dword is masked to zero)
  const int A0 = (g < 3 ? g * PPB : 12) + 24 * pl;
  const int D1 = g < 3 ? 4 : PPB;
  const int A1 = A0 + D1, A2 = A0 + 2 * D1, A3 = g < 3 ? A0 + 16 : A0;
  const unsigned int M3 = g < 3 ? 0x0000ffffu : 0u;
  // piece it of this thread: p = it*256 + tid -> (patch row, 16-byte column); LDS offset row*PPB + c16*16 = p*16 (PPB = PRP*16):
  // nothing to keep in registers
  auto dst_of = [&](int it) { const int p = it * 256 + tid; return p < PPCS ? p * 16 : -1; };
  const int cb = g * 8;                                               // this lane's couts: cb..cb+7 and 32+cb..32+cb+7
  const float NEGF = __uint_as_float(0xff800000u);
  const unsigned int img_bytes = (unsigned int)a.H * a.W * 6;
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      SQ ? a.s_out : a.y, 0, (unsigned int)((size_t)a.N * a.Hp * a.Wp * (SQ ? 16 : a.y_cstride) * 2), 0x00020000);

  struct Tile { int n, ty, tx; bool live; };
  const int dtx = G8 % a.tiles_x, dty = (G8 / a.tiles_x) % a.tiles_y, dn = (G8 / a.tiles_x) / a.tiles_y;
  int slot = slot0;
  Tile cur;
  {
    const int ti = xcd * per_xcd + slot;
    cur.live = slot < per_xcd && ti < ntiles;
    cur.tx = ti % a.tiles_x;
    const int q = ti / a.tiles_x;
    cur.ty = q % a.tiles_y;
    cur.n = q / a.tiles_y;
  }
  auto advance = [&]() {
    slot += G8;
    cur.live = slot < per_xcd && xcd * per_xcd + slot < ntiles;
    cur.tx += dtx;
    int c = 0;
    if (cur.tx >= a.tiles_x) { cur.tx -= a.tiles_x; c = 1; }
    cur.ty += dty + c;
    c = 0;
    if (cur.ty >= a.tiles_y) { cur.ty -= a.tiles_y; c = 1; }
    cur.n += dn + c;
  };
  i32x4 pf[PNIT];
  auto fetch = [&]() {
    const int cy0 = 2 * (cur.ty * PPR) - a.ptp, cx0 = 2 * (cur.tx * PPC) - a.plp;
    const int iy0 = 2 * cy0 - a.ptc, ix0 = 2 * cx0 - a.plc;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f16*>(reinterpret_cast<const f16*>(a.x)) + (size_t)cur.n * a.H * a.W * 3, 0, img_bytes, 0x00020000);
    const int toff = iy0 * a.W * 6 + ix0 * 6;
    int src_off[PNIT];
#pragma unroll
    for (int it = 0; it < PNIT; ++it) {
      const int d = dst_of(it);
      const int row = (int)(__umulhi((unsigned)d, 2739202u) );   // d / 1568 for 0 <= d < 2^15 (2739202 = ceil(2^32 / 1568))
      src_off[it] = d < 0 ? POOB : row * (a.W * 6 - PPB) + d;     // row * W*6 + c16*16
    }
    if (ix0 < 0) {
      // left-edge tiles: a piece of image row 0 starts at a NEGATIVE offset and ends inside the image; fetch by dword
#pragma unroll
      for (int it = 0; it < PNIT; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int off = src_off[it] + toff + 4 * u;
          pf[it][u] = __builtin_amdgcn_raw_buffer_load_b32(rx, src_off[it] == POOB || off < 0 ? POOB : off, 0, 0);
        }
    } else {
#pragma unroll
      for (int it = 0; it < PNIT; ++it)
        pf[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, src_off[it] == POOB ? POOB : src_off[it] + toff, 0, 0);
    }
  };
  if (cur.live && !(dbg & 2)) fetch();
#pragma unroll
  for (int it = 0; it < PNIT; ++it) asm volatile("" : "+v"(pf[it]));

  while (cur.live) {
    const int py0 = cur.ty * PPR, px0 = cur.tx * PPC;
    const int cy0 = 2 * py0 - a.ptp, cx0 = 2 * px0 - a.plp;
    const int ix0 = 2 * cx0 - a.plc;
    const int cn = cur.n;
    // ---- the fetched patch -> LDS; columns outside the image are zero padding (only tiles on the left / right edge) ----
    if (ix0 < 0 || ix0 * 6 + PRP * 16 > a.W * 6) {
      const int lo = -ix0 * 3 / 2, hi = (a.W - ix0) * 3 / 2;         // valid dwords of a patch row (ix0, W even)
#pragma unroll
      for (int it = 0; it < PNIT; ++it) {
        const int d = dst_of(it);
        const int pdw = (d < 0 ? 0 : d - (int)__umulhi((unsigned)d, 2739202u) * PPB) >> 2;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (pdw + u < lo || pdw + u >= hi) pf[it][u] = 0;
      }
    }
#pragma unroll
    for (int it = 0; it < PNIT; ++it)
      if (dst_of(it) >= 0) *reinterpret_cast<i32x4*>(lds + dst_of(it)) = pf[it];
    __syncthreads();
    advance();
    if (cur.live && !(dbg & 2)) fetch();

    // ---- compute: 9 conv rows x 3 phases -> 4 pooled rows ----
    const int cxl = cx0 + 2 * pl;                                    // conv column of phase 0
    const bool edge_cols = cx0 < 0 || cx0 + 2 * PPC + 1 > a.Wc;      // workgroup-uniform
    const int px = px0 + pl;
    const bool store_lane = px < a.Wp;
    auto compute = [&](auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      auto conv_row = [&](int rr, int dx, f32x4 (&acc)[4]) {
        f32x4 ci = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EDGE) {
          const int cy = cy0 + rr, cx = cxl + dx;
          const float c0 = (cy < 0 || cy >= a.Hc || cx < 0 || cx >= a.Wc) ? NEGF : 0.f;
          ci = f32x4{c0, c0, c0, c0};
        }
        const int ro = 2 * rr * PPB + 12 * dx;
        i32x4 bfrag;
        bfrag[0] = *reinterpret_cast<const int*>(lds + A0 + ro);
        bfrag[1] = *reinterpret_cast<const int*>(lds + A1 + ro);
        bfrag[2] = *reinterpret_cast<const int*>(lds + A2 + ro);
        bfrag[3] = (int)(*reinterpret_cast<const unsigned int*>(lds + A3 + ro) & M3);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[tt]), __builtin_bit_cast(f16x8, bfrag), ci, 0, 0, 0);
      };
      f32x4 prev[3][4];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) conv_row(0, dx, prev[dx]);
#pragma unroll
      for (int q = 0; q < PPR; ++q) {
        const int py = py0 + q;
        f32x4 M[4];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          // one accumulator set in flight: prev <- max(prev, row 2q+1), then M <- max(M, prev, row 2q+2) and that row
          // becomes prev (16 more VALU per pooled row than a max3 over three live sets, 16 fewer registers)
          f32x4 v[4];
          conv_row(2 * q + 1, dx, v);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) prev[dx][tt][r] = __builtin_fmaxf(prev[dx][tt][r], v[tt][r]);
          conv_row(2 * q + 2, dx, v);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              M[tt][r] = dx == 0 ? __builtin_fmaxf(prev[dx][tt][r], v[tt][r]) : max3f(M[tt][r], prev[dx][tt][r], v[tt][r]);
            prev[dx][tt] = v[tt];
          }
          if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);   // (tight register budget: phases stay in order)
        }
        unsigned int o[8];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + PBIAS + ((tt >> 1) * 32 + cb + (tt & 1) * 4) * 4);
          const f32x4 m = M[tt] + bias;
          typedef f16 h2 __attribute__((ext_vector_type(2)));
          const h2 lo = {(f16)m[0], (f16)m[1]}, hi = {(f16)m[2], (f16)m[3]};
          const unsigned int ulo = __builtin_bit_cast(unsigned int, lo), uhi = __builtin_bit_cast(unsigned int, hi);
          o[2 * tt] = SQ ? pkmax16_then_idle(ulo, 0u) : pkmax16(ulo, 0u);        // 0u = +0.0 (packed): the ReLU
          o[2 * tt + 1] = SQ ? pkmax16_then_idle(uhi, 0u) : pkmax16(uhi, 0u);
        }
        if (q == PPR - 1) {
          // vmcnt retires in order: claim the prefetched patch HERE, behind the stores of the first pooled rows only
#pragma unroll
          for (int it = 0; it < PNIT; ++it) asm volatile("" : "+v"(pf[it]));
        }
        const bool st_ok = store_lane && py < a.Hp && !(dbg & 1);
        if constexpr (SQ) {
          // squeeze1x1 of the next layer on the pooled row: pixel = lane column j, K = the 64 pooled channels in two chunks
          const i32x4 sqw0 = reinterpret_cast<const i32x4*>(lds + PSQW)[lane], sqw1 = reinterpret_cast<const i32x4*>(lds + PSQW)[64 + lane];
          const f32x4 sqb = *reinterpret_cast<const f32x4*>(lds + PSQW + 2048 + 16 * g);
          f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sqw0),
              __builtin_bit_cast(f16x8, i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sqw1),
              __builtin_bit_cast(f16x8, i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}), sacc, 0, 0, 0);
          sacc += sqb;
          typedef f16 h4 __attribute__((ext_vector_type(4)));
          const h4 hv = {(f16)fmaxf(sacc[0], 0.f), (f16)fmaxf(sacc[1], 0.f), (f16)fmaxf(sacc[2], 0.f), (f16)fmaxf(sacc[3], 0.f)};
          const int so = st_ok ? (int)(((((unsigned)cn * a.Hp + py) * a.Wp + px) * 16 + 4 * g) * 2) : POOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hv), ry, so, 0, 0);
          if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);   // pooled rows stay in order (register budget)
          continue;
        }
        const int so = st_ok ? (int)((((unsigned)cn * a.Hp + py) * a.Wp + px) * a.y_cstride + a.y_coffset + cb) * 2 : POOB;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, ry, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}, ry, st_ok ? so + 64 : POOB, 0, 0);
        if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (edge_cols || cy0 < 0 || cy0 + PCR > a.Hc) compute(std::true_type{});
    else compute(std::false_type{});
    __syncthreads();                                                 // every wave is done with the patch before it is overwritten
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// DMA form (round 4): the same tile arithmetic, but the next tile's input patch arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`)
// in a SECOND LDS buffer instead of 32 prefetch registers + a ds_write pass: 1-KiB blocks of 64 consecutive 16-byte pieces of the
// linear [row][1568 B] patch image, four to six per wave, requested at the top of a tile and waited for with a counted
// `s_waitcnt vmcnt(stores issued behind them)` at the top of the next -- ONE barrier per tile instead of two.  Without the prefetch
// registers the kernel fits 168 registers: THREE workgroups per CU with 3-pooled-row tiles (2 x 23 KiB of patch).
//   * rows above / below the image are out of range of the per-image buffer resource: zeros = conv1's SAME padding, as before;
//   * left-edge tiles (ix0 = -2: the two columns only conv column -1 reads, which is -inf under the pool): the patch origin moves to
//     column 0 and the gather base by -12 bytes instead -- what a lane then reads in front of a row is the previous row's tail (or the
//     16 zero bytes in front of the first buffer): finite, and only ever added to -inf;
//   * right-edge tiles: pieces beyond the image row are requested out of range (zeros); the ONE piece per row that straddles the row
//     end gets its tail zeroed by the wave that requested it, behind its own wait (no extra barrier).
template <int PPRT> struct PGeo {
  static constexpr int PCR = 2 * PPRT + 1, PTR = 4 * PPRT + 3;
  static constexpr int PPCS = PTR * PRP;                    // 16-byte pieces per tile
  static constexpr int NBLK = (PPCS + 63) / 64;             // 1-KiB DMA blocks
  static constexpr int BUF = NBLK * 1024;
  static constexpr int NBW = (NBLK + 3) / 4;                // blocks per wave
  static constexpr int OFF0 = 16;                           // 16 zero bytes in front of buffer 0
  static constexpr int OBIAS = OFF0 + 2 * BUF, OSQW = OBIAS + 256, LDS = OSQW + 2048 + 64;
};

__device__ __forceinline__ void stem_dma16(int voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
template <int N> __device__ __forceinline__ void stem_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool SQ, int PPRT, int WPC>
__global__ __launch_bounds__(256, WPC) void stem_phase_dma(StemArgs a, int ntiles, int per_xcd, int dbg) {
  using G = PGeo<PPRT>;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;

  i32x4 af[4];
  {
    const f16* wsrc = reinterpret_cast<const f16*>(a.wp);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      f16x8 v;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int i = s >> 1, h = s & 1;
        const int dy = g < 3 ? g : i;
        const int e = g < 3 ? (i < 3 ? 2 * i : 8) + h : 6 + h;
        const bool ok = g < 3 ? !(i == 3 && h == 1) : i < 3;
        const int kq = ok ? dy * 9 + e : 0;
        const int co = (t >> 1) * 32 + 8 * (j >> 2) + (t & 1) * 4 + (j & 3);
        const int pt = (co >> 2) & 3, pm = ((co >> 4) << 2) | (co & 3);
        const f16 wv = wsrc[((pt * 64) + (kq >> 3) * 16 + pm) * 8 + (kq & 7)];
        v[s] = ok ? wv : (f16)0;
      }
      af[t] = __builtin_bit_cast(i32x4, v);
    }
  }
  if (tid < 4) reinterpret_cast<int*>(lds)[tid] = 0;                  // the 16 bytes in front of buffer 0
  if (tid < 64) reinterpret_cast<float*>(lds + G::OBIAS)[tid] = a.bias[tid];
  if constexpr (SQ) {
    if (tid < 128) reinterpret_cast<i32x4*>(lds + G::OSQW)[tid] = reinterpret_cast<const i32x4*>(a.ws2)[tid];
    if (tid < 16) reinterpret_cast<float*>(lds + G::OSQW + 2048)[tid] = a.bs2[tid];
  }
  const int pl = 16 * wave + j;
  const int A0 = (g < 3 ? g * PPB : 12) + 24 * pl;
  const int D1 = g < 3 ? 4 : PPB;
  const int A1 = A0 + D1, A2 = A0 + 2 * D1, A3 = g < 3 ? A0 + 16 : A0;
  const unsigned int M3 = g < 3 ? 0x0000ffffu : 0u;
  const int cb = g * 8;
  const float NEGF = __uint_as_float(0xff800000u);
  const unsigned int img_bytes = (unsigned int)a.H * a.W * 6;
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      SQ ? a.s_out : a.y, 0, (unsigned int)((size_t)a.N * a.Hp * a.Wp * (SQ ? 16 : a.y_cstride) * 2), 0x00020000);

  struct Tile { int n, ty, tx; bool live; };
  const int dtx = G8 % a.tiles_x, dty = (G8 / a.tiles_x) % a.tiles_y, dn = (G8 / a.tiles_x) / a.tiles_y;
  int slot = slot0;
  Tile cur;
  {
    const int ti = xcd * per_xcd + slot;
    cur.live = slot < per_xcd && ti < ntiles;
    cur.tx = ti % a.tiles_x;
    const int q = ti / a.tiles_x;
    cur.ty = q % a.tiles_y;
    cur.n = q / a.tiles_y;
  }
  auto advance = [&]() {
    slot += G8;
    cur.live = slot < per_xcd && xcd * per_xcd + slot < ntiles;
    cur.tx += dtx;
    int c = 0;
    if (cur.tx >= a.tiles_x) { cur.tx -= a.tiles_x; c = 1; }
    cur.ty += dty + c;
    c = 0;
    if (cur.ty >= a.tiles_y) { cur.ty -= a.tiles_y; c = 1; }
    cur.n += dn + c;
  };
  if (!cur.live) return;                                              // (whole workgroup)

  // ---- DMA: per-lane constants of this wave's blocks: piece p = block * 64 + lane -> (patch row, 16-byte column) ----
  // (WPC >= 3: recomputed per tile from a laundered lane id -- 12 registers that must not live across the compute phase)
  auto piece = [&](int i, int ln, int& srel_i, int& c16b_i) {         // row * W*6 + c16 * 16 (or out of range), c16 * 16
    const int p = (wave + 4 * i) * 64 + ln;
    const int row = (int)__umulhi((unsigned)p, 43826197u);            // p / 98 for p < 2^15 (43826197 = ceil(2^32 / 98))
    const int c16 = p - row * PRP;
    const bool ex = wave + 4 * i < G::NBLK && p < G::PPCS;
    srel_i = ex ? row * (a.W * 6) + c16 * 16 : POOB;
    c16b_i = c16 * 16;
  };
  int srel[G::NBW], c16b[G::NBW];
#pragma unroll
  for (int i = 0; i < G::NBW; ++i) piece(i, lane, srel[i], c16b[i]);
  auto refresh = [&]() {
    if constexpr (WPC >= 3) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int i = 0; i < G::NBW; ++i) piece(i, ln, srel[i], c16b[i]);
    }
  };
  const unsigned long long xaddr = (unsigned long long)(uintptr_t)a.x;
  // geometry of a tile's patch: (first input row, first input column after the left-edge shift, shift in bytes)
  auto issue = [&](const Tile& t, int buf) {
    const int cy0 = 2 * (t.ty * PPRT) - a.ptp, cx0 = 2 * (t.tx * PPC) - a.plp;
    const int iy0 = 2 * cy0 - a.ptc, ix0r = 2 * cx0 - a.plc;
    const int ix0 = ix0r < 0 ? 0 : ix0r;
    const unsigned long long base = xaddr + (unsigned long long)t.n * img_bytes;
    const i32x4 rx = {(int)(unsigned)base, (int)(unsigned)((base >> 32) & 0xffffu), (int)img_bytes, 0x00020000};
    const int toff = iy0 * a.W * 6 + ix0 * 6;
    const int rowend = (a.W - ix0) * 6;                               // bytes of a patch row inside the image row
    const bool right = rowend < PPB;                                  // wave-uniform
    refresh();
#pragma unroll
    for (int i = 0; i < G::NBW; ++i) {
      if (wave + 4 * i < G::NBLK) {
        int off = srel[i] == POOB ? POOB : srel[i] + toff;            // (rows above the image: negative = out of range)
        if (right && c16b[i] >= rowend) off = POOB;                   // pieces wholly beyond the row end: zeros
        stem_dma16(off, rx, lds_addr + (unsigned)(G::OFF0 + buf * G::BUF + (wave + 4 * i) * 1024));
      }
    }
  };
  if (!(dbg & 2)) issue(cur, 0);
  stem_vm_wait<0>();
  int buf = 0;
  bool first = true;
  while (true) {
    const int py0 = cur.ty * PPRT, px0 = cur.tx * PPC;
    const int cy0 = 2 * py0 - a.ptp, cx0 = 2 * px0 - a.plp;
    const int ix0r = 2 * cx0 - a.plc;
    const int cn = cur.n;
    // ---- this tile's patch has landed once at most the previous tile's stores (issued behind its DMA) are outstanding ----
    if (!first) stem_vm_wait<SQ ? PPRT : 2 * PPRT>();
    first = false;
    {
      const int ix0 = ix0r < 0 ? 0 : ix0r;
      const int rowend = (a.W - ix0) * 6;
      if (rowend < PPB && (rowend & 15) != 0) {
        // the piece of every row that straddles the image row's end: its tail is the next row's first pixels -> zero padding
        unsigned char* bb = lds + G::OFF0 + buf * G::BUF;
        refresh();
#pragma unroll
        for (int i = 0; i < G::NBW; ++i) {
          const int p = (wave + 4 * i) * 64 + lane;
          if (wave + 4 * i < G::NBLK && p < G::PPCS && c16b[i] < rowend && c16b[i] + 16 > rowend) {
            for (int o = rowend & 15; o < 16; o += 4) *reinterpret_cast<int*>(bb + p * 16 + o) = 0;
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const unsigned char* pb = lds + G::OFF0 + buf * G::BUF - (ix0r < 0 ? -ix0r * 6 : 0);
    Tile nxt = cur;
    advance();
    const bool more = cur.live;
    if (more && !(dbg & 2)) issue(cur, buf ^ 1);

    const int cxl = cx0 + 2 * pl;
    const bool edge_cols = cx0 < 0 || cx0 + 2 * PPC + 1 > a.Wc;
    const int px = px0 + pl;
    const bool store_lane = px < a.Wp;
    auto compute = [&](auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      auto conv_row = [&](int rr, int dx, f32x4 (&acc)[4]) {
        f32x4 ci = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EDGE) {
          const int cy = cy0 + rr, cx = cxl + dx;
          const float c0 = (cy < 0 || cy >= a.Hc || cx < 0 || cx >= a.Wc) ? NEGF : 0.f;
          ci = f32x4{c0, c0, c0, c0};
        }
        const int ro = 2 * rr * PPB + 12 * dx;
        i32x4 bfrag;
        bfrag[0] = *reinterpret_cast<const int*>(pb + A0 + ro);
        bfrag[1] = *reinterpret_cast<const int*>(pb + A1 + ro);
        bfrag[2] = *reinterpret_cast<const int*>(pb + A2 + ro);
        bfrag[3] = (int)(*reinterpret_cast<const unsigned int*>(pb + A3 + ro) & M3);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[tt]), __builtin_bit_cast(f16x8, bfrag), ci, 0, 0, 0);
      };
      // horizontal first: h(row) = max over the three column phases of the conv row, then the vertical 3-max over h(2q), h(2q+1),
      // h(2q+2) -- the same nine values under the same exact max as the phase-major order of stem_phase, with one "previous row" set
      // live instead of three (48 -> 16 registers) and 80 instead of 96 max instructions per pooled row
      auto conv_row_h = [&](int rr, f32x4 (&h)[4]) {
        if constexpr (WPC >= 3) {        // (tight register budget: one more set in flight, two maxes per element)
          f32x4 v[4];
          conv_row(rr, 0, h);
          conv_row(rr, 1, v);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[tt][r] = __builtin_fmaxf(h[tt][r], v[tt][r]);
          __builtin_amdgcn_sched_barrier(0);
          conv_row(rr, 2, v);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[tt][r] = __builtin_fmaxf(h[tt][r], v[tt][r]);
        } else {
          f32x4 v1[4], v2[4];
          conv_row(rr, 0, h);
          conv_row(rr, 1, v1);
          conv_row(rr, 2, v2);
#pragma unroll
          for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h[tt][r] = max3f(h[tt][r], v1[tt][r], v2[tt][r]);
        }
      };
      f32x4 hprev[4];
      conv_row_h(0, hprev);
#pragma unroll
      for (int q = 0; q < PPRT; ++q) {
        const int py = py0 + q;
        f32x4 M[4], h2[4];
        conv_row_h(2 * q + 1, M);
        if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);
        conv_row_h(2 * q + 2, h2);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) M[tt][r] = max3f(hprev[tt][r], M[tt][r], h2[tt][r]);
          hprev[tt] = h2[tt];
        }
        if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);
        unsigned int o[8];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + G::OBIAS + ((tt >> 1) * 32 + cb + (tt & 1) * 4) * 4);
          const f32x4 m = M[tt] + bias;
          typedef f16 h2 __attribute__((ext_vector_type(2)));
          const h2 lo = {(f16)m[0], (f16)m[1]}, hi = {(f16)m[2], (f16)m[3]};
          const unsigned int ulo = __builtin_bit_cast(unsigned int, lo), uhi = __builtin_bit_cast(unsigned int, hi);
          o[2 * tt] = SQ ? pkmax16_then_idle(ulo, 0u) : pkmax16(ulo, 0u);
          o[2 * tt + 1] = SQ ? pkmax16_then_idle(uhi, 0u) : pkmax16(uhi, 0u);
        }
        const bool st_ok = store_lane && py < a.Hp && !(dbg & 1);
        if constexpr (SQ) {
          int lq = lane;
          if constexpr (WPC >= 3) asm volatile("" : "+v"(lq));      // (not hoisted out of the tile loop: 8 registers)
          const i32x4 sqw0 = reinterpret_cast<const i32x4*>(lds + G::OSQW)[lq], sqw1 = reinterpret_cast<const i32x4*>(lds + G::OSQW)[64 + lq];
          const f32x4 sqb = *reinterpret_cast<const f32x4*>(lds + G::OSQW + 2048 + 16 * g);
          f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sqw0),
              __builtin_bit_cast(f16x8, i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sqw1),
              __builtin_bit_cast(f16x8, i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}), sacc, 0, 0, 0);
          sacc += sqb;
          typedef f16 h4 __attribute__((ext_vector_type(4)));
          const h4 hv = {(f16)fmaxf(sacc[0], 0.f), (f16)fmaxf(sacc[1], 0.f), (f16)fmaxf(sacc[2], 0.f), (f16)fmaxf(sacc[3], 0.f)};
          const int so = st_ok ? (int)(((((unsigned)cn * a.Hp + py) * a.Wp + px) * 16 + 4 * g) * 2) : POOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hv), ry, so, 0, 0);
          if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);
          continue;
        }
        const int so = st_ok ? (int)((((unsigned)cn * a.Hp + py) * a.Wp + px) * a.y_cstride + a.y_coffset + cb) * 2 : POOB;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, ry, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}, ry, st_ok ? so + 64 : POOB, 0, 0);
        if constexpr (WPC >= 3) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (edge_cols || cy0 < 0 || cy0 + G::PCR > a.Hc) compute(std::true_type{});
    else compute(std::false_type{});
    (void)nxt;
    if (!more) break;
    buf ^= 1;
  }
}

template <bool SQ, int PPRT, int WPC>
static void launch_phase_dma(StemArgs a, int dbg, hipStream_t st) {
  using G = PGeo<PPRT>;
  a.tiles_x = (a.Wp + PPC - 1) / PPC;
  a.tiles_y = (a.Hp + PPRT - 1) / PPRT;
  const int ntiles = a.N * a.tiles_x * a.tiles_y;
  const int per_xcd = (ntiles + 7) / 8;
  int grid = cu_count() * WPC;
  if (per_xcd < grid / 8) grid = per_xcd * 8;
  static PerDevice once;
  (void)once.run([] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_phase_dma<SQ, PPRT, WPC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  hipLaunchKernelGGL((stem_phase_dma<SQ, PPRT, WPC>), dim3(grid), dim3(256), G::LDS, st, a, ntiles, per_xcd, dbg);
}

}  // namespace

// fp16, 3x3 / 64 couts, even W and even left pad (dword-aligned patch rows), an image at least two patches wide.
int stem_phase_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (dtype != SQDET_F16 || k != 3 || a.Cout != 64) return SQDET_OK;
  if (a.W % 2 != 0 || a.plc % 2 != 0 || a.y_cstride % 8 != 0 || a.y_coffset % 8 != 0) return SQDET_OK;
  if ((size_t)a.H * a.W * 6 >= (1ull << 31) || a.W * 6 < 2 * PRP * 16) return SQDET_OK;
  if ((size_t)a.N * a.Hp * a.Wp * a.y_cstride * 2 >= (1ull << 31)) return SQDET_OK;
  a.tiles_x = (a.Wp + PPC - 1) / PPC;
  a.tiles_y = (a.Hp + PPR - 1) / PPR;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt >= (1l << 30)) return SQDET_OK;
  const int ntiles = (int)nt;
  const int per_xcd = (ntiles + 7) / 8;
  const int dbg = tune(TUNE_DBG) >= 100 && tune(TUNE_DBG) < 104 ? tune(TUNE_DBG) - 100 : 0;   // 101 no stores, 102 no loads, 103 neither
  // The DMA form at (4 pooled rows per tile, 2 workgroups per CU) is the default since round 4 (same box, the launch alone: 47.3 us
  // against 53.9 for the register-prefetch form below -- "stem_algo" 4; the 32-image step 0.4923 against 0.5003 ms, three alternating
  // runs); "stem_algo" 5, 7, 8: (3,3) -- 168 registers, 6 spilled -- 48.6 us, (2,4) spills 48 registers, (3,2) 49.0 us  [A/B].
  // Three resident workgroups buy nothing: what the tile loop waits for is its own 4-byte LDS gather (432 ds_read_b32 per tile at
  // a few per wait), not latency another wave could cover.
  const int alg = tune(TUNE_STEM_ALGO);
  if (alg == 0 || (alg >= 5 && alg <= 8)) {
    if ((long)a.N * ((a.Wp + PPC - 1) / PPC) * ((a.Hp + 1) / 2) >= (1l << 30)) return SQDET_OK;
    if (a.ws2) {
      if (alg == 5) launch_phase_dma<true, 3, 3>(a, dbg, st);
      else if (alg == 6 || alg == 0) launch_phase_dma<true, 4, 2>(a, dbg, st);
      else if (alg == 7) launch_phase_dma<true, 2, 4>(a, dbg, st);
      else launch_phase_dma<true, 3, 2>(a, dbg, st);
    } else {
      if (alg == 5) launch_phase_dma<false, 3, 3>(a, dbg, st);
      else if (alg == 6 || alg == 0) launch_phase_dma<false, 4, 2>(a, dbg, st);
      else if (alg == 7) launch_phase_dma<false, 2, 4>(a, dbg, st);
      else launch_phase_dma<false, 3, 2>(a, dbg, st);
    }
    SQDET_CHECK_HIP(hipGetLastError());
    *handled = true;
    return SQDET_OK;
  }
  int grid = 512;                                                    // 2 workgroups per CU, a multiple of 8
  if (per_xcd < grid / 8) grid = per_xcd * 8;
  if (a.ws2) hipLaunchKernelGGL((stem_phase<true, 2>), dim3(grid), dim3(256), PLDS, st, a, ntiles, per_xcd, dbg);
  else hipLaunchKernelGGL((stem_phase<false, 2>), dim3(grid), dim3(256), PLDS, st, a, ntiles, per_xcd, dbg);
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
