// Fused stem, persistent form, fp16 / 3x3 / 64 couts (reference src/nets/squeezeDet.py:40-44: conv1 3x3/s2 + bias + ReLU,
// then pool1 3x3/s2 SAME).  Same arithmetic as stem2.hip's stem_strip (one MFMA per 16 conv columns x 16 couts, pooling
// in registers, only the pooled tensor is written); what changed is everything around the MFMAs, because the strip
// kernel was bound by its VALU / SALU instruction stream and by one exposed memory round trip per workgroup (73 us =
// 0.34 of HBM peak, profiles/r02_a_*; this kernel: 57.6 us):
//   * PERSISTENT workgroups (4 per CU) walk tiles of 4 pooled rows x 28 pooled columns in XCD-contiguous bands (9216
//     tiles at batch 32 = 9 per workgroup); the NEXT tile's 19 x 117 x 3 input patch is fetched into registers (raw buffer
//     loads over a per-image resource: rows above / below the image are out of range = zero padding, no branches)
//     while the current tile is computed.  vmcnt retires in order, so the prefetched registers are claimed BEFORE the
//     tile's last stores are issued -- at the loop top the same wait would cover those stores (a write round trip per
//     tile: 61 -> 57.6 us);
//   * im2col gather by DWORD: with the K order k = (group, dword) described below every pair of K slots is one aligned
//     LDS dword -- 4 ds_read_b32 per MFMA block instead of 8 ds_read_u16 + packing.  The A fragments for that order
//     are gathered once per workgroup from the standard packed weights, with the cout -> MFMA-row map chosen so that the
//     four lanes of a pixel store 64 contiguous bytes per instruction;
//   * vertical 3-max on the raw fp32 accumulators (v_max3_f32), THEN bias, THEN one fp16 conversion per pooled row (max
//     commutes with the monotonic x -> fl(x + b) and with rounding: the same values as bias -> convert -> max); the
//     horizontal 3-tap max stays packed-fp16 DPP row shifts, ReLU last;
//   * a tile's compute is ONE basic block (interior tiles; edge tiles get -inf for rows / columns outside the conv map
//     through the accumulator input, stores are out-of-range-predicated buffer stores): the scheduler overlaps the gather,
//     the MFMAs and the pooling of neighbouring rows.  (v_max3_f32 must come from the compiler, not from inline asm:
//     only for its own instructions does it insert the wait states between an MFMA and a VALU read of its result.)
// Results equal stem_strip's except for the float32 summation order inside the MFMA (different K order): a 1-ulp fp16
// flip in ~6e-5 of the elements (tests/test_gpu_ops.py).  Measured ladder at batch 32 (tune dbg 101/102/103): no
// stores 52.5 us, no loads 47.6, neither 43.3 -- the kernel is bound by dependent-issue latency at 4 waves per SIMD
// (removing 8 of 72 VALU instructions per pooled row by taking the bias through the accumulator input changed nothing).
// LDS row pitch 196 dwords: the four gather reads of a wave are at most 2-way bank-conflicted (brute-forced).
// Tried in round 3 and dropped: handing the tiles out by a per-XCD atomic counter instead of the static stride (so that
// workgroups placed late -- the previous batch's filter workgroups hold 32 CUs when this launch starts in the serving step:
// 58 -> 71 us -- simply take fewer tiles).  Device-scope atomics are served memory-side at ~60 ns apiece per XCD and address;
// a batch-32 launch needs 20 per microsecond per XCD: with the eight counters in one 128-byte line the launch took 155 us,
// with a line per XCD and the request issued a whole tile ahead 75 us against 58 (bitwise the same output).  The serving
// step moves the side work to where it costs nothing instead (nn_skeleton.py: the filter of batch k runs on the 16 CUs the
// chain launches of batch k+1 leave idle).
#include <type_traits>
#include "stem.h"

namespace sqdet {
namespace {

constexpr int QPR = 4;                    // pooled rows per tile
constexpr int QSP = 7;                    // pooled columns per wave strip
constexpr int QCR = 2 * QPR + 1;          // conv rows under the tile (13)
constexpr int QCC = 4 * 2 * QSP + 2;      // conv columns under the tile (58)
constexpr int QTR = 2 * (QCR - 1) + 3;    // staged input rows (27)
constexpr int QPB = 196 * 4;              // LDS row pitch in bytes
constexpr int QRP = 44;                   // 16-byte pieces fetched per row (704 B >= 117 * 3 * 2)
constexpr int QPCS = QTR * QRP;           // pieces per tile (1188)
constexpr int QNIT = (QPCS + 255) / 256;  // fetch rounds per thread (5)
constexpr int QBIAS = QTR * QPB;          // LDS offset of the 64 float32 biases
constexpr int QSQW = QBIAS + 256;         // squeeze form: the next layer's 2 KiB of squeeze1x1 fragments + 16 biases
constexpr int QLDS = QSQW + 2048 + 64;
constexpr int QOOB = (int)0x80000000;

__device__ __forceinline__ unsigned int pkmax16(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// the same followed by two idle cycles: in the squeeze form the result feeds an MFMA directly, and the wait states between
// a VALU write and a matrix instruction reading it are only inserted for instructions the compiler emits itself
__device__ __forceinline__ unsigned int pkmax16_then_idle(unsigned int a, unsigned int b) {
  unsigned int r;
  asm("v_pk_max_f16 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// NOT inline asm: the operands are MFMA results, and only for instructions it emits itself does the compiler insert the
// wait states gfx950 needs between a matrix instruction and a VALU read of its result (an asm v_max3_f32 read stale registers)
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

template <bool SQ>
__global__ __launch_bounds__(256, 4) void stem_pers(StemArgs a, int ntiles, int per_xcd, int dbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;

  // ---- once per workgroup: A fragments in the dword-gather K order, biases to LDS, lane constants ----
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
        const int kq = ok ? dy * 9 + e : 0;                          // slot of (dy, dx, ch) in the packed im2col order
                                                                     // (unused slots must not index past the 27 real ones)
        // MFMA row j of tile t computes cout co (so that a lane's registers are couts 8g..8g+7 and 32+8g..32+8g+7: the four
        // lanes of a pixel store 64 contiguous bytes per instruction); (pt, pm) = where the standard packing keeps co
        const int co = (t >> 1) * 32 + 8 * (j >> 2) + (t & 1) * 4 + (j & 3);
        const int pt = (co >> 2) & 3, pm = ((co >> 4) << 2) | (co & 3);
        const f16 wv = wsrc[((pt * 64) + (kq >> 3) * 16 + pm) * 8 + (kq & 7)];
        v[s] = ok ? wv : (f16)0;
      }
      af[t] = __builtin_bit_cast(i32x4, v);
    }
  }
  if (tid < 64) reinterpret_cast<float*>(lds + QBIAS)[tid] = a.bias[tid];
  // SQ: the next layer's squeeze1x1 (64 -> 16): its two K-chunk fragments in the standard packing ARE what the MFMA wants
  // here -- a lane's packed pooled registers o[0..3] / o[4..7] are channels 8g..8g+7 of K-chunk 0 / 1 (see the cout map above)
  // (kept in LDS, not in registers: the tile loop has none to spare at 128 per wave)
  if constexpr (SQ) {
    if (tid < 128) reinterpret_cast<i32x4*>(lds + QSQW)[tid] = reinterpret_cast<const i32x4*>(a.ws2)[tid];
    if (tid < 16) reinterpret_cast<float*>(lds + QSQW + 2048)[tid] = a.bs2[tid];
  }
  const int cc = 2 * QSP * wave + j;                                  // conv column of this lane within the tile
  // gather addresses (bytes, patch row 0): groups 0..2 read dwords 0,1,2,4 of patch row dy = g at the lane's 12*cc;
  // group 3 reads dword 3 of rows 0,1,2 (its 4th dword is masked to zero)
  const int A0 = (g < 3 ? g * QPB : 12) + 12 * cc;
  const int D1 = g < 3 ? 4 : QPB;
  const int A1 = A0 + D1, A2 = A0 + 2 * D1, A3 = g < 3 ? A0 + 16 : A0;
  const unsigned int M3 = g < 3 ? 0x0000ffffu : 0u;
  // staging constants: piece p = it*256 + tid -> (row, 16-byte column)
  int dst_off[QNIT];   // LDS byte offset of the piece; its source offset is re-derived per fetch (registers are the tight resource)
#pragma unroll
  for (int it = 0; it < QNIT; ++it) {
    const int p = it * 256 + tid;
    const int row = p / QRP, c16 = p - row * QRP;
    dst_off[it] = p < QPCS ? row * QPB + c16 * 16 : -1;
  }
  const int cb = g * 8;                                               // this lane's couts: cb..cb+7 and 32+cb..32+cb+7
  const float NEGF = __uint_as_float(0xff800000u);
  const unsigned int img_bytes = (unsigned int)a.H * a.W * 6;
  const int G8 = gridDim.x >> 3;
  const int xcd = blockIdx.x & 7, slot0 = blockIdx.x >> 3;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      SQ ? a.s_out : a.y, 0, (unsigned int)((size_t)a.N * a.Hp * a.Wp * (SQ ? 16 : a.y_cstride) * 2), 0x00020000);

  // Tile loop: the registers pf hold the patch of the NEXT tile (fetched right after this tile's patch became visible in
  // LDS), so the global-memory latency hides behind the tile's compute.  (A double-buffered LDS variant that stages the
  // next patch in the middle of a tile -- to keep the loop-top vmcnt from also waiting for the tile's last stores --
  // was built and is slower: 82 vs 61 us, spills and a split basic block.)
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
  // the walk advances by G8 tiles: (dtx, dty, dn) = decode(G8) once, then add-with-carry (no divisions in the loop)
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
  i32x4 pf[QNIT];
  auto fetch = [&]() {
    const int cy0 = 2 * (cur.ty * QPR) - a.ptp, cx0 = 2 * (cur.tx * 4 * QSP) - a.plp;
    const int iy0 = 2 * cy0 - a.ptc, ix0 = 2 * cx0 - a.plc;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f16*>(reinterpret_cast<const f16*>(a.x)) + (size_t)cur.n * a.H * a.W * 3, 0, img_bytes, 0x00020000);
    const int toff = iy0 * a.W * 6 + ix0 * 6;
    int src_off[QNIT];
#pragma unroll
    for (int it = 0; it < QNIT; ++it) {
      const int row = dst_off[it] / QPB;
      src_off[it] = dst_off[it] < 0 ? QOOB : row * (a.W * 6 - QPB) + dst_off[it];   // row * W*6 + c16*16
    }
    if (ix0 < 0) {
      // left-edge tiles: a piece of image row 0 starts at a NEGATIVE offset and ends inside the image; fetch by dword so
      // that nothing depends on how a 16-byte load range-checks a wrapped offset (1 of 12 tile columns)
#pragma unroll
      for (int it = 0; it < QNIT; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int off = src_off[it] + toff + 4 * u;
          pf[it][u] = __builtin_amdgcn_raw_buffer_load_b32(rx, src_off[it] == QOOB || off < 0 ? QOOB : off, 0, 0);
        }
    } else {
#pragma unroll
      for (int it = 0; it < QNIT; ++it)
        pf[it] = __builtin_amdgcn_raw_buffer_load_b128(rx, src_off[it] == QOOB ? QOOB : src_off[it] + toff, 0, 0);
    }
  };
  if (cur.live && !(dbg & 2)) fetch();
#pragma unroll
  for (int it = 0; it < QNIT; ++it) asm volatile("" : "+v"(pf[it]));   // first patch claimed here: on every path into the
                                                                        // loop top pf has no load pending (see below)
  while (cur.live) {
    const int py0 = cur.ty * QPR, px0 = cur.tx * (4 * QSP);
    const int cy0 = 2 * py0 - a.ptp, cx0 = 2 * px0 - a.plp;
    const int ix0 = 2 * cx0 - a.plc;
    const int cn = cur.n;
    // ---- the fetched patch -> LDS; columns outside the image are zero padding (only tiles on the left / right edge) ----
    if (ix0 < 0 || ix0 * 6 + QRP * 16 > a.W * 6) {
      const int lo = -ix0 * 3 / 2, hi = (a.W - ix0) * 3 / 2;         // valid dwords of a patch row (ix0, W even)
#pragma unroll
      for (int it = 0; it < QNIT; ++it) {
        const int pdw = (dst_off[it] % QPB) >> 2;                    // first dword of the piece within its patch row
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (pdw + u < lo || pdw + u >= hi) pf[it][u] = 0;
      }
    }
#pragma unroll
    for (int it = 0; it < QNIT; ++it)
      if (dst_off[it] >= 0) *reinterpret_cast<i32x4*>(lds + dst_off[it]) = pf[it];
    __syncthreads();
    // ---- next tile's patch goes in flight now ----
    advance();
    if (cur.live && !(dbg & 2)) fetch();

    // ---- compute: 9 conv rows -> 4 pooled rows ----
    const int cx = cx0 + cc;
    const bool col_ok = cx >= 0 && cx < a.Wc;
    const bool edge_cols = cx0 < 0 || cx0 + QCC > a.Wc;              // workgroup-uniform
    const bool col_bad = edge_cols && !col_ok;
    const int pxl = QSP * wave + (j >> 1);
    const int px = px0 + pxl;
    const bool store_lane = (j & 1) == 0 && j < 2 * QSP && px < a.Wp;
    // No branches inside a tile (one basic block: the scheduler overlaps the gather / MFMA / pooling of neighbouring rows).
    // EDGE tiles (top / bottom / left / right of the map, 1 in 4): a conv row or column outside the map gets -inf through
    // the accumulator input -- it never wins a max; interior tiles use the inline constant 0.
    auto compute = [&](auto edge_tag) {
      constexpr bool EDGE = decltype(edge_tag)::value;
      auto conv_row = [&](int rr, f32x4 (&acc)[4]) {
        f32x4 ci = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EDGE) {
          const int cy = cy0 + rr;
          const float c0 = (cy < 0 || cy >= a.Hc || col_bad) ? NEGF : 0.f;
          ci = f32x4{c0, c0, c0, c0};
        }
        const int ro = 2 * rr * QPB;
        i32x4 bfrag;
        bfrag[0] = *reinterpret_cast<const int*>(lds + A0 + ro);
        bfrag[1] = *reinterpret_cast<const int*>(lds + A1 + ro);
        bfrag[2] = *reinterpret_cast<const int*>(lds + A2 + ro);
        bfrag[3] = (int)(*reinterpret_cast<const unsigned int*>(lds + A3 + ro) & M3);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[tt]), __builtin_bit_cast(f16x8, bfrag), ci, 0, 0, 0);
      };
      f32x4 prev[4], va[4], vb[4];
      conv_row(0, prev);
#pragma unroll
      for (int q = 0; q < QPR; ++q) {
        const int py = py0 + q;
        conv_row(2 * q + 1, va);
        conv_row(2 * q + 2, vb);
        unsigned int o[8];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
          const f32x4 bias = *reinterpret_cast<const f32x4*>(lds + QBIAS + ((tt >> 1) * 32 + cb + (tt & 1) * 4) * 4);
          f32x4 m;
#pragma unroll
          for (int r = 0; r < 4; ++r) m[r] = max3f(prev[tt][r], va[tt][r], vb[tt][r]);
          m += bias;
          typedef f16 h2 __attribute__((ext_vector_type(2)));
          const h2 lo = {(f16)m[0], (f16)m[1]}, hi = {(f16)m[2], (f16)m[3]};
          o[2 * tt] = __builtin_bit_cast(unsigned int, lo);
          o[2 * tt + 1] = __builtin_bit_cast(unsigned int, hi);
          prev[tt] = vb[tt];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          // lanes j+1, j+2 of the 16-lane row; lanes 14, 15 read zeros past the row end (bound_ctrl) and store nothing
          const unsigned int s1 = (unsigned int)__builtin_amdgcn_mov_dpp((int)o[i], 0x101, 0xf, 0xf, true);
          const unsigned int s2 = (unsigned int)__builtin_amdgcn_mov_dpp((int)o[i], 0x102, 0xf, 0xf, true);
          const unsigned int hm = pkmax16(o[i], pkmax16(s1, s2));
          o[i] = SQ ? pkmax16_then_idle(hm, 0u) : pkmax16(hm, 0u);   // 0u = +0.0 (packed): the ReLU
        }
        if (q == QPR - 1) {
          // vmcnt retires in order: claim the prefetched patch HERE, behind the stores of the first pooled rows only --
          // at the loop top the same wait would also cover this tile's last stores (a full write round trip per tile)
#pragma unroll
          for (int it = 0; it < QNIT; ++it) asm volatile("" : "+v"(pf[it]));
        }
        // raw buffer stores: a lane without a pixel (odd j, beyond the strip / image, a pooled row below the map) stores
        // out of range = nowhere
        const bool st_ok = store_lane && py < a.Hp && !(dbg & 1);
        if constexpr (SQ) {
          // squeeze1x1 of the next layer on the pooled row: pixel = lane column j (the odd columns carry no pooled pixel and
          // produce values nobody stores), K = the 64 pooled channels in two chunks, ascending; D row 4g + r = squeeze cout
          const i32x4 sqw0 = reinterpret_cast<const i32x4*>(lds + QSQW)[lane], sqw1 = reinterpret_cast<const i32x4*>(lds + QSQW)[64 + lane];
          const f32x4 sqb = *reinterpret_cast<const f32x4*>(lds + QSQW + 2048 + 16 * g);
          f32x4 sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sqw0),
              __builtin_bit_cast(f16x8, i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          sacc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, sqw1),
              __builtin_bit_cast(f16x8, i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}), sacc, 0, 0, 0);
          sacc += sqb;
          typedef f16 h4 __attribute__((ext_vector_type(4)));
          const h4 hv = {(f16)fmaxf(sacc[0], 0.f), (f16)fmaxf(sacc[1], 0.f), (f16)fmaxf(sacc[2], 0.f), (f16)fmaxf(sacc[3], 0.f)};
          const int so = st_ok ? (int)(((((unsigned)cn * a.Hp + py) * a.Wp + px) * 16 + 4 * g) * 2) : QOOB;
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(i32x2, hv), ry, so, 0, 0);
          __builtin_amdgcn_sched_barrier(0);   // pooled rows stay in order: overlapping them costs more registers than the wave has
          continue;
        }
        const int so = st_ok ? (int)((((unsigned)cn * a.Hp + py) * a.Wp + px) * a.y_cstride + a.y_coffset + cb) * 2 : QOOB;
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[0], (int)o[1], (int)o[2], (int)o[3]}, ry, so, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(i32x4{(int)o[4], (int)o[5], (int)o[6], (int)o[7]}, ry, st_ok ? so + 64 : QOOB, 0, 0);
      }
    };
    if (edge_cols || cy0 < 0 || cy0 + QCR > a.Hc) compute(std::true_type{});
    else compute(std::false_type{});
    __syncthreads();                                                 // every wave is done with the patch before it is overwritten
  }
}

}  // namespace

// fp16, 3x3 / 64 couts, even W and even left pad (dword-aligned patch rows), one image below 2 GiB.
int stem_pers_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (dtype != SQDET_F16 || k != 3 || a.Cout != 64) return SQDET_OK;
  if (a.W % 2 != 0 || a.plc % 2 != 0 || a.y_cstride % 8 != 0 || a.y_coffset % 8 != 0) return SQDET_OK;
  if ((size_t)a.H * a.W * 6 >= (1ull << 31) || a.W * 6 < 2 * QRP * 16) return SQDET_OK;
  if ((size_t)a.N * a.Hp * a.Wp * a.y_cstride * 2 >= (1ull << 31)) return SQDET_OK;
  a.tiles_x = (a.Wp + 4 * QSP - 1) / (4 * QSP);
  a.tiles_y = (a.Hp + QPR - 1) / QPR;
  const long nt = (long)a.N * a.tiles_x * a.tiles_y;
  if (nt >= (1l << 30)) return SQDET_OK;
  const int ntiles = (int)nt;
  const int per_xcd = (ntiles + 7) / 8;
  int grid = 1024;                                                   // 4 workgroups per CU, a multiple of 8
  if (per_xcd < grid / 8) grid = per_xcd * 8;
  const int dbg = tune(TUNE_DBG) >= 100 && tune(TUNE_DBG) < 200 ? tune(TUNE_DBG) - 100 : 0;
  if (a.ws2) hipLaunchKernelGGL(stem_pers<true>, dim3(grid), dim3(256), QLDS, st, a, ntiles, per_xcd, dbg);
  else hipLaunchKernelGGL(stem_pers<false>, dim3(grid), dim3(256), QLDS, st, a, ntiles, per_xcd, dbg);
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet
