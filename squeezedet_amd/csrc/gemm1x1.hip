// 1x1 convolution (any stride) with deep K as a workgroup-level GEMM tile (the squeeze / expand1x1 layers of SqueezeDet+,
// every 1x1 of ResNet50's bottleneck blocks incl. the residual-accumulate ones; reference src/nn_skeleton.py:471-563 via
// nets/squeezeDetPlus.py:81-106 and nets/resnet50_convDet.py:134-169).
//
// The generic conv_direct lets every wave fetch BOTH operands itself (8 KB per 16 MFMAs: the four SIMDs ask the CU's
// 64 B/clk L1 for about twice what it delivers); conv1x1_stream keeps the weights in registers and only fits K <= 4
// chunks.  Here a 256-thread workgroup owns 16*MB consecutive NHWC pixels (a 1x1 conv has no halo: "pixel" = row of the
// [P, Cin] activation matrix) and up to 4*NTW cout tiles:
//   * the activations of the tile are staged into LDS in stages of SC 64-byte K-chunks, laid out [chunk][pixel][4 x 16 B]
//     with the 16-byte slot of lane group g stored at g ^ ((pixel>>1)&3) (conv3x3.hip's bank-conflict-free layout), so
//     each activation byte crosses L2 -> CU once per workgroup and is read from LDS by the four waves;
//   * a wave owns MBW pixel blocks x NTW cout tiles (MBW*NTW accumulators) -- one 1-KiB weight fragment from L1 feeds MBW
//     MFMAs, one LDS B fragment NTW of them: (NTW + MBW) KiB of operands per MBW*NTW MFMAs.  The four waves are laid out
//     WR along the pixel blocks x 4/WR along the cout slices: WR = 1 when the conv has >= 4 slices (a slice = NTW cout
//     tiles), 2 or 4 when it has fewer (ResNet50's 256 -> 64 reductions have ONE: without the pixel split three of the
//     four waves only helped staging);
//   * weight fragments come straight from global one K-chunk ahead (register double buffer), across the staging barriers;
//   * epilogue: bias (+ the residual already in y: ResNet's branch2c) + ReLU, 8*NTW contiguous bytes per lane.
// Accumulation order = ascending K chunks, as in every other conv kernel here (bitwise-equal to conv_direct).
#include "conv_common.h"

namespace sqdet {
namespace {

struct G1Args {
  ConvArgs c;
  int nt_pack;       // tiles per packed cout group
  int slices;        // wave slices = ngroups * (nt_pack / NTW)
  int grid_y;        // workgroups per pixel tile = ceil(slices / (4 / WR))
  int ptiles;        // pixel tiles
  int pieces;        // 16-byte pieces per pixel = Cin * sizeof(T) / 16
  int stage_chunks;  // K-chunks resident in LDS at a time
};

template <typename T, int MBW, int NTW, int WR>
__global__ __launch_bounds__(256) void conv1x1_tile(G1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int MB = MBW * WR;           // pixel blocks per tile
  constexpr int WS = 4 / WR;             // cout slices per workgroup
  constexpr int TP = 16 * MB;            // pixels per tile
  constexpr int CH = TP * 64;            // bytes of one K-chunk of the tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  // XCD-aware order: workgroup L runs on XCD L % 8; every XCD gets a contiguous band of pixel tiles, and the grid_y
  // workgroups that share a pixel tile (different couts) are neighbours on the same XCD (the tile is fetched into ONE L2)
  const int per_xcd = (a.ptiles + 7) / 8;
  const int idx = (int)(blockIdx.x >> 3);
  const int tl = idx / a.grid_y, ysl = idx - tl * a.grid_y;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + tl;
  if (tl >= per_xcd || tile >= a.ptiles) return;
  const int p0 = tile * TP;

  const int wr = wave % WR;              // this wave's pixel blocks: wr*MBW ..
  const int slice = ysl * WS + wave / WR;
  const bool active = slice < a.slices;
  const int spg = a.nt_pack / NTW;       // slices per packed group
  const int group = active ? slice / spg : 0;
  const int n0 = active ? (slice - group * spg) * NTW : 0;

  f32x4 acc[MBW][NTW];
#pragma unroll
  for (int m = 0; m < MBW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const i32x4* wbase = reinterpret_cast<const i32x4*>(a.c.wp) + ((size_t)group * a.c.steps * a.nt_pack + n0) * 64 + lane;
  auto wptr = [&](int c) { return wbase + (size_t)c * a.nt_pack * 64; };
  i32x4 af[NTW], afn[NTW];
  if (active) {
#pragma unroll
    for (int t = 0; t < NTW; ++t) af[t] = wptr(0)[t * 64];
  }

  // the input may be a channel slice [x_coffset, x_coffset + Cin) of rows x_cstride channels wide (backward-data of a fire
  // module reads the expand1x1 half of dY)
  const unsigned char* x = reinterpret_cast<const unsigned char*>(a.c.x) + (size_t)a.c.x_coffset * sizeof(T);
  const int row_bytes = a.c.x_cstride * (int)sizeof(T);
  const int nchunk = a.c.nchunk;
  for (int c0 = 0; c0 < nchunk; c0 += a.stage_chunks) {
    const int nload = nchunk - c0 < a.stage_chunks ? nchunk - c0 : a.stage_chunks;
    if (c0 > 0) __syncthreads();           // everyone is done reading the previous stage
    {
      // stage nload chunks of the tile: batches of 4 loads in flight per thread before the LDS stores
      const int ppc = nload * 4, total = TP * ppc;
      for (int base = threadIdx.x; base < total; base += 256 * 4) {
        i32x4 v[4];
        int off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int id = base + u * 256;
          const int P = id / ppc, q = id - P * ppc;
          const int gq = c0 * 4 + q;
          v[u] = i32x4{0, 0, 0, 0};
          if (id < total && gq < a.pieces && p0 + P < a.c.P) {
            size_t sp = (size_t)(p0 + P);                 // source pixel of output pixel p0 + P
            if (a.c.stride != 1) {                        // strided 1x1 (ResNet50's res3a / res4a reductions): every stride-th pixel
              const int hw = a.c.Ho * a.c.Wo;
              const int n = (p0 + P) / hw, r = (p0 + P) - n * hw;
              const int oy = r / a.c.Wo, ox = r - oy * a.c.Wo;
              sp = ((size_t)n * a.c.H + (size_t)oy * a.c.stride) * a.c.W + (size_t)ox * a.c.stride;
            }
            v[u] = *reinterpret_cast<const i32x4*>(x + sp * row_bytes + gq * 16);
          }
          off[u] = id < total ? (q >> 2) * CH + P * 64 + (((q & 3) ^ ((P >> 1) & 3)) << 4) : -1;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (off[u] >= 0) *reinterpret_cast<i32x4*>(lds + off[u]) = v[u];
      }
    }
    __syncthreads();
    if (!active) continue;
#pragma unroll 1
    for (int cl = 0; cl < nload; ++cl) {
      const int c = c0 + cl;
      if (c + 1 < nchunk) {
        const i32x4* wp = wptr(c + 1);
#pragma unroll
        for (int t = 0; t < NTW; ++t) afn[t] = wp[t * 64];
      }
      // pixel (wr*MBW + m)*16 + j: (P>>1)&3 == (j>>1)&3
      const unsigned char* lc = lds + cl * CH + (wr * MBW * 16 + j) * 64 + ((g ^ ((j >> 1) & 3)) << 4);
      i32x4 bf[MBW];
#pragma unroll
      for (int m = 0; m < MBW; ++m) bf[m] = *reinterpret_cast<const i32x4*>(lc + m * 16 * 64);
#pragma unroll
      for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], af[t], bf[m]);
      if (c + 1 < nchunk) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) af[t] = afn[t];
      }
    }
  }
  if (!active) return;

  // epilogue: lane = pixel (block m, column j); couts group*16*NTp + g*4*NTp + n0*4 .. : 4*NTW consecutive
  T* y = reinterpret_cast<T*>(a.c.y);
  const int cb = group * 16 * a.nt_pack + g * 4 * a.nt_pack + n0 * 4;
  f32x4 bias[NTW];
  int nt_valid = 0;   // Cout is a multiple of 4: whole 4-cout pieces beyond Cout are skipped
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const bool ok = cb + t * 4 < a.c.Cout;
    bias[t] = ok && a.c.bias ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    nt_valid += ok ? 1 : 0;
  }
#pragma unroll
  for (int m = 0; m < MBW; ++m) {
    const int p = p0 + (wr * MBW + m) * 16 + j;
    if (p >= a.c.P) continue;
    T* dst = y + (size_t)p * a.c.y_cstride + a.c.y_coffset + cb;
    f32x4 v[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v[t] = acc[m][t] + bias[t];
      if (a.c.accum && t < nt_valid) {
        v[t][0] += (float)dst[t * 4 + 0]; v[t][1] += (float)dst[t * 4 + 1];
        v[t][2] += (float)dst[t * 4 + 2]; v[t][3] += (float)dst[t * 4 + 3];
      }
      if (a.c.relu) {
        v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
        v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
      }
      if (a.c.relu_of && t < nt_valid) {   // ReLU backward of the layer below (see ConvArgs)
        const T* r = reinterpret_cast<const T*>(a.c.relu_of) + (dst - y) + t * 4;
        v[t][0] = (float)r[0] > 0.f ? v[t][0] : 0.f; v[t][1] = (float)r[1] > 0.f ? v[t][1] : 0.f;
        v[t][2] = (float)r[2] > 0.f ? v[t][2] : 0.f; v[t][3] = (float)r[3] > 0.f ? v[t][3] : 0.f;
      }
    }
    store_couts<T, NTW>(dst, v, nt_valid);
  }
}

template <typename T, int MBW, int NTW, int WR>
void launch_g1(const G1Args& a, size_t lds, hipStream_t st) {
  const int per_xcd = (a.ptiles + 7) / 8;
  const dim3 grid((unsigned)(per_xcd * a.grid_y * 8));
  hipLaunchKernelGGL((conv1x1_tile<T, MBW, NTW, WR>), grid, dim3(256), lds, st, a);
}

template <typename T, int MBW, int WR>
bool dispatch_ntw(const G1Args& a, int ntw, size_t lds, hipStream_t st) {
  switch (ntw) {
    case 1: if constexpr (WR == 4) { launch_g1<T, MBW, 1, WR>(a, lds, st); return true; } return false;
    case 2: if constexpr (WR == 4) { launch_g1<T, MBW, 2, WR>(a, lds, st); return true; } return false;
    case 3: launch_g1<T, MBW, 3, WR>(a, lds, st); return true;
    case 4: launch_g1<T, MBW, 4, WR>(a, lds, st); return true;
    case 5: if constexpr (MBW <= 4) { launch_g1<T, MBW, 5, WR>(a, lds, st); return true; } return false;
    default: return false;
  }
}

template <typename T>
bool dispatch_g1(const G1Args& a, int mbw, int wr, int ntw, size_t lds, hipStream_t st) {
  if (wr == 1) return mbw == 8 ? dispatch_ntw<T, 8, 1>(a, ntw, lds, st) : dispatch_ntw<T, 4, 1>(a, ntw, lds, st);
  if (wr == 2) return mbw == 4 ? dispatch_ntw<T, 4, 2>(a, ntw, lds, st) : dispatch_ntw<T, 2, 2>(a, ntw, lds, st);
  return mbw == 4 ? dispatch_ntw<T, 4, 4>(a, ntw, lds, st) : dispatch_ntw<T, 2, 4>(a, ntw, lds, st);
}


// ---------------------------------------------------------------------------------------------------------------------------
// conv1x1_pipe (round 6): the same decomposition -- a wave owns MBW pixel blocks x NTW cout tiles, the four waves are laid out
// WR x 4/WR -- as a software PIPELINE.  conv1x1_tile alternates "fetch a 4-chunk stage, store it to LDS, barrier, compute": a
// workgroup's loads and MFMAs never overlap, and where a launch has only one workgroup per CU (the 13-15 k-pixel maps of ResNet50
// res4 / SqueezeDet+ fire9-11 at the benchmark batches) every stage is an exposed memory round trip -- 26-37 us for launches whose
// bytes move in 8 and whose MFMAs issue in 4 (profiles/r05_conv1x1_shapes_ab.txt).  Here
//   * the activation tile arrives by LDS-DMA (`buffer_load_dwordx4 ... lds`: no registers, no LDS stores) in a ring of NS one-chunk
//     slots, chunk c + NS - 1 requested while chunk c computes; out-of-range offsets (pixels past the end, channel padding of the last
//     chunk, chunks past the last one) land as zeros, so there is no branch in the loop;
//   * the weight fragments of chunk c + NS - 1 are requested at the same point into the register set chunk c - 1 has left: NS sets
//     named statically over an NS-times unrolled trip, no copies (chunks past the last one re-read the last chunk's weights and
//     multiply zeros: K is walked in whole trips);
//   * ONE counted wait + one bare barrier per chunk: `s_waitcnt vmcnt((NS - 2) * (Q + NTW))` leaves the two younger chunks' requests
//     in flight (vmcnt is one in-order counter over the DMA pieces and the fragment loads), the barrier publishes the other waves'
//     pieces of the slot and retires everybody's reads of the slot that is refilled behind it;
//   * <= 256 registers: two workgroups per CU where the launch has them, so one's ramp and epilogue run under the other's MFMAs.
// Accumulation order = ascending K chunks (+ 0 for the padding chunks): bitwise conv1x1_tile's and conv_direct's results.
struct P1Args {
  ConvArgs c;
  int nt_pack, slices, grid_y, ptiles, pieces;
  int nchunk_pad;      // K chunks rounded up to whole NS-chunk trips
  unsigned x_bytes;    // extent of x (buffer resource: offsets beyond it read zeros)
};

// -DSQDET_G1_TIMELINE (experiments only, tools/g1_timeline.py): six s_memrealtime (100 MHz) stamps per workgroup -- entry, first NS - 1
// chunks requested, chunk 0 landed, K loop done, queue drained, last store issued -- written once at the very end
#ifdef SQDET_G1_TIMELINE
__device__ unsigned long long g_g1_tl[8192 * 8];
#define GTL(k) do { gtl[k] = wall_clock64(); } while (0)
#else
#define GTL(k) do {} while (0)
#endif

__device__ __forceinline__ void g1_dma16(unsigned voff, const i32x4& rsrc, unsigned lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}

template <typename T, int MBW, int NTW, int WR, int NS>
__global__ __launch_bounds__(256, 2) void conv1x1_pipe(P1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int MB = MBW * WR;           // pixel blocks per tile
  constexpr int WS = 4 / WR;             // cout slices per workgroup
  constexpr int TP = 16 * MB;            // pixels per tile
  constexpr int CH = TP * 64;            // bytes of one K-chunk of the tile = one ring slot
  constexpr int Q = MB / 4;              // DMA pieces (16 pixels x 64 B) per wave per chunk
  static_assert(MB % 4 == 0, "every wave fetches the same number of pieces (one wait count for all)");
  constexpr unsigned OOB = 0x80000000u;
  const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int j = lane & 15, g = lane >> 4;
#ifdef SQDET_G1_TIMELINE
  unsigned long long gtl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  GTL(0);
  // XCD-aware order (as conv1x1_tile): workgroup L runs on XCD L % 8; an XCD owns a contiguous band of pixel tiles, the grid_y
  // workgroups of one pixel tile are neighbours on the same XCD
  const int per_xcd = (a.ptiles + 7) / 8;
  const int idx = (int)(blockIdx.x >> 3);
  const int tl = idx / a.grid_y, ysl = idx - tl * a.grid_y;
  const int tile = (int)(blockIdx.x & 7) * per_xcd + tl;
  if (tl >= per_xcd || tile >= a.ptiles) return;
  const int p0 = tile * TP;

  const int wr = wave % WR;
  const int slice_raw = ysl * WS + wave / WR;
  const bool active = slice_raw < a.slices;
  const int slice = active ? slice_raw : a.slices - 1;   // (a wave without a slice still fetches its pieces and keeps the barriers)
  const int spg = a.nt_pack / NTW;
  const int group = slice / spg;
  const int n0 = (slice - group * spg) * NTW;
  const int nchunk = a.c.nchunk;

  // ---- this lane's DMA sources: piece i = pixel block wave + 4 i of the tile; lane = (pixel l >> 2, LDS slot l & 3)
  const unsigned long long xaddr = (unsigned long long)(uintptr_t)a.c.x;
  const i32x4 rx = {(int)(unsigned)xaddr, (int)(unsigned)((xaddr >> 32) & 0xffffu), (int)a.x_bytes, 0x00020000};
  const int esz = (int)sizeof(T);
  const int row_bytes = a.c.x_cstride * esz;
  const int piece = (lane & 3) ^ ((lane >> 3) & 3);              // the slot's XOR swizzle: (P >> 1) & 3 == (l >> 3) & 3
  const int nvalid = (a.pieces - piece + 3) >> 2;                // chunks c < nvalid hold this lane's piece
  unsigned xoff[Q];
#pragma unroll
  for (int i = 0; i < Q; ++i) {
    const int p = p0 + (wave + 4 * i) * 16 + (lane >> 2);
    unsigned sp = (unsigned)p;
    if (a.c.stride != 1) {                                       // strided 1x1: every stride-th pixel of the input map
      const int hw = a.c.Ho * a.c.Wo;
      const int n = p / hw, r = p - n * hw;
      const int oy = r / a.c.Wo, ox = r - oy * a.c.Wo;
      sp = (unsigned)((n * a.c.H + oy * a.c.stride) * a.c.W + ox * a.c.stride);
    }
    xoff[i] = p < a.c.P ? sp * (unsigned)row_bytes + (unsigned)(a.c.x_coffset * esz + piece * 16) : OOB;
  }
  const i32x4* wbase = reinterpret_cast<const i32x4*>(a.c.wp) + ((size_t)group * a.c.steps * a.nt_pack + n0) * 64 + lane;

  i32x4 wf[NS][NTW];
  // requests chunk c: the tile's pieces into ring slot `slot`, the weight fragments into register set `set` (both static)
  auto request = [&](int c, int set, int slot) {
#pragma unroll
    for (int i = 0; i < Q; ++i)
      g1_dma16(c < nvalid ? xoff[i] + (unsigned)(c * 64) : OOB, rx, lds_addr + (unsigned)(slot * CH + (wave + 4 * i) * 1024));
    const int cw = c < nchunk ? c : nchunk - 1;
    const i32x4* wp = wbase + (size_t)cw * a.nt_pack * 64;
#pragma unroll
    for (int t = 0; t < NTW; ++t) wf[set][t] = wp[t * 64];
  };
#pragma unroll
  for (int u = 0; u < NS - 1; ++u) request(u, u, u);
  GTL(1);

  f32x4 acc[MBW][NTW];
#pragma unroll
  for (int m = 0; m < MBW; ++m)
#pragma unroll
    for (int t = 0; t < NTW; ++t) acc[m][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned char* lrd = lds + (wr * MBW * 16 + j) * 64 + ((g ^ ((j >> 1) & 3)) << 4);
#pragma unroll 1
  for (int c0 = 0; c0 < a.nchunk_pad; c0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      // chunk c0 + u has landed (mine: the wait; everybody's: the barrier); everybody is done with slot (u - 1) % NS
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NS - 2) * (Q + NTW)) : "memory");
#ifdef SQDET_G1_TIMELINE
      if (c0 == 0 && u == 0) GTL(2);
#endif
      request(c0 + u + NS - 1, (u + NS - 1) % NS, (u + NS - 1) % NS);
      i32x4 bf[MBW];
#pragma unroll
      for (int m = 0; m < MBW; ++m) bf[m] = *reinterpret_cast<const i32x4*>(lrd + u * CH + m * 16 * 64);
#pragma unroll
      for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int t = 0; t < NTW; ++t) mma16<T>(acc[m][t], wf[u][t], bf[m]);
    }
  }
  GTL(3);
  // (the trailing requests -- zeros into free slots, fragments nobody uses -- retire before the wave ends)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTL(4);
  if (!active) return;

  T* y = reinterpret_cast<T*>(a.c.y);
  const int cb = group * 16 * a.nt_pack + g * 4 * a.nt_pack + n0 * 4;
  f32x4 bias[NTW];
  int nt_valid = 0;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const bool ok = cb + t * 4 < a.c.Cout;
    bias[t] = ok && a.c.bias ? *reinterpret_cast<const f32x4*>(a.c.bias + cb + t * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    nt_valid += ok ? 1 : 0;
  }
#pragma unroll
  for (int m = 0; m < MBW; ++m) {
    const int p = p0 + (wr * MBW + m) * 16 + j;
    if (p >= a.c.P) continue;
    T* dst = y + (size_t)p * a.c.y_cstride + a.c.y_coffset + cb;
    f32x4 v[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      v[t] = acc[m][t] + bias[t];
      if (a.c.accum && t < nt_valid) {
        v[t][0] += (float)dst[t * 4 + 0]; v[t][1] += (float)dst[t * 4 + 1];
        v[t][2] += (float)dst[t * 4 + 2]; v[t][3] += (float)dst[t * 4 + 3];
      }
      if (a.c.relu) {
        v[t][0] = fmaxf(v[t][0], 0.f); v[t][1] = fmaxf(v[t][1], 0.f);
        v[t][2] = fmaxf(v[t][2], 0.f); v[t][3] = fmaxf(v[t][3], 0.f);
      }
      if (a.c.relu_of && t < nt_valid) {
        const T* r = reinterpret_cast<const T*>(a.c.relu_of) + (dst - y) + t * 4;
        v[t][0] = (float)r[0] > 0.f ? v[t][0] : 0.f; v[t][1] = (float)r[1] > 0.f ? v[t][1] : 0.f;
        v[t][2] = (float)r[2] > 0.f ? v[t][2] : 0.f; v[t][3] = (float)r[3] > 0.f ? v[t][3] : 0.f;
      }
    }
    store_couts<T, NTW>(dst, v, nt_valid);
  }
#ifdef SQDET_G1_TIMELINE
  GTL(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GTL(6);
  if (threadIdx.x == 0 && blockIdx.x < 8192) {
#pragma unroll
    for (int k = 0; k < 8; ++k) g_g1_tl[(size_t)blockIdx.x * 8 + k] = gtl[k];
  }
#endif
}

template <typename T, int MBW, int NTW, int WR, int NS>
void launch_p1(const P1Args& a, hipStream_t st) {
  const int per_xcd = (a.ptiles + 7) / 8;
  const dim3 grid((unsigned)(per_xcd * a.grid_y * 8));
  const size_t lds = (size_t)NS * 16 * MBW * WR * 64;
  hipLaunchKernelGGL((conv1x1_pipe<T, MBW, NTW, WR, NS>), grid, dim3(256), lds, st, a);
}

template <typename T, int MBW, int WR, int NS>
bool dispatch_p1_ntw(const P1Args& a, int ntw, hipStream_t st) {
  switch (ntw) {
    case 3: launch_p1<T, MBW, 3, WR, NS>(a, st); return true;
    case 4: launch_p1<T, MBW, 4, WR, NS>(a, st); return true;
    case 5: if constexpr (MBW <= 4) { launch_p1<T, MBW, 5, WR, NS>(a, st); return true; } return false;
    default: return false;
  }
}

template <typename T, int NS>
bool dispatch_p1(const P1Args& a, int mbw, int wr, int ntw, hipStream_t st) {
  if (wr == 1) return mbw == 8 ? dispatch_p1_ntw<T, 8, 1, NS>(a, ntw, st) : dispatch_p1_ntw<T, 4, 1, NS>(a, ntw, st);
  if (wr == 2) return mbw == 4 ? dispatch_p1_ntw<T, 4, 2, NS>(a, ntw, st) : dispatch_p1_ntw<T, 2, 2, NS>(a, ntw, st);
  return mbw == 4 ? dispatch_p1_ntw<T, 4, 4, NS>(a, ntw, st) : dispatch_p1_ntw<T, 2, 4, NS>(a, ntw, st);
}

}  // namespace

// Eligibility + configuration.  *handled = false means "use the generic kernel".  Serves plain and residual-accumulate
// (accum) 1x1 / stride-1 convolutions whose input is a whole tensor (no channel slice) with Cin a multiple of the lane chunk.
int conv1x1_tile_launch(const ConvArgs& c, const ConvGeom& g, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (c.k != 1 || c.stride < 1 || c.pt != 0 || c.pl != 0 || g.gather) return SQDET_OK;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  if ((c.Cin * esz) % 16 != 0 || (c.x_cstride * esz) % 16 != 0 || (c.x_coffset * esz) % 16 != 0) return SQDET_OK;
  if (g.nt < 3 && g.ngroups != 1) return SQDET_OK;          // 1- and 2-tile groups: only the one-slice layout (Cout <= 32)
  G1Args a;
  a.c = c;
  a.nt_pack = g.nt;
  a.pieces = c.Cin * esz / 16;
  const int ntw = g.nt <= 5 ? g.nt : g.nt / 2;            // 6-tile groups: two waves per group (accumulator budget)
  a.slices = g.ngroups * (g.nt / ntw);
  const int wr = a.slices >= 4 ? 1 : a.slices >= 2 ? 2 : 4;   // waves along the pixel blocks
  a.grid_y = (a.slices + 4 / wr - 1) / (4 / wr);
  // pixel blocks per wave: large tiles (128 pixels per wave column, 256 for the one-slice layout) unless that leaves the
  // chip under-filled (the 24x78 maps at batch 8 are 15 k pixels)
  int mbw = wr == 1 ? 8 : 4;
  auto wgs = [&](int m) { return (long)((c.P + 16 * m * wr - 1) / (16 * m * wr)) * a.grid_y; };
  if (wgs(mbw) < 768 || (wr == 1 && ntw == 5)) mbw /= 2;
  a.ptiles = (c.P + 16 * mbw * wr - 1) / (16 * mbw * wr);
  // the pipelined form (conv1x1_pipe) for 3..5 cout tiles per wave; "dbg" 57: conv1x1_tile for everything (A/B)
  {
    const size_t xb = (size_t)c.N * c.H * c.W * c.x_cstride * esz;
    if (ntw >= 3 && xb < (1ull << 31) && tune(TUNE_DBG) != 57) {
      P1Args p;
      p.c = c; p.nt_pack = a.nt_pack; p.slices = a.slices; p.grid_y = a.grid_y; p.ptiles = a.ptiles; p.pieces = a.pieces;
      p.x_bytes = (unsigned)xb;
      // ring depth = register sets = chunks per unrolled trip: the one that walks K in whole trips (4 when both do)
      const int pad4 = (g.nchunk + 3) / 4 * 4, pad3 = (g.nchunk + 2) / 3 * 3;
      const int ns = pad4 <= pad3 ? 4 : 3;
      p.nchunk_pad = ns == 4 ? pad4 : pad3;
      const bool okp = dtype == SQDET_F16 ? (ns == 4 ? dispatch_p1<f16, 4>(p, mbw, wr, ntw, st) : dispatch_p1<f16, 3>(p, mbw, wr, ntw, st))
                                          : (ns == 4 ? dispatch_p1<float, 4>(p, mbw, wr, ntw, st) : dispatch_p1<float, 3>(p, mbw, wr, ntw, st));
      if (okp) {
        SQDET_CHECK_HIP(hipGetLastError());
        *handled = true;
        return SQDET_OK;
      }
    }
  }
  a.stage_chunks = g.nchunk < 4 ? g.nchunk : 4;
  const size_t lds = (size_t)a.stage_chunks * 16 * mbw * wr * 64;   // <= 64 KiB
  const bool ok = dtype == SQDET_F16 ? dispatch_g1<f16>(a, mbw, wr, ntw, lds, st) : dispatch_g1<float>(a, mbw, wr, ntw, lds, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_G1_TIMELINE
extern "C" int sqdet_debug_g1_timeline(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_g1_tl), sizeof(unsigned long long) * count);
}
#endif
