// Host side of the LDS-staged 3x3 convolution (kernel template: conv3x3_tile.h): eligibility, tile / wave layout choice and
// the cout-split instantiations; the split-K (ConvDet) ones are compiled in convdet.hip.
#include "conv3x3_tile.h"

namespace sqdet {

template <typename T, int MT>
static bool dispatch_ntw(const TileArgs& a, int ntw, int grid_y, size_t lds, hipStream_t st) {
  switch (ntw) {
    case 1: launch_tile<T, MT, 1>(a, grid_y, lds, st); return true;
    case 2: launch_tile<T, MT, 2>(a, grid_y, lds, st); return true;
    case 3: launch_tile<T, MT, 3>(a, grid_y, lds, st); return true;
    case 4: launch_tile<T, MT, 4>(a, grid_y, lds, st); return true;
    case 5: launch_tile<T, MT, 5>(a, grid_y, lds, st); return true;
    case 6: launch_tile<T, MT, 6>(a, grid_y, lds, st); return true;
    default: return false;
  }
}

template <typename T>
static bool dispatch_tile(const TileArgs& a, int mt, int ntw, bool splitk, int grid_y, size_t lds, hipStream_t st) {
  if (splitk) {
    if (ntw != 5) return false;
    return convdet_tile_launch(a, sizeof(T) == 2 ? SQDET_F16 : SQDET_F32, st) == SQDET_OK;
  }
  if (mt == 8) {  // one wave = all 8 tile rows x a slice of a group (4 waves along the cout tiles)
    switch (ntw) {
      case 1: launch_tile<T, 8, 1>(a, grid_y, lds, st); return true;
      case 2: launch_tile<T, 8, 2>(a, grid_y, lds, st); return true;
      case 3: launch_tile<T, 8, 3>(a, grid_y, lds, st); return true;
      case 4: launch_tile<T, 8, 4>(a, grid_y, lds, st); return true;
      default: return false;
    }
  }
  return mt == 4 ? dispatch_ntw<T, 4>(a, ntw, grid_y, lds, st) : dispatch_ntw<T, 2>(a, ntw, grid_y, lds, st);
}

// Eligibility + configuration.  *handled = false means "use the generic kernels".
int conv3x3_tile_launch(const ConvArgs& c, const ConvGeom& g, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (c.k != 3 || c.stride != 1 || c.pt != 1 || c.pl != 1 || g.gather) return SQDET_OK;
  if (c.Ho != c.H || c.Wo != c.W) return SQDET_OK;
  const int esz = dtype == SQDET_F16 ? 2 : 4;
  if ((c.Cin * esz) % 16 != 0 || (c.x_cstride * esz) % 16 != 0 || (c.x_coffset * esz) % 16 != 0) return SQDET_OK;
  const bool plain = c.x_cstride == c.Cin && c.x_coffset == 0 && !c.accum && c.bias && !c.relu_of;
  TileArgs a;
  a.c = c;
  a.tiles_x = (c.W + TCOLS - 1) / TCOLS;
  a.tiles_y = (c.H + TROWS - 1) / TROWS;
  a.nt_pack = g.nt;
  a.total_tiles = g.nt * g.ngroups;
  a.nchunk = g.nchunk;
  a.pieces = c.Cin * esz / 16;
  a.stage_chunks = g.nchunk;
  a.wp1 = nullptr; a.bias1 = nullptr; a.y_coffset1 = 0;
  a.dma = 0; a.chunk_pitch = CHUNK_BYTES;
  if ((long)c.N * a.tiles_x * a.tiles_y > 0x7fffffffL) return SQDET_OK;
  if (a.total_tiles * 16 > 1024) return SQDET_OK;     // (the kernel keeps the bias in LDS: up to 1024 couts)

  int ntw = g.nt, mt = 2;
  bool splitk = false;
  size_t lds = 0;
  int grid_y = 1;
  const long x_bytes = (long)c.N * c.H * c.W * c.Cin * esz;
  a.x_bytes = (unsigned)(x_bytes < 0x7fffffffL ? x_bytes : 0);
  if (plain && g.nt == 5 && g.ngroups == 1 && g.nchunk >= 8 && g.nchunk % 4 == 0 && x_bytes < 0x7fffffffL) {
    // ConvDet-like: few couts, deep K -> split K over the 4 waves, 4 chunks per stage
    splitk = true;
    mt = 8;
  } else {
    // a wave owns one whole packed group; 2 groups per workgroup (waves 2 rows x 2 groups) when
    // there are several, else the 4 waves split the 8 tile rows
    // Measured on MI355X (tools/kbench.py, batch 32): MFMA-bound deep-K layers (fire10/11: 96 -> 384)
    // want MT = 8 (one A fragment from L1 feeds 8 MFMAs); the mid layers want whole-group waves.
    // up to 6 K-chunks (192 fp16 channels) the whole halo tile is resident; deeper K (SqueezeDet+ fire6-11: 9 / 12 chunks,
    // ResNet50 res4 / res5: 8 / 16) is walked in stages of 4 chunks = 46 KB, three workgroups per CU hiding each other's staging
    a.stage_chunks = g.nchunk <= 6 ? g.nchunk : 4;
    // staging by LDS-DMA where the input fits 32-bit buffer offsets ("dbg" 97: through registers, the rounds 1-5 form -- A/B)
    const long xs_bytes = (long)c.N * c.H * c.W * c.x_cstride * esz - (long)c.x_coffset * esz;
    if (xs_bytes > 0 && xs_bytes < 0x7fffffffL && tune(TUNE_DBG) != 97) {
      a.dma = 1; a.chunk_pitch = 12288;
      a.x_bytes = (unsigned)xs_bytes;
    }
    // a launch with at most one workgroup per CU gains nothing from small stages: the whole tile resident (<= 12 chunks = 144 KiB),
    // ONE burst of DMA blocks and no stage hand-overs ("dbg" 98: not)
    {
      const int wc_ = g.ngroups >= 2 ? 2 : 1;
      const long wgs_ = (long)c.N * a.tiles_x * a.tiles_y * ((g.ngroups + wc_ - 1) / wc_);
      const bool mt8_ = (g.nt == 6 && g.ngroups >= 2 && g.nchunk >= 3) || (g.nt == 4 && g.ngroups == 1) ||
                        (g.nt == 4 && g.ngroups % 4 == 0 && g.nchunk >= 4 && (long)c.N * a.tiles_x * a.tiles_y * (g.ngroups / 4) >= 384);
      if (a.dma && !mt8_ && g.nchunk > 6 && g.nchunk <= 12 && wgs_ <= cu_count() && tune(TUNE_DBG) != 98) a.stage_chunks = g.nchunk;
    }
    lds = (size_t)a.stage_chunks * a.chunk_pitch + (size_t)a.total_tiles * 16 * 4;   // + the bias
    if (g.nt == 6 && g.ngroups >= 2 && g.nchunk >= 3) {
      mt = 8; ntw = 3;
      grid_y = (a.total_tiles + 11) / 12;
    } else if (g.nt == 4 && g.ngroups == 1) {
      mt = 8; ntw = 1;
      grid_y = 1;
    } else if (g.nt == 4 && g.ngroups % 4 == 0 && g.nchunk >= 4 && (long)c.N * a.tiles_x * a.tiles_y * (g.ngroups / 4) >= 384) {
      // whole-tile waves, four cout groups per workgroup: an A fragment feeds 8 MFMAs instead of 4 (SqueezeDet+ fire8
      // expand3x3 110 -> 98 us = 1.0 PF/s); only with enough tiles -- on the 22x76 maps at batch 8 the 120 workgroups of this
      // layout take 76 us against 43
      mt = 8; ntw = 4;
      grid_y = g.ngroups / 4;
    } else {
      mt = g.ngroups >= 2 ? 4 : 2;
      const int wc = mt == 4 ? 2 : 1;
      grid_y = (g.ngroups + wc - 1) / wc;
    }
  }
  if (c.scores && !splitk) return SQDET_OK;   // only the ConvDet kernel has the score epilogue (the caller reports UNSUPPORTED)
  const bool ok = dtype == SQDET_F16 ? dispatch_tile<f16>(a, mt, ntw, splitk, grid_y, lds, st)
                                     : dispatch_tile<float>(a, mt, ntw, splitk, grid_y, lds, st);
  if (!ok) return SQDET_OK;
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

// ---- expand1x1 || expand3x3 of a fire module in ONE launch of the tile kernel's PAIR form (float16; SqueezeDet+'s squeeze depths
// 192 / 384, nets/squeezeDetPlus.py:46-73,81-106) ----
// Shapes: e1 == e3 = a multiple of 64 in two or more packed groups of 4 tiles (the <4, 4> wave layout), and the WHOLE halo tile
// resident: up to 6 K chunks (S <= 192) anywhere -- two workgroups per CU as for the plain conv --, up to 12 (S <= 384: 135 KiB,
// one workgroup per CU) only where the launch has at most one workgroup per CU anyway (the 22 x 76 maps at batch 8: 240).
// ("dbg" 93: never; 94: the 12-chunk form at any size -- tests)
bool conv3x3_pair_eligible(int n, int h, int w, int s, int e1, int e3, int dtype) {
  if (conv_algo() != 0 || tune(TUNE_DBG) == 93 || dtype != SQDET_F16 || e1 != e3 || e1 % 64 != 0 || s % 8 != 0) return false;
  const ConvGeom g1 = conv_geom(1, s, e1, dtype), g3 = conv_geom(3, s, e3, dtype);
  if (g1.gather || g3.gather || g3.nt != 4 || g1.nt != 4 || g3.ngroups < 2 || g1.ngroups != g3.ngroups || g1.nchunk != g3.nchunk) return false;
  if (g3.nchunk <= 6) return true;
  if (g3.nchunk > 12) return false;
  const long wgs = (long)n * ((w + TCOLS - 1) / TCOLS) * ((h + TROWS - 1) / TROWS) * ((g3.ngroups + 1) / 2);
  return wgs <= cu_count() || tune(TUNE_DBG) == 94;
}

int conv3x3_pair_launch(const void* x, const void* w3, const float* b3, const void* w1, const float* b1, void* y, int n, int h, int w,
                        int s, int e1, int e3, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (!conv3x3_pair_eligible(n, h, w, s, e1, e3, dtype)) return SQDET_OK;
  const ConvGeom g = conv_geom(3, s, e3, dtype);
  const long px = (long)n * h * w;
  if (px * (e1 + e3) * 2 >= (1L << 31) || px > (1L << 30)) return SQDET_OK;
  TileArgs a;
  ConvArgs& c = a.c;
  c.x = x; c.wp = w3; c.bias = b3; c.y = y;
  c.N = n; c.H = h; c.W = w; c.Cin = s; c.Cout = e3; c.k = 3; c.stride = 1; c.pt = 1; c.pl = 1; c.Ho = h; c.Wo = w;
  c.P = (int)px; c.ntiles = 0;
  c.nchunk = g.nchunk; c.steps = g.steps; c.ngroups = g.ngroups;
  c.y_cstride = e1 + e3; c.y_coffset = e1; c.relu = 1;
  c.x_cstride = s; c.x_coffset = 0; c.accum = 0; c.res = nullptr; c.relu_of = nullptr;
  c.scores = nullptr; c.score_apg = 0; c.score_classes = 0;
  a.tiles_x = (w + TCOLS - 1) / TCOLS;
  a.tiles_y = (h + TROWS - 1) / TROWS;
  a.nt_pack = g.nt;
  a.total_tiles = g.nt * g.ngroups;
  a.nchunk = g.nchunk;
  a.pieces = s * 2 / 16;
  a.stage_chunks = g.nchunk;                // resident
  a.x_bytes = (unsigned)(px * s * 2);
  a.wp1 = w1; a.bias1 = b1; a.y_coffset1 = 0;
  a.dma = tune(TUNE_DBG) != 97 ? 1 : 0;
  a.chunk_pitch = a.dma ? 12288 : CHUNK_BYTES;
  const size_t lds = (size_t)g.nchunk * a.chunk_pitch + (size_t)a.total_tiles * 16 * 4 * 2;   // + both biases
  launch_tile<f16, 4, 4, true>(a, (g.ngroups + 1) / 2, lds, st);
  SQDET_CHECK_HIP(hipGetLastError());
  *handled = true;
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_C3_TIMELINE
extern "C" int sqdet_debug_c3_timeline(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_c3_tl), sizeof(unsigned long long) * count);
}
#endif
