// Fused network stem for gfx950, host side: conv (KSxKS, stride 2, Cin = 3) + bias + ReLU + max-pool 3x3/s2 in ONE launch --
// conv1 + pool1 of SqueezeDet (reference src/nets/squeezeDet.py:40-44: 3x3/s2 SAME + pool SAME), SqueezeDet+
// (src/nets/squeezeDetPlus.py:40-44: 7x7/s2 VALID + pool VALID) and ResNet50 (src/nets/resnet50_convDet.py:41-45: 7x7/s2,
// 64 couts, pool VALID).  Unfused, conv1 writes 188x621x64 and pool1 reads it back -- 36.4 MB of the 133 MB per image
// (fp16); fused, the conv activations never leave the CU.  Kernels: stem4.hip (persistent, lane-local pooling over three column
// phases; fp16 3x3 on images >= 523 wide: the default there since the end of round 3 -- the 32-image step 0.5254 against
// 0.5305 ms, six alternating runs on two boxes), stem3.hip (persistent, strip lanes + DPP pooling; fp16 3x3 otherwise, or
// "stem_algo" 3), stem5.hip (the float16 7x7 stems: 4-channel LDS rows, one K chunk per kernel row, no gather; round 6) and
// stem2.hip (strip kernel with the pool in registers; every other shape / dtype, or "stem_algo" 2).  (The round-1 LDS-conv-tile kernel that lived here -- conv
// tile written to LDS, pooled from LDS -- was superseded by both and is gone.)
#include "stem.h"

namespace sqdet {

// conv(k, stride 2, Cin 3) + bias + relu + maxpool(3, stride 2).  *handled=false: not eligible.
int stem_launch(const void* x, const void* w_packed, const float* bias, void* y, int n, int h, int w, int cout, int k,
                int conv_pad, int pool_pad, int dtype, int y_cstride, int y_coffset, hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0) return SQDET_OK;
  if (!((k == 3 && cout == 64) || (k == 7 && (cout == 96 || cout == 64)))) return SQDET_OK;   // SqueezeDet, SqueezeDet+, ResNet50
  const ConvGeom g = conv_geom(k, 3, cout, dtype);
  if (!g.gather || g.ngroups != 1) return SQDET_OK;
  StemArgs a;
  a.x = x; a.wp = w_packed; a.bias = bias; a.y = y;
  a.N = n; a.H = h; a.W = w;
  a.Hc = out_size(h, k, 2, conv_pad); a.Wc = out_size(w, k, 2, conv_pad);
  a.Hp = out_size(a.Hc, 3, 2, pool_pad); a.Wp = out_size(a.Wc, 3, 2, pool_pad);
  a.ptc = pad_before(h, k, 2, conv_pad); a.plc = pad_before(w, k, 2, conv_pad);
  a.ptp = pad_before(a.Hc, 3, 2, pool_pad); a.plp = pad_before(a.Wc, 3, 2, pool_pad);
  a.Cout = cout; a.nchunk = g.nchunk; a.kdim = g.kdim;
  a.tiles_x = a.tiles_y = 0;   // (set by the kernel's launcher)
  a.y_cstride = y_cstride; a.y_coffset = y_coffset;
  a.ws2 = nullptr; a.bs2 = nullptr; a.s_out = nullptr;
  if (a.Hp <= 0 || a.Wp <= 0) return SQDET_OK;
  if (tune(TUNE_STEM_ALGO) == 0 || tune(TUNE_STEM_ALGO) >= 4) {  // the phase kernel (stem4.hip; wide images only, else the kernels below)
    const int rc4 = stem_phase_launch(a, k, dtype, st, handled);
    if (rc4 != SQDET_OK || *handled) return rc4;
  }
  if (tune(TUNE_STEM_ALGO) == 0 || tune(TUNE_STEM_ALGO) >= 3) {  // default for the fp16 3x3 stem: the persistent kernel (stem3.hip)
    const int rc3 = stem_pers_launch(a, k, dtype, st, handled);
    if (rc3 != SQDET_OK || *handled) return rc3;
  }
  if (tune(TUNE_STEM_ALGO) == 0) {                               // the float16 7x7 stems: one K chunk per kernel row (stem5.hip)
    const int rc5 = stem_k7_launch(a, k, dtype, st, handled);
    if (rc5 != SQDET_OK || *handled) return rc5;
  }
  // otherwise the in-register-pool strip kernel (stem2.hip); channel strides it cannot store with 16-byte vectors: not
  // handled (the caller runs conv and pool apart)
  return stem_strip_launch(a, k, dtype, st, handled);
}

// conv1 + pool1 + the next layer's squeeze1x1 (64 -> 16 couts) in one launch: only the squeeze tensor [n, Hp, Wp, 16] is written
// (persistent fp16 3x3 stem only).  *handled = false: not eligible.
int stem_squeeze_launch(const void* x, const void* w_packed, const float* bias, const void* ws2_packed, const float* bs2,
                        void* s_out, int n, int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype,
                        hipStream_t st, bool* handled) {
  *handled = false;
  if (conv_algo() != 0 || (tune(TUNE_STEM_ALGO) != 0 && tune(TUNE_STEM_ALGO) < 3) || k != 3 || cout != 64 || s2 != 16 || dtype != SQDET_F16) return SQDET_OK;
  const ConvGeom g = conv_geom(k, 3, cout, dtype), gs = conv_geom(1, cout, s2, dtype);
  if (!g.gather || g.ngroups != 1 || gs.gather || gs.nchunk != 2 || gs.nt != 1 || gs.ngroups != 1) return SQDET_OK;
  StemArgs a;
  a.x = x; a.wp = w_packed; a.bias = bias; a.y = nullptr;
  a.N = n; a.H = h; a.W = w;
  a.Hc = out_size(h, k, 2, conv_pad); a.Wc = out_size(w, k, 2, conv_pad);
  a.Hp = out_size(a.Hc, 3, 2, pool_pad); a.Wp = out_size(a.Wc, 3, 2, pool_pad);
  a.ptc = pad_before(h, k, 2, conv_pad); a.plc = pad_before(w, k, 2, conv_pad);
  a.ptp = pad_before(a.Hc, 3, 2, pool_pad); a.plp = pad_before(a.Wc, 3, 2, pool_pad);
  a.Cout = cout; a.nchunk = g.nchunk; a.kdim = g.kdim;
  a.tiles_x = a.tiles_y = 0;
  a.y_cstride = cout; a.y_coffset = 0;
  a.ws2 = ws2_packed; a.bs2 = bs2; a.s_out = s_out;
  if (a.Hp <= 0 || a.Wp <= 0) return SQDET_OK;
  if (tune(TUNE_STEM_ALGO) == 0 || tune(TUNE_STEM_ALGO) >= 4) {
    const int rc4 = stem_phase_launch(a, k, dtype, st, handled);
    if (rc4 != SQDET_OK || *handled) return rc4;
  }
  return stem_pers_launch(a, k, dtype, st, handled);
}

bool stem_squeeze_eligible(int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype, int n) {
  if (conv_algo() != 0 || (tune(TUNE_STEM_ALGO) != 0 && tune(TUNE_STEM_ALGO) < 3) || k != 3 || cout != 64 || s2 != 16 || dtype != SQDET_F16) return false;
  const int plc = pad_before(w, k, 2, conv_pad);
  const int hc = out_size(h, k, 2, conv_pad), wc = out_size(w, k, 2, conv_pad);
  const int hp = out_size(hc, 3, 2, pool_pad), wp = out_size(wc, 3, 2, pool_pad);
  if (hp <= 0 || wp <= 0 || w % 2 != 0 || plc % 2 != 0) return false;
  if ((size_t)h * w * 6 >= (1ull << 31) || w * 6 < 2 * 44 * 16) return false;          // as stem_pers_launch
  if ((size_t)n * hp * wp * cout * 2 >= (1ull << 31)) return false;
  const ConvGeom gs = conv_geom(1, cout, s2, dtype);
  return !gs.gather && gs.nchunk == 2 && gs.nt == 1 && gs.ngroups == 1;
}

}  // namespace sqdet

extern "C" int sqdet_stem_conv_pool_squeeze_fwd(const void* x, const void* w_packed, const float* bias, const void* w_next_s_packed,
                                                const float* b_next_s, void* sq_out, int n, int h, int w, int cout, int k,
                                                int conv_pad_mode, int pool_pad_mode, int next_s, int dtype, sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(x && w_packed && bias && w_next_s_packed && b_next_s && sq_out, "stem_squeeze: null pointer");
  bool handled = false;
  const int rc = stem_squeeze_launch(x, w_packed, bias, w_next_s_packed, b_next_s, sq_out, n, h, w, cout, k, conv_pad_mode,
                                     pool_pad_mode, next_s, dtype, as_stream(stream), &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "stem_squeeze: needs the float16 3x3 / 64-cout stem of even width followed by a 64 -> 16 squeeze1x1");
  return SQDET_OK;
}

extern "C" int sqdet_stem_conv_pool_squeeze_supported(int h, int w, int cout, int k, int conv_pad_mode, int pool_pad_mode,
                                                      int next_s, int dtype, int n) {
  return sqdet::stem_squeeze_eligible(h, w, cout, k, conv_pad_mode, pool_pad_mode, next_s, dtype, n) ? 1 : 0;
}

// C ABI: the fused stem as its own entry point (see include/sqdet.h).
extern "C" int sqdet_stem_conv_pool_fwd(const void* x, const void* w_packed, const float* bias, void* y, int n, int h,
                                        int w, int cout, int k, int conv_pad_mode, int pool_pad_mode, int dtype,
                                        sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(x && w_packed && bias && y, "stem: null pointer");
  SQDET_REQUIRE(dtype == SQDET_F16 || dtype == SQDET_F32, "stem: bad dtype");
  bool handled = false;
  const int saved = conv_algo();
  SQDET_UNSUPPORTED(saved != 0, "stem: fused kernel disabled by conv_algo=generic");
  int rc = stem_launch(x, w_packed, bias, y, n, h, w, cout, k, conv_pad_mode, pool_pad_mode, dtype, cout, 0,
                       as_stream(stream), &handled);
  if (rc != SQDET_OK) return rc;
  SQDET_UNSUPPORTED(!handled, "stem: only (k=3, cout=64) and (k=7, cout=96 or 64) stems are fused");
  return SQDET_OK;
}
