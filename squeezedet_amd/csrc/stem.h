// Arguments shared by the fused-stem kernels (stem2.hip: strip kernel, pool in registers; stem3.hip: persistent); stem.hip is
// their host side.
#pragma once
#include "conv_common.h"

namespace sqdet {

struct StemArgs {
  const void* x;
  const void* wp;
  const float* bias;
  void* y;
  int N, H, W;          // input
  int Hc, Wc;           // conv output
  int Hp, Wp;           // pooled output
  int ptc, plc;         // conv pad before (top, left)
  int ptp, plp;         // pool pad before
  int Cout, nchunk, kdim;
  int tiles_x, tiles_y;
  int y_cstride, y_coffset;
  // stem3.hip, squeeze form: the pooled pixels are not stored; the NEXT layer's squeeze1x1 (64 -> S2 = 16 couts: fire2's) runs on
  // them in registers and only that tensor [n, Hp, Wp, 16] is written
  const void* ws2;      // packed squeeze kernel (standard fragment order, 2 K-chunks x 1 tile), NULL = plain stem
  const float* bs2;
  void* s_out;
};

int stem_strip_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled);
int stem_pers_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled);   // stem3.hip: fp16, 3x3, 64 couts
int stem_phase_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled);  // stem4.hip: the same shapes, images >= 523 wide
int stem_k7_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled);     // stem5.hip: fp16 7x7 stems (SqueezeDet+, ResNet50)
bool stem_squeeze_eligible(int h, int w, int cout, int k, int conv_pad, int pool_pad, int s2, int dtype, int n);

}  // namespace sqdet
