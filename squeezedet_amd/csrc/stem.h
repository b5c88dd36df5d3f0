// Arguments shared by the fused-stem kernels (stem.hip: LDS conv tile; stem2.hip: in-register pool; stem3.hip: persistent).
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
};

int stem_strip_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled);
int stem_pers_launch(StemArgs a, int k, int dtype, hipStream_t st, bool* handled);   // stem3.hip: fp16, 3x3, 64 couts

}  // namespace sqdet
