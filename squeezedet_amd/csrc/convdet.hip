// The split-K form of conv3x3_tile (conv3x3_tile.h) = the ConvDet head (3x3 / SAME, 768 -> 72 couts; reference
// src/nets/squeezeDet.py:76-79, src/nn_skeleton.py:471-563), in a translation unit of its own: it is the one kernel here
// that needs more than 256 registers per lane (160 accumulators + three weight sets + two B sets + the next stage's
// input).  With the accumulators in VGPRs (-amdgpu-mfma-vgpr-form=1, the build's choice for every other MFMA kernel) the
// compiler parks 36 values in AGPRs and moves them back and forth inside the tap loop -- 143 v_accvgpr_* + 29 extra s_nop
// per 360 MFMAs.  Compiled with hipcc's default form the accumulators live in AGPRs, the 256 VGPRs hold everything else,
// and the loop has no register moves at all (conv12 at batch 32: 70.3 -> 69.1 us, at batch 1: 27.5 -> 25.3 us).
#include "conv3x3_tile.h"

namespace sqdet {

int convdet_tile_launch(const TileArgs& a, size_t lds, int dtype, hipStream_t st) {
  if (dtype == SQDET_F16) launch_tile<f16, 8, 5, true>(a, 1, lds, st);
  else launch_tile<float, 8, 5, true>(a, 1, lds, st);
  return SQDET_OK;
}

}  // namespace sqdet

#ifdef SQDET_FIRE_TIMING
extern "C" int sqdet_debug_convdet_timing(unsigned long long* host, int count) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(sqdet::g_cd_timing), sizeof(unsigned long long) * count);
}
#endif
