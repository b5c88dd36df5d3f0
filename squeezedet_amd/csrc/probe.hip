// Hardware self-test: observes the MFMA fragment layouts the conv kernels rely on
// (guide: cdna_hip_programming.md section 3) so a wrong assumption shows up as a clear
// test failure instead of silently transposed activations.
//
// For each shape we run ONE MFMA with A = "row selector" and B = "col selector" encodings
// chosen so that D[i][j] = 64*i + j + 1 exactly, then report for every (lane, reg) the
// observed (row, col).  host_out layout: [shape(2)][lane(64)][reg(4)][2].
#include "common.h"

namespace sqdet {

// shape 0: mfma_f32_16x16x32_f16, assumed operand layout: A lane l holds row i=l&15,
// k-slots 8*(l>>4)..+7 ; B lane l holds col j=l&15, same k-slots.
// Encoding: k-slot 0 carries (64*i+1) in A and 1 in B; k-slot 1 carries 1 in A and j in B.
// Then D[i][j] = (64i+1)*1 + 1*j exactly (values < 2048 are exact in f16).
__global__ void probe_kernel(float* out) {
  const int l = threadIdx.x;
  const int i = l & 15, g = l >> 4;
  {
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    if (g == 0) {
      a[0] = (f16)(64 * i + 1); b[0] = (f16)1;
      a[1] = (f16)1;            b[1] = (f16)i;  // i == this lane's column index for B
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(0 * 64 + l) * 4 + r] = acc[r];
  }
  {
    // shape 1: mfma_f32_16x16x4f32: A lane l = A[i=l&15][k=l>>4], B lane l = B[k=l>>4][j=l&15]
    float a = 0.f, b = 0.f;
    if (g == 0) { a = (float)(64 * i + 1); b = 1.f; }
    if (g == 1) { a = 1.f; b = (float)i; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(1 * 64 + l) * 4 + r] = acc[r];
  }
}

// ---- box calibration (bench.py: `box_mfma_tflops`, `box_copy_gbs`): two fixed ~1 ms microkernels whose only purpose is to tell a slow
// box (profiles/r03_slowbox_*: every MFMA-heavy launch 20-27 % slower at the same reported clock) from a regression of the build.
// 512 workgroups x 4 waves (two per SIMD), `iters` x 8 independent 16x16x32 float16 MFMAs per wave on non-trivial operands
// (zero operands clock ~19 % higher: MI355X_MICROARCH.md, DVFS give-back).
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* out, int iters) {
  const int l = threadIdx.x & 63;
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (f16)(0.01f * (float)((l * 7 + i * 3) % 13 - 6));
    b[i] = (f16)(0.02f * (float)((l * 5 + i) % 11 - 5));
  }
  f32x4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) s += acc[k];
  out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// Round 5: the same loop in the shapes that settle WHAT the box's ceiling is (the 16x16x32 figure reads half of the
// guide's 2.5 PF/s).  SHAPE 0 = 16x16x32 (8 independent accumulators of 4 registers), SHAPE 1 = 32x32x16 (4 independent
// accumulators of 16 registers: the instruction the guide's 2495 TF/s was measured with); one wave per SIMD per workgroup, the
// launcher makes `waves_per_simd` workgroups per CU co-resident; zero != 0: all-zero operands (the chip's power management gives
// clock back on them -- MI355X_MICROARCH.md "DVFS give-back" -- so zero vs non-trivial operands on the SAME instruction stream
// separates "the instruction stream cannot issue faster" from "the box is power-limited").  Wave 0 of every workgroup brackets its
// loop with s_memtime (shader cycles) and s_memrealtime (constant 100 MHz): ticks[2 * wg] / ticks[2 * wg + 1] give the EFFECTIVE
// shader clock and the cycles per MFMA without a profiler (rocprofv3's GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES of the same
// launches: tools/calib_pmc.sh -> profiles/r05_box_calibration.txt).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ __launch_bounds__(256) void calib_mfma2_kernel(float* out, unsigned long long* ticks, int iters, int zero) {
  const int l = threadIdx.x & 63;
  f16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = zero ? (f16)0.f : (f16)(0.01f * (float)((l * 7 + i * 3) % 13 - 6));
    b[i] = zero ? (f16)0.f : (f16)(0.02f * (float)((l * 5 + i) % 11 - 5));
  }
  float r = 0.f;
  unsigned long long c0, c1, w0, w1;
  if constexpr (SHAPE == 0) {
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) r += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
  } else {
    f32x16 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;
    c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k], 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int e = 0; e < 16; ++e) r += acc[k][e];
    c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
  }
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = r;
  if (threadIdx.x == 0) {
    ticks[2 * (size_t)blockIdx.x] = c1 - c0;
    ticks[2 * (size_t)blockIdx.x + 1] = w1 - w0;
  }
}

__global__ __launch_bounds__(256) void calib_copy_kernel(const i32x4* __restrict__ src, i32x4* __restrict__ dst, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

}  // namespace sqdet

extern "C" int sqdet_calib_mfma(float* scratch, size_t scratch_floats, int iters, double* flops, sqdet_stream_t stream) {
  using namespace sqdet;
  constexpr int GRID = 512;
  SQDET_REQUIRE(scratch && scratch_floats >= (size_t)GRID * 256 && iters > 0 && flops, "calib_mfma: scratch >= 131072 floats, iters > 0");
  hipLaunchKernelGGL(calib_mfma_kernel, dim3(GRID), dim3(256), 0, as_stream(stream), scratch, iters);
  SQDET_CHECK_HIP(hipGetLastError());
  *flops = (double)GRID * 4.0 * (double)iters * 8.0 * (2.0 * 16 * 16 * 32);
  return SQDET_OK;
}

extern "C" int sqdet_calib_mfma2(float* scratch, size_t scratch_floats, unsigned long long* ticks, size_t ticks_count, int iters, int shape,
                                 int waves_per_simd, int zero_operands, double* flops, int* workgroups, sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(scratch && ticks && flops && workgroups && iters > 0, "calib_mfma2: null pointer / iters");
  SQDET_REQUIRE((shape == 0 || shape == 1) && waves_per_simd >= 1 && waves_per_simd <= 8, "calib_mfma2: shape 0 | 1, 1..8 waves per SIMD");
  const int grid = cu_count() * waves_per_simd;
  SQDET_REQUIRE(scratch_floats >= (size_t)grid * 256 && ticks_count >= (size_t)grid * 2, "calib_mfma2: scratch >= %d floats, ticks >= %d", grid * 256, grid * 2);
  if (shape == 0) hipLaunchKernelGGL(calib_mfma2_kernel<0>, dim3(grid), dim3(256), 0, as_stream(stream), scratch, ticks, iters, zero_operands);
  else hipLaunchKernelGGL(calib_mfma2_kernel<1>, dim3(grid), dim3(256), 0, as_stream(stream), scratch, ticks, iters, zero_operands);
  SQDET_CHECK_HIP(hipGetLastError());
  *flops = (double)grid * 4.0 * (double)iters * (shape == 0 ? 8.0 * (2.0 * 16 * 16 * 32) : 4.0 * (2.0 * 32 * 32 * 16));
  *workgroups = grid;
  return SQDET_OK;
}

extern "C" int sqdet_calib_copy(const void* src, void* dst, size_t bytes, sqdet_stream_t stream) {
  using namespace sqdet;
  SQDET_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0, "calib_copy: non-null pointers, bytes a multiple of 16");
  hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 8), dim3(256), 0, as_stream(stream), reinterpret_cast<const i32x4*>(src),
                     reinterpret_cast<i32x4*>(dst), bytes / 16);
  SQDET_CHECK_HIP(hipGetLastError());
  return SQDET_OK;
}

extern "C" int sqdet_probe_mfma_layout(int32_t* host_out, int capacity) {
  using namespace sqdet;
  SQDET_REQUIRE(host_out && capacity >= 2 * 64 * 4 * 2, "probe: need capacity >= 1024 int32");
  float* d = nullptr;
  SQDET_CHECK_HIP(hipMalloc(&d, 2 * 64 * 4 * sizeof(float)));
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, d);
  float h[2 * 64 * 4];
  hipError_t e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  SQDET_CHECK_HIP(e);
  for (int s = 0; s < 2; ++s)
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        const int v = (int)h[(s * 64 + l) * 4 + r] - 1;  // 64*row + col
        host_out[((s * 64 + l) * 4 + r) * 2 + 0] = v / 64;
        host_out[((s * 64 + l) * 4 + r) * 2 + 1] = v % 64;
      }
  return SQDET_OK;
}
