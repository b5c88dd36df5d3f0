// Probe of ds_read_b64_tr_b16 (gfx950): which LDS element lands in lane l, element j, when every lane passes its own
// 8-byte-aligned address.  Lane a of each 16-lane group is pointed at row a/4, columns 4*(a%4).. of a [4][16] block of
// a row-major image with a 40-element pitch; group g's block starts 8 rows further down.  Prints out[l][j] as (row, col).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/tr_probe tools/microbench/tr_read_probe.hip && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 40;
__global__ void k(short* out) {
  __shared__ short lds[64 * PITCH];
  for (int i = threadIdx.x; i < 64 * PITCH; i += 64) lds[i] = (short)((i / PITCH) * 100 + (i % PITCH));
  __syncthreads();
  const int l = threadIdx.x, a = l & 15, g = l >> 4;
  auto p = (__attribute__((address_space(3))) s4*)(lds + (8 * g + (a >> 2)) * PITCH + 4 * (a & 3));
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d; short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int j = 0; j < 4; ++j) printf(" (%d,%d)", h[l * 4 + j] / 100, h[l * 4 + j] % 100);
    printf("\n");
  }
  return 0;
}
