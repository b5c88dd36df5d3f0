// Probe: buffer_load_dwordx4 ... lds (LDS-DMA) with out-of-range lanes: does the LDS slot receive zeros, or keep its old bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* src, unsigned nbytes, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = -7;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<int*>(src), 0, nbytes, 0x00020000);
  const int l = threadIdx.x;
  // lanes with l % 3 == 0 are out of range
  unsigned off = (l % 3 == 0) ? 0x80000000u : (unsigned)(l * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
  // second block at +1024 with an in-range-but-past-end offset for lane 5
  unsigned off2 = (l == 5) ? nbytes : (unsigned)(l * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + 256), 16, off2, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
  int *d, *o; int h[256], ho[512];
  for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (unsigned)sizeof(h), o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    for (int e = 0; e < 4; ++e) {
      int exp = (l % 3 == 0) ? 0 : 1000 + l * 4 + e;
      if (ho[l * 4 + e] != exp) { if (bad < 8) printf("blk0 lane %d e %d got %d exp %d\n", l, e, ho[l * 4 + e], exp); ++bad; }
      int exp2 = (l == 5) ? 0 : 1000 + l * 4 + e;
      if (ho[256 + l * 4 + e] != exp2) { if (bad < 8) printf("blk1 lane %d e %d got %d exp %d\n", l, e, ho[256 + l * 4 + e], exp2); ++bad; }
    }
  }
  printf("dma_oob: %s (%d mismatches)\n", bad ? "OOB lanes do NOT read as zero" : "OOB lanes write zeros to LDS", bad);
  return bad != 0;
}
