// Shared host/device helpers for libsqdet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "sqdet.h"

namespace sqdet {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define SQDET_CHECK_HIP(expr)                                         \
  do {                                                                \
    hipError_t e__ = (expr);                                          \
    if (e__ != hipSuccess) return ::sqdet::hip_fail(e__, #expr);      \
  } while (0)

#define SQDET_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      ::sqdet::set_error(__VA_ARGS__);  \
      return SQDET_EINVAL;              \
    }                                   \
  } while (0)

#define SQDET_UNSUPPORTED(cond, ...)    \
  do {                                  \
    if (cond) {                         \
      ::sqdet::set_error(__VA_ARGS__);  \
      return SQDET_EUNSUPPORTED;        \
    }                                   \
  } while (0)

// TF output size / SAME padding (extra cell goes bottom/right).
inline int out_size(int n, int k, int s, int pad_mode) {
  return pad_mode == SQDET_PAD_SAME ? (n + s - 1) / s : (n - k) / s + 1;
}
inline int pad_before(int n, int k, int s, int pad_mode) {
  if (pad_mode != SQDET_PAD_SAME) return 0;
  int o = (n + s - 1) / s;
  int tot = (o - 1) * s + k - n;
  if (tot < 0) tot = 0;
  return tot / 2;
}
// One-time, PER-DEVICE launcher setup (hipFuncSetAttribute of the dynamic-LDS ceiling, occupancy queries): function attributes belong to
// the device's copy of the code object, so a process that drives several GPUs must set them on each.  `static PerDevice once;` in a
// launcher; `once.run(f)` calls f (returns hipError_t) the first time the CURRENT device is seen (up to 64 devices) and records
// success; racing first calls may both run f -- it is idempotent.
struct PerDevice {
  std::atomic<uint64_t> done{0};
  template <class F>
  hipError_t run(F&& f) {
    int d = 0;
    (void)hipGetDevice(&d);
    const uint64_t bit = 1ull << (d & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    const hipError_t e = f();
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
  }
};
// Compute units of the current device, rounded down to a multiple of 8 (the persistent kernels deal tiles XCD-major: blockIdx & 7),
// at least 8; 256 on an MI355X in SPX mode.  Cached per device.
int cu_count();
int current_device();

inline size_t dtype_size(int dtype) { return dtype == SQDET_F16 ? 2 : 4; }
inline hipStream_t as_stream(sqdet_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

typedef _Float16 f16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

// ---- conv weight packing geometry (shared by pack, conv and the executor) ----
struct ConvGeom {
  int kg;       // elements per 16-byte lane chunk (8 f16 / 4 f32)
  int kc;       // channels per chunk-step = 4 lane groups * kg
  bool gather;  // Cin not a multiple of kg: im2col-gather path, K' = k*k*cin
  int kdim;     // per-tap channel count the packer sees (cin, or k*k*cin when gather)
  int taps;     // k*k, or 1 when gather
  int nchunk;   // chunk-steps per tap
  int steps;    // taps * nchunk
  int nt;       // 16-wide cout tiles per group
  int group;    // couts per group = 16*nt
  int ngroups;
};
ConvGeom conv_geom(int k, int cin, int cout, int dtype);

}  // namespace sqdet
