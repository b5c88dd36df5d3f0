#include "common.h"

#include <stdarg.h>
#include <stdio.h>

namespace sqdet {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
  return SQDET_EHIP;
}

int current_device() {
  int d = 0;
  (void)hipGetDevice(&d);
  return d;
}

int cu_count() {
  static std::atomic<int> cached[64];
  const int d = current_device() & 63;
  int n = cached[d].load(std::memory_order_relaxed);
  if (n == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, current_device()) != hipSuccess || v < 8) {
      (void)hipGetLastError();
      v = 256;
    }
    n = v / 8 * 8;
    cached[d].store(n, std::memory_order_relaxed);
  }
  return n;
}

ConvGeom conv_geom(int k, int cin, int cout, int dtype) {
  ConvGeom g;
  g.kg = dtype == SQDET_F16 ? 8 : 4;
  g.kc = 4 * g.kg;
  g.gather = (cin % g.kg) != 0;
  g.kdim = g.gather ? k * k * cin : cin;
  g.taps = g.gather ? 1 : k * k;
  g.nchunk = (g.kdim + g.kc - 1) / g.kc;
  g.steps = g.taps * g.nchunk;
  int t16 = (cout + 15) / 16;
  int best = 1, best_pad = t16;
  for (int nt = 1; nt <= 6; ++nt) {
    int pad = (t16 + nt - 1) / nt * nt;
    if (pad < best_pad || (pad == best_pad && nt > best)) {
      best = nt;
      best_pad = pad;
    }
  }
  g.nt = best;
  g.group = 16 * best;
  g.ngroups = best_pad / best;
  return g;
}

}  // namespace sqdet

extern "C" const char* sqdet_version(void) { return "squeezedet_amd 0.1 (gfx950)"; }
extern "C" const char* sqdet_last_error(void) { return sqdet::g_err; }
