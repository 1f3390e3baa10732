// Library-level entry points: version, error string, device check, workspace sizing.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int dm_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" int dm_version(void) { return 1; }
extern "C" const char* dm_last_error(void) { return g_err; }

extern "C" int dm_device_check(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return dm_fail(DM_E_DEVICE, "hipGetDevice failed (no HIP device visible)");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return dm_fail(DM_E_DEVICE, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return dm_fail(DM_E_DEVICE, "device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
  return DM_OK;
}

// Scratch sizing: the largest transient of any fused operator at this shape (see DESIGN.md "HBM layout").
//  - conv decoder layer 3 column matrix: rows = N*13*13, cols = 36*cnn_depth (k6 x k6 x 48 at depth 48)
//  - conv encoder layer 2 patch matrix: rows = N*14*14, cols = 16*cnn_depth
//  - split-K partials: bounded by 64 MiB
extern "C" size_t dm_workspace_bytes(const dm_shape* s) {
  if (!s) return 0;
  const size_t N = (size_t)s->T * s->B * (s->I > 0 ? s->I : 1);
  const size_t d = (size_t)s->cnn_depth;
  size_t conv = 0;
  // decoder column matrices (fwd) / patch matrices (bwd), image 64: spatial 1->5->13->30->64
  const size_t dec2 = N * 25 * (25 * 2 * d);     // layer 2: rows N*5*5, cols k5*k5*(2d)
  const size_t dec3 = N * 169 * (36 * d);        // layer 3: rows N*13*13, cols k6*k6*d
  const size_t dec4 = N * 900 * (36 * (size_t)s->img_ch);
  const size_t enc1 = N * 961 * (16 * (size_t)s->img_ch);
  const size_t enc2 = N * 196 * (16 * d);
  const size_t enc3 = N * 36 * (16 * 2 * d);
  conv = dec2;
  if (dec3 > conv) conv = dec3;
  if (dec4 > conv) conv = dec4;
  if (enc1 > conv) conv = enc1;
  if (enc2 > conv) conv = enc2;
  if (enc3 > conv) conv = enc3;
  // two column-sized scratch matrices (column matrix + its gradient) + split-K partial region + slack
  size_t floats = 2 * conv + (size_t)16 * 1024 * 1024 + (size_t)(s->H + 2) * N * 64 + (1u << 20);
  return floats * sizeof(float);
}
