// Error reporting and device checks of libfcd_hip.so.
#include "fcd_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void fcd_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* fcd_last_error(void) { return g_err; }
extern "C" int fcd_abi_version(void) { return FCD_ABI_VERSION; }

extern "C" int fcd_init(int device) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fcd_set_error("fcd_init: no HIP device (%s)", hipGetErrorString(e));
    return FCD_ENODEVICE;
  }
  if (device < 0 || device >= n) {
    fcd_set_error("fcd_init: device %d out of range (%d devices)", device, n);
    return FCD_EINVAL;
  }
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    fcd_set_error("fcd_init: %s", hipGetErrorString(e));
    return (int)e;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    fcd_set_error("fcd_init: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    return FCD_ENODEVICE;
  }
  e = hipSetDevice(device);
  if (e != hipSuccess) {
    fcd_set_error("fcd_init: %s", hipGetErrorString(e));
    return (int)e;
  }
  return fcd_noalloc_scratch_init(device);
}
