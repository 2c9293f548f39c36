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

// FCD_PREC_F16X2 overflow word, one per device, owned by the caller (device memory)
#define FCD_MAX_DEVICES 64
static unsigned* g_overflow[FCD_MAX_DEVICES];

unsigned* fcd_overflow_flag_current(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FCD_MAX_DEVICES) return nullptr;
  return __atomic_load_n(&g_overflow[dev], __ATOMIC_RELAXED);
}

extern "C" int fcd_set_overflow_flag(int device, uint32_t* flag) {
  FCD_REQUIRE(device >= 0 && device < FCD_MAX_DEVICES, "fcd_set_overflow_flag: device %d out of range", device);
  FCD_REQUIRE(((uintptr_t)flag & 3u) == 0, "fcd_set_overflow_flag: the word must be 4-byte aligned");
  __atomic_store_n(&g_overflow[device], (unsigned*)flag, __ATOMIC_RELAXED);
  return FCD_OK;
}

extern "C" int fcd_conv_desc_bytes(void) { return (int)sizeof(FcdConvDesc); }
extern "C" int fcd_wgrad_desc_bytes(void) { return (int)sizeof(FcdWgradDesc); }

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
