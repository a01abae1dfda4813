// Runtime of libsynthray.so: device selection, the stream, error text.
#include "build_id.h"
#include "common.hpp"

namespace sr {

static thread_local std::string g_err;

void set_error(const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int fail(int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

Context &ctx() {
  static Context c;
  return c;
}

int ensure_init() {
  if (ctx().device >= 0) return SR_OK;
  return sr_init(0);
}

}  // namespace sr

extern "C" {

int sr_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    if (e == hipErrorNoDevice) return 0;
    return sr::fail(SR_ERR_HIP, "hipGetDeviceCount failed: %s", hipGetErrorString(e));
  }
  return n;
}

int sr_init(int device) {
  sr::Context &c = sr::ctx();
  if (c.device == device && c.stream) return SR_OK;
  int n = sr_device_count();
  if (n < 0) return n;
  if (n == 0) return sr::fail(SR_ERR_HIP, "no HIP device visible: libsynthray needs an MI355X (gfx950)");
  SR_CHECK(device >= 0 && device < n, "sr_init: device %d out of range (0..%d)", device, n - 1);
  SR_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SR_HIP(hipGetDeviceProperties(&prop, device));
  if (c.stream) {
    (void)hipStreamDestroy(c.stream);
    for (auto &e : c.ev)
      if (e) (void)hipEventDestroy(e);
  }
  SR_HIP(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
  for (auto &e : c.ev) SR_HIP(hipEventCreate(&e));
  c.device = device;
  c.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return SR_OK;
}

int sr_synchronize(void) {
  if (sr::ctx().device < 0) return SR_OK;
  SR_HIP(hipStreamSynchronize(sr::ctx().stream));
  return SR_OK;
}

const char *sr_last_error(void) { return sr::g_err.c_str(); }

// SR_BUILD_ID: sha256 of the library's sources, written into build_id.h by the Makefile.  Profiles committed under
// profiles/ carry the id of the build they were measured on; bench.py prints them only when it matches.
const char *sr_version(void) { return "synthray 0.2 (gfx950) src:" SR_BUILD_ID; }

}  // extern "C"
