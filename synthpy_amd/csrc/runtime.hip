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
  for (int q = 0; q < sr::kStreams; ++q) {
    if (c.streams[q]) (void)hipStreamDestroy(c.streams[q]);
    c.streams[q] = nullptr;
    for (auto &e : c.evs[q]) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
  SR_HIP(hipStreamCreateWithFlags(&c.streams[0], hipStreamNonBlocking));
  for (auto &e : c.evs[0]) SR_HIP(hipEventCreate(&e));
  c.current = 0;
  c.stream = c.streams[0];
  c.ev = c.evs[0];
  c.device = device;
  c.n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  return SR_OK;
}

// A second stream, so that a job of many small ray bundles keeps the GPU full: every call queues on the SELECTED stream;
// work queued on different streams may overlap (one bundle's tail with the next one's start-up).  The caller keeps what
// the streams share in order: sr_synchronize() (both streams) after creating volumes / zeroing images and before reading
// them; a ray bundle is used with one stream at a time.
int sr_stream_select(int index) {
  SR_CHECK(index >= 0 && index < sr::kStreams, "sr_stream_select: stream %d (0..%d)", index, sr::kStreams - 1);
  int rc = sr::ensure_init();
  if (rc) return rc;
  sr::Context &c = sr::ctx();
  if (!c.streams[index]) {
    SR_HIP(hipStreamCreateWithFlags(&c.streams[index], hipStreamNonBlocking));
    for (auto &e : c.evs[index]) SR_HIP(hipEventCreate(&e));
  }
  c.current = index;
  c.stream = c.streams[index];
  c.ev = c.evs[index];
  return SR_OK;
}

int sr_host_alloc(void **out, size_t bytes) {
  SR_CHECK(out != nullptr, "sr_host_alloc: NULL out");
  *out = nullptr;
  int rc = sr::ensure_init();
  if (rc) return rc;
  if (bytes == 0) return SR_OK;
  SR_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return SR_OK;
}

void sr_host_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int sr_synchronize(void) {
  if (sr::ctx().device < 0) return SR_OK;
  for (int q = 0; q < sr::kStreams; ++q)
    if (sr::ctx().streams[q]) SR_HIP(hipStreamSynchronize(sr::ctx().streams[q]));
  return SR_OK;
}

const char *sr_last_error(void) { return sr::g_err.c_str(); }

// SR_BUILD_ID: sha256 of the library's sources, written into build_id.h by the Makefile.  Profiles committed under
// profiles/ carry the id of the build they were measured on; bench.py prints them only when it matches.
const char *sr_version(void) { return "synthray 0.2 (gfx950) src:" SR_BUILD_ID; }

}  // extern "C"
