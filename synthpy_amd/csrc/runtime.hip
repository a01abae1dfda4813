// Runtime of libsynthray.so: device selection, the stream, error text.
#include <dlfcn.h>
#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <thread>

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

// ---- per-call staging memory ------------------------------------------------------------------------------------------
namespace {
struct ScratchBlock {
  void *p = nullptr;
  size_t bytes = 0;
} g_scratch[kStreams];
}  // namespace

void *scratch(size_t bytes) {
  ScratchBlock &b = g_scratch[ctx().current];
  if (b.bytes >= bytes && b.p) return b.p;
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.bytes = 0;
  const size_t want = bytes + bytes / 8;  // a little room: the next call's rays may differ by a few
  hipError_t e = hipMalloc(&b.p, want ? want : 1);
  if (e != hipSuccess) {
    b.p = nullptr;
    e = hipMalloc(&b.p, bytes ? bytes : 1);
    if (e != hipSuccess) {
      b.p = nullptr;
      fail(SR_ERR_HIP, "hipMalloc(%zu) for staging failed: %s", bytes, hipGetErrorString(e));
      return nullptr;
    }
    b.bytes = bytes;
    return b.p;
  }
  b.bytes = want;
  return b.p;
}

void scratch_trim() {
  ScratchBlock &b = g_scratch[ctx().current];
  if (b.p && b.bytes > kScratchKeep) {
    (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
}

void scratch_release() {
  for (auto &b : g_scratch) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
  }
}

// ---- host -> device from pageable memory ----------------------------------------------------------------------------------
namespace {
constexpr int kUpThreads = 6;
constexpr size_t kUpSlice = (size_t)8 << 20;  // bytes per DMA
struct Uploader {
  int device = -1;
  char *bounce[kUpThreads][2] = {};
  hipStream_t st[kUpThreads] = {};
  hipEvent_t done[kUpThreads][2] = {};
  bool ok = false;
} g_up;

bool uploader_ready() {
  Context &c = ctx();
  if (g_up.device == c.device) return g_up.ok;
  g_up.device = c.device;
  g_up.ok = true;
  for (int t = 0; t < kUpThreads && g_up.ok; ++t) {
    g_up.ok = hipStreamCreateWithFlags(&g_up.st[t], hipStreamNonBlocking) == hipSuccess;
    for (int q = 0; q < 2 && g_up.ok; ++q)
      g_up.ok = hipHostMalloc(reinterpret_cast<void **>(&g_up.bounce[t][q]), kUpSlice, hipHostMallocDefault) == hipSuccess &&
                hipEventCreateWithFlags(&g_up.done[t][q], hipEventDisableTiming) == hipSuccess;
  }
  if (!g_up.ok) (void)hipGetLastError();  // no page-locked memory to be had: the runtime's own staging does
  return g_up.ok;
}
}  // namespace

int upload_sync(void *dst, const void *src, size_t bytes, hipStream_t st) {
  if (bytes == 0) return SR_OK;
  hipPointerAttribute_t attr;
  const bool pinned = hipPointerGetAttributes(&attr, src) == hipSuccess && attr.type != hipMemoryTypeUnregistered;
  (void)hipGetLastError();
  // Opt-in (SYNTHRAY_UPLOAD_THREADS=1): measured on two boxes of the pool (round 4, tools/solve_breakdown.py, 0.72 GB of s0) the
  // runtime's own staging moved pageable memory at 29-33 GB/s on one and 49.7 GB/s on the other; the threads here 38-47 GB/s on the
  // second, after a first call that page-locks the bounce buffers (90 ms).  Not a clear win: the runtime's path is the default.
  const char *on = getenv("SYNTHRAY_UPLOAD_THREADS");
  if (pinned || bytes < 4 * kUpSlice || !(on && on[0] == '1') || !uploader_ready()) {
    SR_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st));
    SR_HIP(hipStreamSynchronize(st));
    return SR_OK;
  }
  SR_HIP(hipStreamSynchronize(st));  // whatever `st` holds comes first: the slices go on streams of their own
  const size_t n_slices = (bytes + kUpSlice - 1) / kUpSlice;
  const int device = ctx().device;
  hipError_t errs[kUpThreads];
  std::thread workers[kUpThreads];
  auto work = [&](int t) {
    hipError_t e = hipSetDevice(device);
    int q = 0;
    for (size_t k = (size_t)t; k < n_slices && e == hipSuccess; k += kUpThreads, q ^= 1) {
      const size_t at = k * kUpSlice, len = std::min(kUpSlice, bytes - at);
      if (k >= (size_t)2 * kUpThreads) e = hipEventSynchronize(g_up.done[t][q]);  // the DMA that last read this buffer
      if (e != hipSuccess) break;
      memcpy(g_up.bounce[t][q], static_cast<const char *>(src) + at, len);
      e = hipMemcpyAsync(static_cast<char *>(dst) + at, g_up.bounce[t][q], len, hipMemcpyHostToDevice, g_up.st[t]);
      if (e == hipSuccess) e = hipEventRecord(g_up.done[t][q], g_up.st[t]);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(g_up.st[t]);
    errs[t] = e;
  };
  for (int t = 1; t < kUpThreads; ++t) workers[t] = std::thread(work, t);
  work(0);
  hipError_t bad = errs[0];
  for (int t = 1; t < kUpThreads; ++t) {
    workers[t].join();
    if (errs[t] != hipSuccess) bad = errs[t];
  }
  if (bad != hipSuccess) return fail(SR_ERR_HIP, "host -> device copy of %zu bytes failed: %s", bytes, hipGetErrorString(bad));
  return SR_OK;
}

}  // namespace sr

extern "C" {

// ---- opening the device ---------------------------------------------------------------------------------------------
// Two ranks of a job started in the same instant have left one of them with "no HIP device" (round 2's record:
// gpurun_out/r2_pytest.log, rank 0 of test_slab_pipeline_two_processes_one_gpu).  The HIP runtime initialises ONCE per
// process (std::call_once around hsa_init + device discovery): when that first attempt finds no agent, every later HIP
// call of the process reports hipErrorNoDevice, so retrying hipGetDeviceCount cannot help, and the real reason (the
// ROCr / KFD status) is lost.  The library therefore opens the device in two steps, BEFORE its first HIP call:
//   1. /dev/kfd must open read-write (what ROCr does first); errno is kept for the message;
//   2. hsa_init() through the ROCr library HIP itself uses (dlopen of the already-mapped libhsa-runtime64.so.1): it may
//      be called again after a failure, and its reference stays held, so HIP's own hsa_init afterwards only counts up.
// Either step is retried a bounded number of times with back-off (50 ms doubling, 6 tries, 3.2 s in all) -- only in a
// process that has not initialised HIP yet.  What failed, with errno / the HSA status text, goes into sr_last_error().
namespace {

bool g_preflight_done = false;
std::string g_preflight_note;  // what the pre-flight saw: part of every "no device" message

void sleep_ms(int ms) {
  struct timespec ts = {ms / 1000, (long)(ms % 1000) * 1000000L};
  nanosleep(&ts, nullptr);
}

// SR_OK: go on to HIP; 1: this machine has no GPU driver node for this process; < 0: the device could not be opened (error text set)
int preflight() {
  if (g_preflight_done) return SR_OK;
  typedef int (*hsa_init_fn)(void);
  typedef int (*hsa_status_string_fn)(int, const char **);
  void *hsa = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
  hsa_init_fn p_init = hsa ? (hsa_init_fn)dlsym(hsa, "hsa_init") : nullptr;
  hsa_status_string_fn p_str = hsa ? (hsa_status_string_fn)dlsym(hsa, "hsa_status_string") : nullptr;
  char note[512];
  int wait = 50;
  for (int attempt = 1; attempt <= 6; ++attempt, wait *= 2) {
    const int fd = open("/dev/kfd", O_RDWR | O_CLOEXEC);
    if (fd < 0) {
      const int err = errno;
      snprintf(note, sizeof note, "open(/dev/kfd): %s (errno %d), attempt %d", strerror(err), err, attempt);
      g_preflight_note = note;
      if (err == ENOENT || err == EACCES || err == EPERM) return 1;  // no driver / no permission: a machine without a GPU for this process; waiting does not change that
      sleep_ms(wait);
      continue;
    }
    close(fd);
    if (!p_init) {  // no ROCr entry point to try by itself: HIP's own initialisation decides
      g_preflight_note = "/dev/kfd opens; libhsa-runtime64.so.1 not loadable for a pre-flight hsa_init";
      g_preflight_done = true;
      return SR_OK;
    }
    const int st = p_init();
    if (st == 0) {
      snprintf(note, sizeof note, "/dev/kfd opens, hsa_init ok at attempt %d", attempt);
      g_preflight_note = note;
      g_preflight_done = true;
      return SR_OK;
    }
    const char *txt = nullptr;
    if (p_str) (void)p_str(st, &txt);
    snprintf(note, sizeof note, "hsa_init: %s (status 0x%x), attempt %d", txt ? txt : "?", st, attempt);
    g_preflight_note = note;
    sleep_ms(wait);
  }
  return sr::fail(SR_ERR_HIP, "no HIP device visible: %s", g_preflight_note.c_str());
}

}  // namespace

int sr_device_count(void) {
  int rc = preflight();
  if (rc > 0) return 0;  // no amdgpu driver node: zero devices, said plainly (sr_init gives the reason)
  if (rc) return rc;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess)  // hipErrorNoDevice included: the caller gets the runtime's own words, not "0 devices"
    return sr::fail(SR_ERR_HIP, "hipGetDeviceCount: %s (%s); %s; HIP_VISIBLE_DEVICES=%s ROCR_VISIBLE_DEVICES=%s",
                    hipGetErrorName(e), hipGetErrorString(e), g_preflight_note.c_str(),
                    getenv("HIP_VISIBLE_DEVICES") ? getenv("HIP_VISIBLE_DEVICES") : "(unset)",
                    getenv("ROCR_VISIBLE_DEVICES") ? getenv("ROCR_VISIBLE_DEVICES") : "(unset)");
  return n;
}

// library stream `index`, its timing events, and its side stream (common.hpp).  SYNTHRAY_SIDE_PRIORITY=low|high: the side
// stream below / above the library streams (default: the same priority)
static int make_stream(sr::Context &c, int index) {
  SR_HIP(hipStreamCreateWithFlags(&c.streams[index], hipStreamNonBlocking));
  for (auto &e : c.evs[index]) SR_HIP(hipEventCreate(&e));
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const char *pe = getenv("SYNTHRAY_SIDE_PRIORITY");
  int prio = 0;
  if (pe && pe[0] == 'l') prio = least;
  if (pe && pe[0] == 'h') prio = greatest;
  SR_HIP(hipStreamCreateWithPriority(&c.side[index], hipStreamNonBlocking, prio));
  for (auto &e : c.side_ev[index]) SR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return SR_OK;
}

int sr_init(int device) {
  sr::Context &c = sr::ctx();
  if (c.device == device && c.stream) return SR_OK;
  int n = sr_device_count();
  if (n < 0) return n;
  if (n == 0) return sr::fail(SR_ERR_HIP, "no HIP device visible: libsynthray needs an MI355X (gfx950); %s", g_preflight_note.c_str());
  SR_CHECK(device >= 0 && device < n, "sr_init: device %d out of range (0..%d)", device, n - 1);
  SR_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SR_HIP(hipGetDeviceProperties(&prop, device));
  if (c.stream) (void)sr_release_caches();  // another device: what sr_trace kept on the first one goes
  for (int q = 0; q < sr::kStreams; ++q) {
    if (c.streams[q]) (void)hipStreamDestroy(c.streams[q]);
    c.streams[q] = nullptr;
    if (c.side[q]) (void)hipStreamDestroy(c.side[q]);
    c.side[q] = nullptr;
    for (auto &e : c.evs[q]) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
    for (auto &e : c.side_ev[q]) {
      if (e) (void)hipEventDestroy(e);
      e = nullptr;
    }
  }
  {
    int rc0 = make_stream(c, 0);
    if (rc0) return rc0;
  }
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
    rc = make_stream(c, index);
    if (rc) return rc;
  }
  c.current = index;
  c.stream = c.streams[index];
  c.ev = c.evs[index];
  return SR_OK;
}

// Work queued on stream `waiting` after this call starts only when everything queued on stream `on` BEFORE this call is done
// (an event recorded on `on`, waited for on `waiting`; the host does not wait).  What the slab pipeline orders its two streams
// with: traces on stream 0, the hand-off records' ncclSend / ncclRecv on stream 1 (distributed.SlabPipeline).
int sr_stream_wait(int waiting, int on) {
  SR_CHECK(waiting >= 0 && waiting < sr::kStreams && on >= 0 && on < sr::kStreams && waiting != on, "sr_stream_wait: streams %d, %d", waiting, on);
  sr::Context &c = sr::ctx();
  const int saved = c.current;
  int rc = sr_stream_select(on);  // creates the streams on first use
  if (!rc) rc = sr_stream_select(waiting);
  if (!rc) rc = sr_stream_select(saved);
  if (rc) return rc;
  static hipEvent_t order[sr::kStreams] = {};
  static int order_device = -1;
  if (order_device != c.device) {  // (re)created with the device's streams
    for (auto &e : order) {
      if (e) (void)hipEventDestroy(e);
      SR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    order_device = c.device;
  }
  SR_HIP(hipEventRecord(order[on], c.streams[on]));
  SR_HIP(hipStreamWaitEvent(c.streams[waiting], order[on], 0));
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

int sr_device_memory(int64_t *free_bytes, int64_t *total_bytes) {
  int rc = sr::ensure_init();
  if (rc) return rc;
  size_t f = 0, t = 0;
  SR_HIP(hipMemGetInfo(&f, &t));
  if (free_bytes) *free_bytes = (int64_t)f;
  if (total_bytes) *total_bytes = (int64_t)t;
  return SR_OK;
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
