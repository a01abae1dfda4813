// Shared host-side plumbing of libsynthray.so: error text, the per-process HIP context
// (device + one stream + timing events) and the opaque handle layouts.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "synthray.h"

namespace sr {

constexpr double kC = 299792458.0;  // scipy.constants.c (full_solver.py:93)

// ---- errors ----------------------------------------------------------------------
void set_error(const char *fmt, ...);
int fail(int code, const char *fmt, ...);

#define SR_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return sr::fail(SR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),     \
                      __FILE__, __LINE__);                                                   \
  } while (0)

#define SR_CHECK(cond, ...)                                 \
  do {                                                      \
    if (!(cond)) return sr::fail(SR_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// ---- context ---------------------------------------------------------------------
constexpr int kStreams = 2;
constexpr int kMaxTileSegs = 8;  // segments of a tile-path trace that are timed one by one (trace.hip: trace_tiled)
struct Context {
  int device = -1;
  hipStream_t stream = nullptr;  // the stream every call of the library queues its work on: streams[current]
  hipEvent_t *ev = nullptr;      // the timing events that go with it: evs[current]
  hipStream_t streams[kStreams] = {nullptr, nullptr};
  // [0..3]: start, kernels' start, kernels' end, end of a trace; [4 + 2q], [5 + 2q]: around the tile kernel of segment q
  hipEvent_t evs[kStreams][4 + 2 * kMaxTileSegs] = {};
  // One SIDE stream per library stream (round 5): the rays a tile-path trace loses are carried to the end of the volume by
  // k_trace_f64 there, beside the tile kernel's remaining segments on the library stream (trace.hip: trace_tiled).  Ordered with
  // side_ev[.][0] (the library stream got this far: the side stream may go on) and [1] (the side stream is done: joined before the
  // trace's last kernel).  Everything queued on a side stream is joined to its library stream before the call that queued it returns.
  hipStream_t side[kStreams] = {nullptr, nullptr};
  hipEvent_t side_ev[kStreams][2] = {};
  int current = 0;
  int n_cu = 256;
};
Context &ctx();
int ensure_init();  // sr_init(0) on first use

template <typename T>
int dev_alloc(T **p, size_t count) {
  *p = nullptr;
  if (count == 0) return SR_OK;
  SR_HIP(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
  return SR_OK;
}
inline void dev_free(void *p) {
  if (p) (void)hipFree(p);
}

inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// Device memory for the staging of ONE call (download in ray order, counts as float64, amplitude, sr_rays_optics): a block per
// stream that grows and is kept, so that the entry points a caller loops over do not hipMalloc / hipFree (each waits for the
// device) every time.  The caller must be done with it (stream synchronised) before it returns.  Blocks beyond kScratchKeep
// are given back by scratch_trim() at the end of the call; sr_release_caches() frees all.  nullptr + error text on failure.
constexpr size_t kScratchKeep = (size_t)2 << 30;
void *scratch(size_t bytes);
void scratch_trim();
void scratch_release();
// Host -> device copy of `bytes` on `st`, then the stream is waited for.  From pageable memory the runtime's own staging runs on
// one thread (8-30 GB/s on this box); here several threads copy slices into page-locked bounce buffers of the library and
// each slice goes to the device by DMA as soon as it is in: the link's rate.  Page-locked sources are copied directly.
int upload_sync(void *dst, const void *src, size_t bytes, hipStream_t st);

// Per-launch totals (ray steps, deposited rays) are summed by one atomic per wavefront.  156 000 wavefronts adding to ONE
// address take ~2 ms of serialised device-scope atomics (it was the whole duration of the deposit kernel), so the totals
// are striped over kStripes cache lines picked by the workgroup index and added up on the host.
constexpr int kStripes = 256, kStripeStride = 16;  // 16 x 8 B = one 128-byte line per stripe
// [0..15] plain counters ([1], [2]: queue lengths of a trace, [3]: rays past the first kernel so far, [5]: edge guard, [8]: the
// rays the tile path has lost so far in this trace = entries of the straggler records), then 3 striped totals: 0 ray steps, 1 deposited rays, 2 the steps of an edge-guard re-trace (not
// part of the job's ray-step count: those rays' steps were counted when the mixed kernel took them)
constexpr size_t kCounterWords = 16 + 3 * (size_t)kStripes * kStripeStride;
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long *stripe(unsigned long long *base, int which) {
  return base + 16 + ((size_t)which * kStripes + (blockIdx.x & (kStripes - 1))) * kStripeStride;
}
#endif
// Append the launch slots of the lanes with `want` to a queue: the wavefront ballots, ONE lane reserves popcount(mask)
// slots with a single atomic, and each lane takes its prefix rank (compaction by ballot + prefix popcount).  Must be
// reached by the whole wavefront.
#ifdef __HIPCC__
__device__ __forceinline__ void queue_push(unsigned long long *count, uint32_t *list, bool want, uint32_t slot) {
  const unsigned long long mask = __ballot(want);
  if (mask == 0ull) return;
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)mask) - 1;
  unsigned long long base = 0;
  if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(mask));
  base = __shfl(base, leader, 64);
  if (want) list[base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
}
#endif
inline unsigned long long stripe_sum(const unsigned long long *host_words, int which) {
  unsigned long long t = 0;
  for (int s = 0; s < kStripes; ++s) t += host_words[16 + ((size_t)which * kStripes + s) * kStripeStride];
  return t;
}

// ---- the packed node order of a volume ---------------------------------------------------------------------------
// Node (ia, ib, ic) -- ia along the probing axis -- sits at
//     ((ia / 8) * nb*nc + ib*nc + ic) * 8 + ia % 8 :
// OCTETS of eight consecutive node planes; inside an octet the lateral columns are contiguous and each column's eight
// planes are one 128-byte line of float4 records.  A ray marching along `a` still walks four line-sized streams (one new
// line per column every eight steps), and the lines all the rays of a launch need at about the same time -- they advance
// through the planes together -- form ONE contiguous region of nb*nc*128 bytes instead of one line in every 16*na bytes:
// that is what HBM sees when the lines are not shared (sparse bundles).  The last octet is padded (zeros).
#if defined(__HIPCC__)
#define SR_HD __host__ __device__
#else
#define SR_HD
#endif
SR_HD inline int64_t node_index(int ia, int ib, int ic, int nb, int nc) {
  return ((int64_t)(ia >> 3) * nb * nc + (int64_t)ib * nc + ic) * 8 + (ia & 7);
}
inline size_t packed_nodes(int na, int nb, int nc) { return (size_t)((na + 7) / 8) * (size_t)nb * (size_t)nc * 8; }

}  // namespace sr

// ---- opaque handles --------------------------------------------------------------
// Volume: one float4 per node {dnd_b, dnd_c, dnd_a, hi(n-1)} in the packed node order above (sr::node_index;
// b = (a+1)%3, c = (a+2)%3), plus lo(n-1) so that n-1 = hi + lo keeps 48 bits, and the node coordinates in float64
// (what scipy's interpolator works with).
struct sr_volume {
  int nx = 0, ny = 0, nz = 0;
  int axis = 2;          // physical index of `a`
  int na = 0, nb = 0, nc = 0;
  int flags = 0;
  double omega = 0;
  float4 *P = nullptr;   // packed node order (sr::node_index), sr::packed_nodes records
  float *L = nullptr;    // the same order, or nullptr
  double *K = nullptr;   // kappa per node (packed order) or nullptr (inverse bremsstrahlung)
  double *Q = nullptr;   // {ne, Bx, By, Bz} per node (packed order), or nullptr (Faraday rotation)
  // The tile path's coefficient records ready-made (trace_tile.inc, REC): per node plane k and lateral CELL (ib, ic) the 16 float64
  // {a[4], b[4], c[4], d[4]} coefs_from_corners forms, at R[((k * (nc-1) + ic) * (nb-1) + ib) * 16] -- 128 B per (cell, plane), built
  // at the first trace that wants them (SYNTHRAY_TILE_RECORDS) and kept with the volume; nullptr: not built
  mutable double *R = nullptr;
  mutable bool R_tried = false;  // the allocation was refused once: do not ask again
  double verdet = 0;
  // a slab of node planes k_lo..k_hi of a domain with n_glob planes on the probing axis (A12); whole volume: 0..n-1
  bool is_slab = false;
  int k_lo = 0, k_hi = 0, n_glob = 0;
  double *g[3] = {nullptr, nullptr, nullptr};   // device node coordinates, order (a, b, c)
  double *rg[3] = {nullptr, nullptr, nullptr};  // device 1/(g[i+1]-g[i]), order (a, b, c)
  std::vector<double> hg[3];                    // host copies, order (a, b, c)
  // per-plane RK4 step constants (trace.hip: StepTab), one device table per `substeps` value, built at the first trace
  // that asks for it and kept until the volume goes: a pure function of the node coordinates, omega and substeps
  mutable std::map<int, void *> step_tabs;
};

struct sr_rays {
  int64_t n = 0;
  int64_t cap = 0;       // rays the buffers were made for (sr_rays_create); sr_trace's pipeline runs a shorter last chunk with n < cap,
                         // and every buffer allocated later (sort_tmp, rec, rec2, order2) is sized by cap, never by the current n
  // launch positions' bounding box, (min x, y, z, max x, y, z), found at upload / generate: the rays per lateral cell of the
  // BEAM (not of the whole lateral grid) decide between the tile path and the per-ray kernel (trace.hip: tile_plan)
  double bbox[6] = {0, 0, 0, 0, 0, 0};
  bool have_bbox = false;
  bool bbox_given = false;  // sr_rays_set_bbox: the caller's word for the beam; stays across hand-offs (comm.hip, sr_rays_handoff_upload)
  double *s0 = nullptr;  // (9, N) original order
  // outputs are kept in LAUNCH order (coalesced); perm[j] = original index of launch slot j
  double *sf = nullptr;  // (9, N)
  double *rf = nullptr;  // (4, N)
  double *Jf = nullptr;  // (2, N, 2)
  uint32_t *perm = nullptr;
  uint32_t *keys = nullptr;
  uint32_t *bins = nullptr;  // counting-sort workspace: [coarse digit][workgroup] counts + scan totals
  uint32_t *sort_tmp = nullptr;  // (key, ray) pairs grouped by coarse digit, 2 x N
  int64_t bins_cap = 0;
  uint32_t *fb_list = nullptr;      // rays for the time-stepping fallback
  unsigned long long *counters = nullptr;  // plain words [1], [2]: queue lengths of a trace, [3]: first-level queue total; stripes: ray steps, deposited
  double *rec = nullptr;                   // (10, N) hand-off records (A12), allocated at first use
  double *rec2 = nullptr;                  // the records' second buffer and the new order, for the tile path's re-binning (trace_tile.inc)
  uint32_t *order2 = nullptr;
  // the tile path's stragglers (trace_tile.inc): (10, cap) records of the rays a tile lost, in the order they were lost -- the state
  // they entered their segment with, then (k_trace_f64 on the side stream) their state on the volume's / slab's last node plane --
  // and the entry counts after each segment (what one straggler launch takes)
  double *strag_rec = nullptr;
  unsigned long long *strag_snap = nullptr;
  int strag_snap_cap = 0;
  bool have_s0 = false, traced = false, sorted = false, have_rec = false;
  bool counters_carry = false;  // the step / fallback totals of earlier traces have not been read yet: keep adding
  int tile_segs = 0;  // the last trace ran the tile path in this many timed segments (0: not the tile path, or more than kMaxTileSegs)
  int tile_segs_run = 0;  // ... in this many segments, timed or not (sr_rays_tile_segments)
  bool tile_rec = false;  // ... with the records kernel (sr_rays_tile_records)
  // Edge guard (deposit.hip): per launch slot, a bound on how far the exit ANGLE of a ray traced by the mixed build may
  // be from the float64 build's [rad]; 0 for rays the float64 kernels wrote, +inf when the kernel keeps no bound.  With
  // it go what a re-trace needs: the volume and the parameters of the last trace (the volume must outlive the deposits).
  float *guard = nullptr;
  const sr_volume *last_vol = nullptr;
  sr_trace_params last_p{};
  void *guard_set = nullptr;             // sr_rays_refine: the diagnostics' chains and detector edges (device), and the host copy
  std::vector<char> guard_set_host;      // the upload reads
  bool guard_live = false;   // the last trace ran the mixed build on a whole volume: guard[] is meaningful
  double guard_len = 0;      // extent of the volume along the probing axis [m]: position bound = guard_len * angle bound
};

namespace sr {
int retrace_f64(const sr_rays *r, const uint32_t *list, const unsigned long long *count);  // trace.hip (edge guard)
}

struct sr_image {
  int kind = 0;
  int nx = 0, ny = 0;  // bins (counts) or edges (complex)
  double x_lo = 0, x_hi = 0, y_lo = 0, y_hi = 0;
  void *d = nullptr;
  int64_t bytes = 0;
};
