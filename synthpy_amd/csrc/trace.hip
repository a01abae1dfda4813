// A2 + A3 + A4 + A6 on the device: ScalarDomain.solve (full_solver.py:376-403), dsdt (:516-544),
// the RegularGridInterpolator gathers (:317-347) and ray_to_Jonesvector (:838-894).
//
// One ray per work-item.  The integrator is the plane-to-plane RK4 stated in
// oracle/synthray_oracle.c (trace_one_planes): the probing coordinate p_a is the independent
// variable, every step runs from node plane k to k+1 of that axis, so
//   * all rays of the launch are in the same cell layer k at the same loop iteration: the
//     interpolation weight along `a` is wave-uniform (0, 1/2, 1 at one step per cell);
//   * a ray needs ONE new node plane (4 float4 corners) per step; the upper plane of step k is
//     the lower plane of step k+1 and stays in registers;
//   * with the probing axis fastest in memory each ray walks four contiguous streams, and rays
//     binned by entry cell (counting sort below) make a wavefront read the same few lines.
// Rays the plane form cannot take (v_a <= 0, not in front of the entry plane, t_exit > t_end)
// are queued and re-traced by the time-stepping form with located faces (trace_one_t).
//
// Compiled with -ffp-contract=off; fused multiply-adds are written out with fma().
#include <sys/mman.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>

#include "common.hpp"

namespace {

struct VolDev {
  const float4 *P;
  const float *L;
  const double *g[3];   // (a, b, c)
  const double *rg[3];  // (a, b, c)
  int na, nb, nc;
  double omega;
  const double *K;  // kappa or nullptr
  const double *Q;  // {ne, Bx, By, Bz} or nullptr
  double verdet;
  int64_t OS;           // nodes per octet of node planes: nb*nc*8 (the packed node order, common.hpp)
  const double *R;      // ready-made coefficient records per (node plane, lateral cell), or nullptr (common.hpp: sr_volume::R)
};

// The packed node order (sr::node_index): column (ib, ic) starts at col8() inside every octet, node plane k of a column
// sits koff(k) further on; the lateral neighbours of a node are kSC (next ic) and sb = nc*8 (next ib) records away.
constexpr int64_t kSC = 8;
__device__ __forceinline__ int64_t col8(const VolDev &V, int ib, int ic) { return ((int64_t)ib * V.nc + ic) * 8; }
__device__ __forceinline__ int64_t koff(const VolDev &V, int k) { return (int64_t)(k >> 3) * V.OS + (k & 7); }

// wave-uniform constants of one RK4 sub-step from node plane k (sub-interval m): index k*sub + m.
// Built on the host with the oracle's formulas (trace_one_planes): dz = (g[k+1]-g[k])/sub, za = g[k] + m*dz,
// zb = g[k+1] for the last sub-interval else g[k] + (m+1)*dz, h = zb - za, plane weights (z - g[k])/(g[k+1]-g[k]).
struct StepTab {
  double h, hh, h6, h6w;  // h, h/2, h/6, omega*h/6
  float hf, hhf, wa0, waH, wa1, pad[3];
};
static_assert(sizeof(StepTab) == 64, "StepTab is read with scalar loads, keep it 64 bytes");
// the same for the all-float64 build (k_trace_planes): plane weights in float64; shares the device buffer
struct StepTab64 {
  double h, hh, h6, wa0, waH, wa1, h6w, pad[1];  // h6w = omega*h/6 (k_trace_f64)
};
static_assert(sizeof(StepTab64) == sizeof(StepTab), "the two step tables share one buffer");

struct TraceArgs {
  VolDev V;
  const StepTab *tab;      // k_trace_mx
  const StepTab64 *tab64;  // k_trace_planes (same buffer, filled for the kernel that is launched)
  const double *s0;
  int64_t N;
  const uint32_t *perm;
  double *sf, *rf, *Jf;
  double t_end, extent, dt;
  int axis, row_order, sub;
  unsigned long long *counters;  // plain words [1], [2]: queue lengths; stripe 0: ray steps
  // Work lists (launch slots).  in_list == nullptr: the kernel takes every slot 0..N-1; else the *in_count slots
  // queued by the level before it.  Rays a kernel cannot finish are appended to out_list (nullptr: none follows).
  const uint32_t *in_list;
  const unsigned long long *in_count;
  uint32_t *out_list;
  unsigned long long *out_count;
  double *rec;   // (10, N) hand-off records in launch order (A12), or nullptr
  int handoff;   // SR_HANDOFF_ENTER | SR_HANDOFF_EXIT
  unsigned n_blocks;  // real blocks (grid is padded to a multiple of 8 for the XCD remap)
  float *guard;       // per launch slot: bound on the exit-angle error of the mixed build (common.hpp), or nullptr
  int step_stripe;    // which striped total the step count goes to: 0, or 2 for an edge-guard re-trace
  float guard_h6, guard_ih6;  // the largest h/6 over the node planes and its reciprocal (k_trace_mx's bound)
  // k_trace_f64 over a range of node planes of a whole volume (the tile path's recovery of the rays a tile lost): steps from
  // node plane k_first to k_last (-1: the last plane), state from / to the hand-off records as `handoff` says; `recover`: a ray
  // the plane form cannot finish goes to out_list (from s0, the usual levels) instead of coming out NaN as on a slab
  int k_first, k_last, recover;
  // The tile path's stragglers (trace_tile.inc): the queue is the entries [*in_first, *in_count) of the straggler records THEMSELVES
  // (in_iota: slot = entry, no list; rec = those records, N = their row pitch), stepped from k_first to the last node plane with
  // SR_HANDOFF_ENTER | SR_HANDOFF_EXIT: the state on the last plane goes back into the entry.  finish_later: the outputs are
  // formed from it afterwards (k_strag_finish) -- the ray's steps are counted here only if it will be finished there (t <= t_end)
  const unsigned long long *in_first;
  int in_iota, finish_later;
};

using sr::queue_push;  // common.hpp

// index of the cell [g[i], g[i+1]) holding p, p in [g[0], g[n-1]]; scipy's rule for the last node
__device__ __forceinline__ int find_cell(const double *g, int n, double p, double g0, double inv_d) {
  int i = (int)((p - g0) * inv_d);
  i = i < 0 ? 0 : (i > n - 2 ? n - 2 : i);
  while (i > 0 && p < g[i]) --i;
  while (i < n - 2 && p >= g[i + 1]) ++i;
  return i;
}

// ---------------------------------------------------------------------------------------
// ray binning: counting sort by entry cell (ib, ic)
// ---------------------------------------------------------------------------------------
__global__ void k_iota(uint32_t *perm, int64_t n) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) perm[i] = (uint32_t)i;
}

// launch positions' bounding box (NaN positions ignored): doubles ordered as unsigned integers, one atomic per wavefront and bound
__device__ __forceinline__ unsigned long long ordered_bits(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
inline double from_ordered_bits(unsigned long long u) {
  u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
  double v;
  memcpy(&v, &u, sizeof v);
  return v;
}
__global__ __launch_bounds__(256) void k_bbox(const double *__restrict__ s0, int64_t N, unsigned long long *__restrict__ out) {
  unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
    for (int q = 0; q < 3; ++q) {
      const double v = s0[q * N + i];
      if (v == v) {
        const unsigned long long u = ordered_bits(v);
        lo[q] = u < lo[q] ? u : lo[q];
        hi[q] = u > hi[q] ? u : hi[q];
      }
    }
  for (int q = 0; q < 3; ++q) {
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long a = __shfl_down(lo[q], off, 64), b = __shfl_down(hi[q], off, 64);
      lo[q] = a < lo[q] ? a : lo[q];
      hi[q] = b > hi[q] ? b : hi[q];
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&out[q], lo[q]);
      atomicMax(&out[3 + q], hi[q]);
    }
  }
}

// Morton (Z-order) index of a lateral cell: consecutive keys form compact square patches at every scale, so the
// 256 rays of a workgroup enter through a few neighbouring cells whatever the ray density
__device__ __forceinline__ uint32_t spread_bits(uint32_t v) {
  v &= 0xffffu;
  v = (v | (v << 8)) & 0x00ff00ffu;
  v = (v | (v << 4)) & 0x0f0f0f0fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

__global__ void k_keys(VolDev V, const double *__restrict__ s0, int64_t N, int axis, uint32_t oob_key,
                       uint32_t *__restrict__ keys) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int b = (axis + 1) % 3, c = (axis + 2) % 3;
  double pa = s0[axis * N + i], pb = s0[b * N + i], pc = s0[c * N + i];
  const double va = s0[(3 + axis) * N + i], vb = s0[(3 + b) * N + i], vc = s0[(3 + c) * N + i];
  const double ga0 = V.g[0][0];
  if (va > 0 && pa < ga0) {  // where the ray meets the entry plane
    const double tau = (ga0 - pa) / va;
    pb = pb + vb * tau;
    pc = pc + vc * tau;
  }
  const double gb0 = V.g[1][0], gbL = V.g[1][V.nb - 1], gc0 = V.g[2][0], gcL = V.g[2][V.nc - 1];
  uint32_t key = oob_key;  // out-of-volume / NaN rays go last
  if (pb >= gb0 && pb <= gbL && pc >= gc0 && pc <= gcL) {
    const int ib = find_cell(V.g[1], V.nb, pb, gb0, (V.nb - 1) / (gbL - gb0));
    const int ic = find_cell(V.g[2], V.nc, pc, gc0, (V.nc - 1) / (gcL - gc0));
    key = (spread_bits((uint32_t)ib) << 1) | spread_bits((uint32_t)ic);
  }
  keys[i] = key;
}

// Two-level counting sort of the rays by Morton key, all counting in LDS (no global atomic per ray):
//   coarse digit = key >> lo_bits (the coarser Morton cell), fine digit = key & (2^lo_bits - 1)
//   1. k_bin_count: every workgroup counts the coarse digits of its tile of kBinTile rays -> counts[digit][workgroup]
//   2. exclusive scan over counts (k_scan_blocks / k_scan / k_scan_add): where each (digit, workgroup) run starts
//   3. k_bin_coarse: the tile again, each ray to its run (rank inside the run by an LDS atomic) -> (key, ray) pairs
//      grouped by coarse digit
//   4. k_bin_fine: one workgroup per coarse digit orders its group by the fine digit (LDS counts + scan) -> perm
// The order of the rays inside one cell is whatever the LDS atomics gave: results do not depend on it.
constexpr int kBinTile = 4096;
constexpr int kBinMaxDigits = 2048;  // LDS counters per workgroup (2^11: lateral grids up to 2048 x 2048 cells)

__global__ __launch_bounds__(256) void k_bin_count(const uint32_t *__restrict__ keys, int64_t N, int lo_bits, int n_coarse,
                                                   uint32_t *__restrict__ counts, unsigned n_wg) {
  __shared__ uint32_t cnt[kBinMaxDigits + 1];
  for (int t = threadIdx.x; t < n_coarse; t += blockDim.x) cnt[t] = 0u;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kBinTile;
  for (int q = threadIdx.x; q < kBinTile; q += blockDim.x)
    if (base + q < N) atomicAdd(&cnt[keys[base + q] >> lo_bits], 1u);
  __syncthreads();
  for (int t = threadIdx.x; t < n_coarse; t += blockDim.x) counts[(int64_t)t * n_wg + blockIdx.x] = cnt[t];
}

__global__ __launch_bounds__(256) void k_bin_coarse(const uint32_t *__restrict__ keys, int64_t N, int lo_bits, int n_coarse,
                                                    const uint32_t *__restrict__ starts, unsigned n_wg,
                                                    uint32_t *__restrict__ out_keys, uint32_t *__restrict__ out_rays) {
  __shared__ uint32_t cur[kBinMaxDigits + 1];
  for (int t = threadIdx.x; t < n_coarse; t += blockDim.x) cur[t] = starts[(int64_t)t * n_wg + blockIdx.x];
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * kBinTile;
  for (int q = threadIdx.x; q < kBinTile; q += blockDim.x) {
    const int64_t i = base + q;
    if (i < N) {
      const uint32_t key = keys[i];
      const uint32_t pos = atomicAdd(&cur[key >> lo_bits], 1u);
      out_keys[pos] = key;
      out_rays[pos] = (uint32_t)i;
    }
  }
}

// group d = rays [starts[d*n_wg], starts[(d+1)*n_wg]) (the scan is over [digit][workgroup], so a digit's first run is its start)
__global__ __launch_bounds__(1024) void k_bin_fine(const uint32_t *__restrict__ in_keys, const uint32_t *__restrict__ in_rays,
                                                    int64_t N, int lo_bits, int n_coarse, const uint32_t *__restrict__ starts,
                                                    unsigned n_wg, uint32_t *__restrict__ perm) {
  __shared__ uint32_t cnt[kBinMaxDigits];
  __shared__ uint32_t wsum[16];
  const int d = blockIdx.x;
  const uint32_t lo = starts[(int64_t)d * n_wg];
  const uint32_t hi = d + 1 < n_coarse ? starts[(int64_t)(d + 1) * n_wg] : (uint32_t)N;
  if (hi <= lo) return;
  const int n_fine = 1 << lo_bits;
  const uint32_t mask = (uint32_t)n_fine - 1u;
  for (int t = threadIdx.x; t < n_fine; t += blockDim.x) cnt[t] = 0u;
  __syncthreads();
  for (uint32_t q = lo + threadIdx.x; q < hi; q += blockDim.x) atomicAdd(&cnt[in_keys[q] & mask], 1u);
  __syncthreads();
  // exclusive scan of cnt[0..n_fine): two counters per thread at most (n_fine <= 2048), wavefront shuffles + one LDS hop
  const int t0 = 2 * (int)threadIdx.x;
  const uint32_t a = t0 < n_fine ? cnt[t0] : 0u, b = t0 + 1 < n_fine ? cnt[t0 + 1] : 0u;
  uint32_t incl = a + b;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  const uint32_t ex = lo + woff + incl - (a + b);
  __syncthreads();
  if (t0 < n_fine) cnt[t0] = ex;
  if (t0 + 1 < n_fine) cnt[t0 + 1] = ex + a;
  __syncthreads();
  for (uint32_t q = lo + threadIdx.x; q < hi; q += blockDim.x) {
    const uint32_t pos = atomicAdd(&cnt[in_keys[q] & mask], 1u);
    perm[pos] = in_rays[q];
  }
}

// exclusive scan of bins[0..n) in place; one workgroup, chunked (n <= a few million cells)
__global__ void k_scan(uint32_t *bins, int64_t n) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t carry;
  const int t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + t;
    const uint32_t v = i < n ? bins[i] : 0u;
    part[t] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t add = t >= off ? part[t - off] : 0u;
      __syncthreads();
      part[t] += add;
      __syncthreads();
    }
    const uint32_t incl = part[t];
    if (i < n) bins[i] = carry + incl - v;
    __syncthreads();
    if (t == 1023) carry += incl;
    __syncthreads();
  }
}

// Two-level exclusive scan for the cell counts: each 256-thread workgroup scans 2048 counts (8 per lane,
// wavefront shuffles + one LDS hop) and writes its total; k_scan (one workgroup) scans the totals; k_scan_add
// adds them back.
constexpr int kScanPerBlock = 2048;
__global__ __launch_bounds__(256) void k_scan_blocks(uint32_t *__restrict__ bins, int64_t n, uint32_t *__restrict__ sums) {
  __shared__ uint32_t wsum[4];
  const int64_t base = (int64_t)blockIdx.x * kScanPerBlock + (int64_t)threadIdx.x * 8;
  uint32_t v[8], run = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v[q] = base + q < n ? bins[base + q] : 0u;
    run += v[q];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  uint32_t incl = run;  // inclusive scan of the lane totals across the wavefront
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wsum[wid] = incl;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wid; ++w) woff += wsum[w];
  uint32_t ex = woff + incl - run;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if (base + q < n) bins[base + q] = ex;
    ex += v[q];
  }
  if (threadIdx.x == 255) sums[blockIdx.x] = woff + incl;
}
__global__ void k_scan_add(uint32_t *__restrict__ bins, int64_t n, const uint32_t *__restrict__ sums) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) bins[i] += sums[i / kScanPerBlock];
}

// ---------------------------------------------------------------------------------------
// outputs: final state at t_end (A2), exit-plane 4-vector and Jones vector (A6)
// ---------------------------------------------------------------------------------------
// A6 for one ray: exit-plane 4-vector and Jones vector from a state in (a, b, c) order
__device__ __forceinline__ void project(int a, int row_order, double extent, int64_t N, int64_t j, double pa, double pb,
                                        double pc, double va, double vb, double vc, double amp, double phase, double pol,
                                        double *__restrict__ rf, double *__restrict__ Jf) {
  // t_bp = (p_a - extent)/v_a ; positions on the plane, angles to it (full_solver.py:856-881)
  const double tb = (pa - extent) / va;
  double q0 = pb - vb * tb, q2 = pc - vc * tb;
  double a1 = atan(vb / va), a3 = atan(vc / va);
  if (a == 1 && row_order == SR_ROWS_LEGACY) {  // legacy y-probing lists (x, z) = (c, b)
    double t = q0;
    q0 = q2;
    q2 = t;
    t = a1;
    a1 = a3;
    a3 = t;
  }
  rf[j] = q0;
  rf[N + j] = a1;
  rf[2 * N + j] = q2;
  rf[3 * N + j] = a3;
  if (Jf) {  // amp*exp(i*phase)*(-sin pol, cos pol)  (full_solver.py:884-890)
    double sp, cp, so, co;
    sincos(phase, &sp, &cp);
    sincos(pol, &so, &co);
    const double cr = amp * cp, ci = amp * sp;
    const double ex = co * 0.0 - so * 1.0, ey = so * 0.0 + co * 1.0;
    Jf[2 * j] = cr * ex;
    Jf[2 * j + 1] = ci * ex;
    Jf[2 * (N + j)] = cr * ey;
    Jf[2 * (N + j) + 1] = ci * ey;
  }
}

__device__ __forceinline__ void write_outputs(const TraceArgs &A, int64_t j, double pa, double pb, double pc, double va,
                                              double vb, double vc, double phase, double amp, double pol) {
  const int64_t N = A.N;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  A.sf[a * N + j] = pa;
  A.sf[b * N + j] = pb;
  A.sf[c * N + j] = pc;
  A.sf[(3 + a) * N + j] = va;
  A.sf[(3 + b) * N + j] = vb;
  A.sf[(3 + c) * N + j] = vc;
  A.sf[6 * N + j] = amp;
  A.sf[7 * N + j] = phase;
  A.sf[8 * N + j] = pol;
  project(a, A.row_order, A.extent, N, j, pa, pb, pc, va, vb, vc, amp, phase, pol, A.rf, A.Jf);
  if (A.guard) A.guard[j] = 0.f;  // the float64 kernels and NaN rows; the mixed kernels overwrite it with their bound
}
// amp and pol unchanged (no attenuation / Faraday field): rows 6 and 8 of s0
__device__ __forceinline__ void write_outputs(const TraceArgs &A, int64_t j, int64_t i, double pa, double pb,
                                              double pc, double va, double vb, double vc, double phase) {
  write_outputs(A, j, pa, pb, pc, va, vb, vc, phase, A.s0[6 * A.N + i], A.s0[8 * A.N + i]);
}

// ---- A12: slab-to-slab hand-off (oracle: orc_trace_slab) ----------------------------------------------
// A slab volume holds a range of node planes; the record handed over is the plane form's state ON the shared
// plane: rec rows = p_b, p_c, v_a, v_b, v_c, phase, t, amp, pol, ray index, in launch order (read and written
// in place by the same work-item).  v_a = NaN marks a ray the plane form lost; there is no time-stepping
// fallback on a slab (it holds only its own planes): such rays come out NaN.
__device__ __forceinline__ bool handoff_enter(const TraceArgs &A, bool have, int64_t j, double &y0, double &y1, double &y2,
                                              double &y3, double &y4, double &y5, double &y6) {
  if (!have) return false;
  const int64_t N = A.N;
  y0 = A.rec[j];
  y1 = A.rec[N + j];
  y2 = A.rec[2 * N + j];
  y3 = A.rec[3 * N + j];
  y4 = A.rec[4 * N + j];
  y5 = A.rec[5 * N + j];
  y6 = A.rec[6 * N + j];
  return y2 == y2;
}
// amp, pol: rows 7 and 8 of the record when the state came in by hand-off, else rows 6 and 8 of s0
__device__ __forceinline__ void handoff_amp_pol(const TraceArgs &A, int64_t j, int64_t i, double &amp, double &pol) {
  if (A.handoff & SR_HANDOFF_ENTER) {
    amp = A.rec[7 * A.N + j];
    pol = A.rec[8 * A.N + j];
  } else {
    amp = A.s0[6 * A.N + i];
    pol = A.s0[8 * A.N + i];
  }
}
__device__ __forceinline__ void handoff_exit(const TraceArgs &A, bool alive, int64_t j, int64_t i, double y0, double y1,
                                             double y2, double y3, double y4, double y5, double y6, double amp, double pol) {
  const int64_t N = A.N;
  const double nan = __builtin_nan("");
  A.rec[j] = alive ? y0 : nan;
  A.rec[N + j] = alive ? y1 : nan;
  A.rec[2 * N + j] = alive ? y2 : nan;
  A.rec[3 * N + j] = alive ? y3 : nan;
  A.rec[4 * N + j] = alive ? y4 : nan;
  A.rec[5 * N + j] = alive ? y5 : nan;
  A.rec[6 * N + j] = alive ? y6 : nan;
  A.rec[7 * N + j] = alive ? amp : nan;
  A.rec[8 * N + j] = alive ? pol : nan;
  if (!(A.handoff & SR_HANDOFF_ENTER)) A.rec[9 * N + j] = (double)i;
}
__device__ __forceinline__ void write_lost(const TraceArgs &A, int64_t j) {
  const double nan = __builtin_nan("");
  write_outputs(A, j, nan, nan, nan, nan, nan, nan, nan, nan, nan);
}
__global__ void k_perm_from_rec(const double *__restrict__ rec, int64_t N, uint32_t *__restrict__ perm) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j < N) perm[j] = (uint32_t)rec[9 * N + j];
}

// The optional terms' fields at one node column: bilinear in (b, c) on node plane q, X = {kappa, ne, Bx, By, Bz}
__device__ __forceinline__ void aux_plane(const VolDev &V, int64_t q, double w00, double w01, double w10, double w11,
                                          double (&X)[5]) {
  const int64_t sc = kSC, sb = (int64_t)V.nc * 8;
  const int64_t o1 = q + sc, o2 = q + sb, o3 = q + sb + sc;
  X[0] = X[1] = X[2] = X[3] = X[4] = 0.0;
  if (V.K) X[0] = fma(V.K[o3], w11, fma(V.K[o2], w10, fma(V.K[o1], w01, V.K[q] * w00)));
  if (V.Q) {  // one 32-byte node record {ne, Bx, By, Bz} per corner: two 16-byte loads each
    const double4 *Q4 = reinterpret_cast<const double4 *>(V.Q);
    const double4 c0 = Q4[q], c1 = Q4[o1], c2 = Q4[o2], c3 = Q4[o3];
    X[1] = fma(c3.x, w11, fma(c2.x, w10, fma(c1.x, w01, c0.x * w00)));
    X[2] = fma(c3.y, w11, fma(c2.y, w10, fma(c1.y, w01, c0.y * w00)));
    X[3] = fma(c3.z, w11, fma(c2.z, w10, fma(c1.z, w01, c0.z * w00)));
    X[4] = fma(c3.w, w11, fma(c2.w, w10, fma(c1.w, w01, c0.w * w00)));
  }
}

// d(amp)/dt and d(pol)/dt of dsdt (full_solver.py:540, 542) from X and the stage's (amp, v) given in (a, b, c) order;
// B.v is summed in the reference's x, y, z order
__device__ __forceinline__ void aux_rates(const VolDev &V, int a, const double (&X)[5], double amp, double va, double vb,
                                          double vc, double &damp, double &dpol) {
  const double vx = a == 0 ? va : (a == 1 ? vc : vb), vy = a == 0 ? vb : (a == 1 ? va : vc),
               vz = a == 0 ? vc : (a == 1 ? vb : va);
  damp = X[0] * amp;
  dpol = (V.verdet * X[1]) * ((X[2] * vx + X[3] * vy) + X[4] * vz);
}

__global__ void k_ray_to_jones(const double *__restrict__ sf, int64_t N, double extent, int a, int row_order,
                               double *__restrict__ rf, double *__restrict__ Jf) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int b = (a + 1) % 3, c = (a + 2) % 3;
  project(a, row_order, extent, N, j, sf[a * N + j], sf[b * N + j], sf[c * N + j], sf[(3 + a) * N + j], sf[(3 + b) * N + j],
          sf[(3 + c) * N + j], sf[6 * N + j], sf[7 * N + j], sf[8 * N + j], rf, Jf);
}

// ---------------------------------------------------------------------------------------
// plane-stepping tracer
// ---------------------------------------------------------------------------------------
template <typename W>
struct Corner4 {
  W x, y, z, w;
};

// 4 corners (b-bit, c-bit) of one node plane, in the blend's working precision W
template <typename W, bool PHASE>
__device__ __forceinline__ void load_plane(const VolDev &V, int64_t q, Corner4<W> (&c)[4]) {
  const int64_t sc = kSC, sb = (int64_t)V.nc * 8;
  const int64_t off[4] = {0, sc, sb, sb + sc};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4 v = V.P[q + off[k]];
    c[k].x = (W)v.z;
    c[k].y = (W)v.x;
    c[k].z = (W)v.y;
    if (PHASE) {
      if (sizeof(W) == 8)
        c[k].w = (W)((double)v.w + (double)V.L[q + off[k]]);
      else
        c[k].w = (W)v.w;
    } else {
      c[k].w = (W)0;
    }
  }
}

template <typename W, bool PHASE>
__device__ __forceinline__ void bilinear(const Corner4<W> (&c)[4], W w00, W w01, W w10, W w11, W (&o)[4]) {
  o[0] = fma(c[3].x, w11, fma(c[2].x, w10, fma(c[1].x, w01, c[0].x * w00)));
  o[1] = fma(c[3].y, w11, fma(c[2].y, w10, fma(c[1].y, w01, c[0].y * w00)));
  o[2] = fma(c[3].z, w11, fma(c[2].z, w10, fma(c[1].z, w01, c[0].z * w00)));
  o[3] = PHASE ? fma(c[3].w, w11, fma(c[2].w, w10, fma(c[1].w, w01, c[0].w * w00))) : (W)0;
}

#include "trace_f64.inc"
#include "trace_tile.inc"
#include "trace_mx.inc"

// end of a trace: the first level's queue length joins the total that stays until the counters are read
__global__ void k_carry(unsigned long long *counters) { counters[3] += counters[1]; }

// ---------------------------------------------------------------------------------------
// time-stepping form with located faces (oracle: trace_one_t) for the queued rays
// ---------------------------------------------------------------------------------------
struct Dsdt {
  double d[7];
};

// full trilinear gather at (pa, pb, pc); `clamp` holds p_a inside [g0, g1] for the lookup.  X (may be null):
// the optional terms' fields {kappa, ne, Bx, By, Bz}
__device__ void rhs_generic(const VolDev &V, bool phase, bool clamp, double pa, double pb, double pc, double (&F)[4],
                            double *X = nullptr) {
  F[0] = F[1] = F[2] = F[3] = 0.0;
  if (X) X[0] = X[1] = X[2] = X[3] = X[4] = 0.0;
  const double a0 = V.g[0][0], aL = V.g[0][V.na - 1];
  if (clamp) pa = pa < a0 ? a0 : (pa > aL ? aL : pa);
  const double b0 = V.g[1][0], bL = V.g[1][V.nb - 1], c0 = V.g[2][0], cL = V.g[2][V.nc - 1];
  if (!(pa >= a0 && pa <= aL && pb >= b0 && pb <= bL && pc >= c0 && pc <= cL)) {
    if (pa != pa || pb != pb || pc != pc) {
      F[0] = F[1] = F[2] = F[3] = __builtin_nan("");
      if (X) X[0] = X[1] = X[2] = X[3] = X[4] = __builtin_nan("");
    }
    return;
  }
  const int ia = find_cell(V.g[0], V.na, pa, a0, (V.na - 1) / (aL - a0));
  const int ib = find_cell(V.g[1], V.nb, pb, b0, (V.nb - 1) / (bL - b0));
  const int ic = find_cell(V.g[2], V.nc, pc, c0, (V.nc - 1) / (cL - c0));
  const double wa = (pa - V.g[0][ia]) * V.rg[0][ia], wb = (pb - V.g[1][ib]) * V.rg[1][ib],
               wc = (pc - V.g[2][ic]) * V.rg[2][ic];
  const int64_t q = col8(V, ib, ic) + koff(V, ia), q1 = col8(V, ib, ic) + koff(V, ia + 1);
  Corner4<double> lo[4], hi[4];
  if (phase) {
    load_plane<double, true>(V, q, lo);
    load_plane<double, true>(V, q1, hi);
  } else {
    load_plane<double, false>(V, q, lo);
    load_plane<double, false>(V, q1, hi);
  }
  const double ub = 1 - wb, uc = 1 - wc;
  double s0v[4], s1v[4];
  bilinear<double, true>(lo, ub * uc, ub * wc, wb * uc, wb * wc, s0v);
  bilinear<double, true>(hi, ub * uc, ub * wc, wb * uc, wb * wc, s1v);
  for (int m = 0; m < 4; ++m) F[m] = fma(wa, s1v[m] - s0v[m], s0v[m]);
  if (X) {
    double X0[5], X1[5];
    aux_plane(V, q, ub * uc, ub * wc, wb * uc, wb * wc, X0);
    aux_plane(V, q1, ub * uc, ub * wc, wb * uc, wb * wc, X1);
    for (int m = 0; m < 5; ++m) X[m] = fma(wa, X1[m] - X0[m], X0[m]);
  }
}

// s = (pa, pb, pc, va, vb, vc, phase, amp, pol)
template <bool AUX>
__device__ void rk4_time_step(const VolDev &V, int axis, bool phase, bool clamp, double (&s)[9], double h) {
  double k[4][9], t[9], F[4], X[5];
  for (int st = 0; st < 4; ++st) {
    const double w = st == 0 ? 0.0 : (st == 3 ? h : 0.5 * h);
    for (int q = 0; q < 9; ++q) t[q] = st == 0 ? s[q] : fma(w, k[st - 1][q], s[q]);
    rhs_generic(V, phase, clamp, t[0], t[1], t[2], F, AUX ? X : nullptr);
    k[st][0] = t[3];
    k[st][1] = t[4];
    k[st][2] = t[5];
    k[st][3] = F[0];
    k[st][4] = F[1];
    k[st][5] = F[2];
    k[st][6] = V.omega * F[3];
    k[st][7] = k[st][8] = 0.0;
    if (AUX) aux_rates(V, axis, X, t[7], t[3], t[4], t[5], k[st][7], k[st][8]);
  }
  const double h6 = h / 6.0;
  for (int q = 0; q < (AUX ? 9 : 7); ++q) s[q] = fma(h6, k[0][q] + 2.0 * k[1][q] + 2.0 * k[2][q] + k[3][q], s[q]);
}

__device__ __forceinline__ bool gone(const double *g, int n, double p, double v) {
  return (p > g[n - 1] && v >= 0) || (p < g[0] && v <= 0);
}

template <bool PHASE, bool AUX>
__global__ __launch_bounds__(256) void k_trace_time(TraceArgs A) {
  const VolDev &V = A.V;
  const unsigned long long count = *A.in_count;
  const int64_t N = A.N;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  unsigned long long mysteps = 0;
  for (unsigned long long f = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; f < count;
       f += (unsigned long long)gridDim.x * blockDim.x) {
    const int64_t j = A.in_list[f];
    const int64_t i = A.perm[j];
    double s[9] = {A.s0[a * N + i],       A.s0[b * N + i],       A.s0[c * N + i],
                   A.s0[(3 + a) * N + i], A.s0[(3 + b) * N + i], A.s0[(3 + c) * N + i],
                   A.s0[7 * N + i],       A.s0[6 * N + i],       A.s0[8 * N + i]};
    const double g0 = V.g[0][0], g1 = V.g[0][V.na - 1];
    const double t_end = A.t_end, dt = A.dt;
    double t = 0.0;
    unsigned long long n = 0;
    bool plain = !(s[3] > 0), done = false;
    if (!plain && s[0] < g0) {
      const double tau = (g0 - s[0]) / s[3];
      if (tau >= t_end) {
        done = true;
      } else {
        s[0] = g0;
        s[1] = s[1] + s[4] * tau;
        s[2] = s[2] + s[5] * tau;
        t = tau;
      }
    }
    if (!done && !plain && s[0] > g1) done = true;
    while (!done && !plain) {
      const double rem = t_end - t;
      if (!(rem > 0)) {
        done = true;
        break;
      }
      const double h = dt < rem ? dt : rem;
      if (s[0] + 1.01 * h * s[3] >= g1) {  // Newton steps onto the exit plane
        for (int it = 0; it < 3; ++it) {
          double hx = (g1 - s[0]) / s[3];
          bool last = false;
          if (t + hx > t_end) {
            hx = t_end - t;
            last = true;
          }
          rk4_time_step<AUX>(V, a, PHASE, true, s, hx);
          t += hx;
          ++n;
          if (last || !(s[3] > 0)) break;
        }
        done = true;
        break;
      }
      rk4_time_step<AUX>(V, a, PHASE, true, s, h);
      t += h;
      ++n;
      if (gone(V.g[1], V.nb, s[1], s[4]) || gone(V.g[2], V.nc, s[2], s[5])) {
        done = true;
        break;
      }
      if (!(s[3] > 0)) plain = true;
    }
    while (!done) {  // plain rule
      const double rem = t_end - t;
      if (!(rem > 0)) break;
      const double h = dt < rem ? dt : rem;
      rk4_time_step<AUX>(V, a, PHASE, false, s, h);
      t += h;
      ++n;
      if (gone(V.g[0], V.na, s[0], s[3]) || gone(V.g[1], V.nb, s[1], s[4]) || gone(V.g[2], V.nc, s[2], s[5])) break;
    }
    const double rem = t_end - t;
    if (rem > 0) {
      s[0] = s[0] + s[3] * rem;
      s[1] = s[1] + s[4] * rem;
      s[2] = s[2] + s[5] * rem;
    }
    write_outputs(A, j, s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
    mysteps += n;
  }
  if (mysteps) atomicAdd(sr::stripe(A.counters, A.step_stripe), mysteps);
}

// A3/A4 at caller-given physical points (x, y, z): out (4, N)
__global__ void k_sample(VolDev V, int axis, bool phase, const double *__restrict__ pts, int64_t N, double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int a = axis, b = (a + 1) % 3, c = (a + 2) % 3;
  const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  double F[4];
  rhs_generic(V, phase, false, p[a], p[b], p[c], F);
  if (p[0] != p[0] || p[1] != p[1] || p[2] != p[2]) F[0] = F[1] = F[2] = F[3] = __builtin_nan("");
  out[a * N + i] = F[0];
  out[b * N + i] = F[1];
  out[c * N + i] = F[2];
  out[3 * N + i] = F[3];
}

__global__ void k_sample_aux(VolDev V, int axis, const double *__restrict__ pts, int64_t N, double *__restrict__ out) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int a = axis, b = (a + 1) % 3, c = (a + 2) % 3;
  const double p[3] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  double F[4], X[5];
  rhs_generic(V, false, false, p[a], p[b], p[c], F, X);
  for (int m = 0; m < 5; ++m) out[m * N + i] = X[m];
}

// launch order -> original order (download path)
__global__ void k_unpermute(const double *__restrict__ src, double *__restrict__ dst, const uint32_t *__restrict__ perm,
                            int64_t N, int rows, int width) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= N) return;
  const int64_t i = perm[j];
  for (int r = 0; r < rows; ++r)
    for (int w = 0; w < width; ++w) dst[((int64_t)r * N + i) * width + w] = src[((int64_t)r * N + j) * width + w];
}

// Threads per workgroup of k_trace_f64 / k_trace_mx: as FEW wavefronts as the LDS node tables allow at the kernel's
// occupancy.  A workgroup's place on its CU (registers, LDS) is given to the next workgroup only when its slowest wavefront
// is done, and wavefronts differ in how many cell changes their rays make: with one-wavefront workgroups a finished
// wavefront is replaced at once.  Measured on BASELINE config 3, float64 kernel: 512 threads 68.5 ms, 256 61.8, 128 60.7,
// 64 60.0; mixed: 256 32.3, 128 32.0, and 64 only 41.4 because 16 tables of 16 KB do not fit the CU's 160 KB.
int small_block(size_t lds_bytes, int waves_per_cu) {
  for (int w = 1; w < 4; w *= 2)
    if ((size_t)(waves_per_cu / w) * lds_bytes <= (size_t)160 * 1024) return 64 * w;
  return 256;
}

VolDev vol_dev(const sr_volume *v) {
  VolDev V{};
  V.P = v->P;
  V.L = v->L;
  for (int k = 0; k < 3; ++k) {
    V.g[k] = v->g[k];
    V.rg[k] = v->rg[k];
  }
  V.na = v->na;
  V.nb = v->nb;
  V.nc = v->nc;
  V.omega = v->omega;
  V.K = v->K;
  V.Q = v->Q;
  V.verdet = v->verdet;
  V.OS = (int64_t)v->nb * v->nc * 8;
  V.R = v->R;
  return V;
}

// sf / rf / Jf of a traced bundle, original ray order, into host arrays whose rows are `ld` rays long, starting at ray
// `off` of every row (ld == r->n, off == 0: the bundle's own arrays).  `staging`: 17*r->n doubles of device memory (9 + 4 + 4
// rows: each array is put back into ray order in its own part, so the three copies need ONE wait), or nullptr to allocate
// them for the call (hipFree waits for every stream: the pipelined sr_trace passes its own).
int download_rows(const sr_rays *r, double *sf, double *rf, double *Jf, int64_t ld, int64_t off, double *staging) {
  if (!r->traced) return sr::fail(SR_ERR_STATE, "sr_rays_download: rays have not been traced");
  const int64_t N = r->n;
  if (N == 0) return SR_OK;
  hipStream_t st = sr::ctx().stream;
  double *tmp = staging;
  const size_t rows = (sf ? 9 : 0) + (rf ? 4 : 0) + (Jf ? 4 : 0);
  if (rows == 0) return SR_OK;
  if (!tmp) {  // the library's per-call staging block (kept between calls: no hipMalloc / hipFree in a loop of solve() calls)
    tmp = static_cast<double *>(sr::scratch(sizeof(double) * rows * (size_t)N));
    if (!tmp) return SR_ERR_HIP;
  }
  const unsigned grid = sr::grid_for(N, 256);
  struct Job {
    const double *src;
    double *dst;
    int rows, width;
  } jobs[3] = {{r->sf, sf, 9, 1}, {r->rf, rf, 4, 1}, {r->Jf, Jf, 2, 2}};
  hipError_t e = hipSuccess;
  double *part = tmp;
  for (auto &jb : jobs) {
    if (!jb.dst) continue;
    hipLaunchKernelGGL(k_unpermute, dim3(grid), dim3(256), 0, st, jb.src, part, (const uint32_t *)r->perm, N, jb.rows, jb.width);
    const size_t row_bytes = sizeof(double) * (size_t)jb.width * (size_t)N;
    if (ld == N)
      e = hipMemcpyAsync(jb.dst, part, row_bytes * jb.rows, hipMemcpyDeviceToHost, st);
    else  // one copy per row (2-D copies of pageable memory are staged row by row anyway)
      for (int q = 0; q < jb.rows && e == hipSuccess; ++q)
        e = hipMemcpyAsync(jb.dst + ((size_t)q * ld + off) * jb.width, part + (size_t)q * N * jb.width, row_bytes, hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) break;
    part += (size_t)jb.rows * jb.width * (size_t)N;
  }
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (!staging) sr::scratch_trim();
  if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_rays_download: %s", hipGetErrorString(e));
  return SR_OK;
}


// Two-level counting sort of r->keys (see k_bin_count): coarse digit = key >> lo_bits (n_coarse of them), fine digit below;
// out[pos] = index of the ray that comes pos-th.
int bin_rays(sr_rays *r, int64_t N, int lo_bits, int n_coarse, uint32_t *out, hipStream_t st) {
  SR_CHECK(lo_bits <= 11 && n_coarse <= kBinMaxDigits + 1, "lateral grid too large for the LDS counters of the ray binning");
  const unsigned n_wg = (unsigned)((N + kBinTile - 1) / kBinTile);
  const int64_t ncount = (int64_t)n_coarse * n_wg;
  const int64_t nsb = (ncount + kScanPerBlock - 1) / kScanPerBlock;  // scan workgroups; their totals follow the counts
  if (r->bins_cap < ncount + nsb) {
    sr::dev_free(r->bins);
    r->bins = nullptr;
    r->bins_cap = 0;
    int rc = sr::dev_alloc(&r->bins, (size_t)(ncount + nsb));
    if (rc) return rc;
    r->bins_cap = ncount + nsb;
  }
  if (!r->sort_tmp) {  // by the bundle's capacity: the pipeline's short last chunk may be the first to come through here
    int rc = sr::dev_alloc(&r->sort_tmp, (size_t)2 * (size_t)std::max(r->cap, N));
    if (rc) return rc;
  }
  uint32_t *sums = r->bins + ncount, *tkeys = r->sort_tmp, *trays = r->sort_tmp + N;
  hipLaunchKernelGGL(k_bin_count, dim3(n_wg), dim3(256), 0, st, (const uint32_t *)r->keys, N, lo_bits, n_coarse, r->bins, n_wg);
  hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nsb), dim3(256), 0, st, r->bins, ncount, sums);
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, sums, nsb);
  hipLaunchKernelGGL(k_scan_add, dim3(sr::grid_for(ncount, 256)), dim3(256), 0, st, r->bins, ncount, (const uint32_t *)sums);
  hipLaunchKernelGGL(k_bin_coarse, dim3(n_wg), dim3(256), 0, st, (const uint32_t *)r->keys, N, lo_bits, n_coarse,
                     (const uint32_t *)r->bins, n_wg, tkeys, trays);
  hipLaunchKernelGGL(k_bin_fine, dim3((unsigned)n_coarse), dim3(1024), 0, st, (const uint32_t *)tkeys, (const uint32_t *)trays, N,
                     lo_bits, n_coarse, (const uint32_t *)r->bins, n_wg, out);
  return SR_OK;
}

// the band order (trace_tile.inc): keys from s0 (rec == nullptr) or from the hand-off records, then the sort
int bin_by_band(sr_rays *r, const sr_volume *v, const TileGeom &g, const double *rec, uint32_t *out, hipStream_t st) {
  const int64_t N = r->n;
  const int64_t n_bands = (v->nb - 1 + g.band - 1) / g.band;
  const int64_t key_max = n_bands * (int64_t)(v->nc - 1) * g.band;  // the key of rays outside the volume / dead rays
  int bits = 1;
  while (((int64_t)1 << bits) <= key_max) ++bits;
  const int lo_bits = std::min(11, (bits + 1) / 2);
  const int n_coarse = (int)(key_max >> lo_bits) + 1;
  SR_CHECK(n_coarse <= kBinMaxDigits + 1, "lateral grid too large for the LDS counters of the ray binning");
  const unsigned nblk = sr::grid_for(N, 256);
  if (rec)
    hipLaunchKernelGGL(k_keys_band_rec, dim3(nblk), dim3(256), 0, st, vol_dev(v), rec, N, g.band, (uint32_t)key_max, r->keys);
  else
    hipLaunchKernelGGL(k_keys_band, dim3(nblk), dim3(256), 0, st, vol_dev(v), (const double *)r->s0, N, v->axis, g.band, (uint32_t)key_max, r->keys);
  return bin_rays(r, N, lo_bits, n_coarse, out, st);
}

// ---- the tile path (trace_tile.inc) -------------------------------------------------------------------------------------------
// Taken by dense float64 bundles (the coefficient records a workgroup builds are shared by the rays of a cell, and its rays have
// to fit a tile); SYNTHRAY_F64_TILE=0 / 1 forces the per-ray kernel / the tile kernel wherever it can run.  (A tile kernel for
// the mixed build existed in round 3, bit-identical to k_trace_mx and slower -- 35.5 against 31.3 ms on BASELINE config 3,
// profiles/r03_mxt_experiments.txt: k_trace_mx has no conversions to shed -- and was removed in round 4.)
// SYNTHRAY_TILE="tb,tc,halo,band,planes per segment" overrides the geometry.
constexpr double kTileMinDensity = 8.0;
struct TilePlan {
  TileGeom g;
  int seg;   // node planes per segment
  bool rec;  // the records kernel (trace_tile.inc, REC): sr_volume::R exists
};
// lateral cells the beam covers: its launch positions' bounding box in cells of the two lateral axes (whole grid when unknown)
double beam_cells(const sr_rays *r, const sr_volume *v) {
  const double all = (double)(v->nb - 1) * (double)(v->nc - 1);
  if (!r || !r->have_bbox) return all;
  double cells = 1.0;
  for (int q = 1; q <= 2; ++q) {  // volume axes b, c = physical axes (axis + q) % 3
    const int phys = (v->axis + q) % 3, n = q == 1 ? v->nb : v->nc;
    const std::vector<double> &g = v->hg[q];
    if (n < 2 || g.size() < 2) return all;
    const double lo = std::max(r->bbox[phys], g.front()), hi = std::min(r->bbox[3 + phys], g.back());
    const double width = (g.back() - g.front()) / (n - 1);
    double span = hi > lo ? (hi - lo) / width + 1.0 : 1.0;
    cells *= std::min(span, (double)(n - 1));
  }
  return cells;
}
// The tile path's ready-made records (sr_volume::R): use = they exist after this call.  Built once per volume, at the first trace
// that takes the tile path without the optional terms, when they fit beside everything else (at most a third of the free HBM:
// 128 bytes per node plane and lateral cell, 17 GB for 512^3), with the volume's own arithmetic (k_build_records =
// coefs_from_corners); otherwise -- and with SYNTHRAY_TILE_RECORDS=0 -- the producers' kernel runs.
int tile_records(const sr_volume *v, hipStream_t st, bool &use) {
  use = false;
  const char *e = getenv("SYNTHRAY_TILE_RECORDS");
  if (e && e[0] == '0') return SR_OK;
  if (!v->R && !v->R_tried) {
    v->R_tried = true;
    const size_t count = (size_t)v->na * (size_t)(v->nb - 1) * (size_t)(v->nc - 1) * 16;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || count * sizeof(double) > free_b / 3) return SR_OK;
    double *R = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&R), count * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      return SR_OK;
    }
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)(count / 16 + 255) / 256, (int64_t)sr::ctx().n_cu * 64);
    if (v->L)
      hipLaunchKernelGGL(k_build_records<true>, dim3(grid), dim3(256), 0, st, vol_dev(v), R);
    else
      hipLaunchKernelGGL(k_build_records<false>, dim3(grid), dim3(256), 0, st, vol_dev(v), R);
    SR_HIP(hipGetLastError());
    // The volume is shared by every stream that traces through it (the job driver alternates its bundles between the library's two
    // streams): the records must be COMPLETE before anybody can see the pointer.  Once per volume: 15 ms for 512^3.
    SR_HIP(hipStreamSynchronize(st));
    v->R = R;
  }
  use = v->R != nullptr;
  return SR_OK;
}

constexpr int64_t kTileMinRays = 32768;  // below this either kernel is a handful of wavefronts: the per-ray kernel spreads them wider

bool tile_plan(const sr_rays *r, const sr_volume *v, const sr_trace_params *p, int64_t N, TilePlan &tp, hipStream_t st) {
  // the producers' kernel, measured on BASELINE config 3 in round 3 (docs/HISTORY.md, profiles/r03_tile_geometry_ab.txt): 256-ray workgroups, 8 x 8 tiles, bands of
  // two cell rows, 171-plane segments 50.6 ms per step (128 planes: 51.6; 256 planes, where only two workgroups fit a CU: 65.8);
  // 768-ray workgroups with 12 x 16 tiles: 128 / 171 / 256 / 511 planes per segment 56.0 / 54.0 / 52.5 / 61.6
  if (p->precision == SR_PREC_MIXED && p->substeps == 1 && !v->K && !v->Q) return false;  // k_trace_mx's traces
  // inverse bremsstrahlung alone: the per-ray kernel that carries kappa only (trace_f64.inc, SEL = 1; two wavefronts per SIMD) is the
  // faster one even for dense bundles -- 4.5 ms against the tile path's 7.3 (five-field records) on 1e6 rays x 256^3,
  // profiles/r05_aux_sparse.txt, r05_aux_rate.txt -- unless the tile path is forced
  if (v->K && !v->Q && p->substeps == 1 && !(getenv("SYNTHRAY_F64_TILE") && getenv("SYNTHRAY_F64_TILE")[0] == '1')) return false;
  tp = TilePlan{{8, 8, 2, 2}, 171, false};
  const double density = (double)N / beam_cells(r, v);
  {
    // rows per band of the ray order: the 256 rays of a workgroup should cover a SQUARE patch of cells (256 / density of them),
    // so that it fits the 8 x 8 tile with room to drift: 2 rows at the headline's 60 rays per cell, 3 at 24, 5-6 at 12 and below
    // (measured, same file: at 15 rays per beam cell 2 rows lose 25 % of the rays per segment and 14.7 ms per step, 6 rows 12.6)
    const int rows = (int)std::lround(std::sqrt(256.0 / std::max(density, 1.0)));
    tp.g.band = std::min(6, std::max(2, rows));
    if (tp.g.band >= 5) tp.seg = 128;
    if (v->K || v->Q) {  // the optional terms' records are 304 bytes: a 6 x 7 tile (64 KB), two workgroups per CU
      tp.g.tb = 6;
      tp.g.tc = 7;
      tp.g.band = std::min(tp.g.band, 4);
      tp.seg = 128;
    }
  }
  const char *on = getenv("SYNTHRAY_F64_TILE");
  if (on && on[0] == '0') return false;
  const bool forced = on && on[0] == '1';
  if (!forced && (N < kTileMinRays || density < kTileMinDensity)) return false;
  const int threads = SR_TILE_THREADS;
  // slabs (A12) are admitted: the kernel steps a range of node planes from / to hand-off records anyway; so are the optional
  // terms (AUX: five more fields per record, a smaller tile, two workgroups per CU)
  if (p->substeps != 1 || !p->sort_rays) return false;
  const bool aux = v->K || v->Q;
  if (!aux && threads == 256 && v->nb - 1 >= 8 && v->nc - 1 >= 7) {
    // The records kernel (trace_tile.inc, REC; round 5): 8 x 7 tiles, four workgroups per CU.  Measured on 512^3 (tools/
    // r05_rec4_sweep.sh, r05_low_density.sh; profiles/r05_tile_variants.txt): at 60 rays per cell of the beam's box (BASELINE config 3)
    // bands of 3 / 4 / 5 rows 45.0 / 44.1 / 45.3 ms per step in 128-plane segments, 4 rows in 171- / 103-plane segments 45.5 / 45.3;
    // at 30 rays per cell 4 rows 24.1 (3: 25.4, 5: 24.6); at 15 rays per cell 5 or 6 rows in 103-plane segments 14.4 / 14.3 (128
    // planes: 14.9 / 14.7; 4 rows: 15.4) -- one or two rows more per band than the producers' kernel wants, and shorter segments.
    bool rec = false;
    if (tile_records(v, st, rec) != SR_OK) rec = false;
    if (rec) {
      tp.rec = true;
      tp.g.tc = 7;
      tp.g.band = std::min(6, std::max(4, tp.g.band + 1));
      tp.seg = tp.g.band >= 5 ? 103 : 128;
    }
  }
  if (const char *e = getenv("SYNTHRAY_TILE")) {
    int a, b, c, d, f;
    if (sscanf(e, "%d,%d,%d,%d,%d", &a, &b, &c, &d, &f) == 5 && a >= 2 && b >= 2 && c >= 0 && d >= 1 && f >= 1 && a * b <= threads) {
      tp.g = TileGeom{a, b, c, d};
      tp.seg = f;
      tp.rec = tp.rec && a == 8 && b <= 8;  // a tile column is one DMA's kilobyte: eight cells
    }
  }
  if (v->nb - 1 < tp.g.tb || v->nc - 1 < tp.g.tc || v->na < 3) return false;
  if (aux && (tp.g.tb * tp.g.tc > 64 || threads < 128)) return false;  // one wavefront of producers per kind
  {
    const int steps = v->na - 1, n_seg = (steps + tp.seg - 1) / tp.seg;
    if (tile_lds_bytes(tp.g, (steps + n_seg - 1) / n_seg + 1, aux, tp.rec) > (size_t)160 * 1024) return false;
  }
  return true;
}

void launch_planes64(const sr_volume *v, const sr_trace_params *p, TraceArgs &A, hipStream_t st);

// The segments' shares of the node planes (equal until measured otherwise: tools/r05_cuts.sh)
// Measured on BASELINE config 3 (tools/r05_cuts.sh, profiles/r05_tile_variants.txt): shares 1 : 1 : 1 48.72 ms per step and 633 000
// stragglers, 1.3 : 1 : 0.7 48.31 ms and 494 000, 1.5 : 1.1 : 0.4 50.6 ms.  Rays start parallel and pick their angles up on the
// way, so a tile loses few rays in the first planes and most in the last; and the LAST segment's stragglers are the only ones with
// no tile launch to run beside.  A ramp from 1.3 down to 0.7.
void tile_cut_weights(int n_seg, std::vector<double> &w) {
  for (int q = 0; q < n_seg; ++q) w[(size_t)q] = n_seg > 1 ? 1.3 - 0.6 * q / (n_seg - 1) : 1.0;
}

// The tile kernel over every ray, segment by segment.  What a segment's launch loses (rays leaving their workgroup's tile or
// the volume: 2 % of the rays per segment on BASELINE config 3) becomes a STRAGGLER (trace_tile.inc): its state on the
// segment's first plane is appended to r->strag_rec, its own record is marked gone, and k_trace_f64 carries the segment's new
// entries from that plane to the END of the volume (slab) on the library's side stream, beside the remaining segments;
// k_strag_finish, after the last segment and the side stream, forms the gone slots' outputs (exit records) from the entries.
// k_trace_f64 from a record is k_trace_f64 from s0 (the slab chain of A12), and the tile kernel is k_trace_f64 ray for ray,
// so the result is the per-ray kernel's, bit for bit, whoever carried a ray where.  Rays that are no plane-form rays at all
// end in r->fb_list (counters[1]) for the usual levels, from s0.
int trace_tiled(sr_rays *r, const sr_volume *v, const sr_trace_params *p, const TilePlan &tp, TraceArgs &A, hipStream_t st) {
  const int64_t N = r->n;
  const bool phase = v->L != nullptr;
  static bool attr_set = false;
  if (!attr_set) {
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_tile<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_tile<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_tile<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_tile<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_tile<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    SR_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_trace_tile<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  sr::Context &c = sr::ctx();
  hipStream_t side = c.side[c.current];
  hipEvent_t ev_go = c.side_ev[c.current][0], ev_done = c.side_ev[c.current][1];
  const int steps = v->na - 1;
  const int n_seg = (steps + tp.seg - 1) / tp.seg;
  const bool aux = v->K != nullptr || v->Q != nullptr;
  const int threads = SR_TILE_THREADS;
  // REC (trace_tile.inc): the node planes' coefficient records ready-made in HBM, brought into the tile's ring by LDS-DMA.  128 bytes
  // per (node plane, lateral cell): 17 GB for 512^3, built at the first trace that takes this path and kept with the volume.
  const bool rec = tp.rec && v->R != nullptr;
  const size_t lds = tile_lds_bytes(tp.g, (steps + n_seg - 1) / n_seg + 1, aux, rec);
  A.V = vol_dev(v);  // with the records, if they have just been built
  const bool ho_enter = (p->handoff & SR_HANDOFF_ENTER) != 0, ho_exit = (p->handoff & SR_HANDOFF_EXIT) != 0;
  const size_t cap = (size_t)std::max(r->cap, N);  // never by the current n (a short chunk in a full-size bundle)
  {
    int rc = SR_OK;
    if (!r->rec && (rc = sr::dev_alloc(&r->rec, 10 * cap))) return rc;
    if (n_seg > 1 || ho_enter) {
      if (!r->rec2 && (rc = sr::dev_alloc(&r->rec2, 10 * cap))) return rc;
      if (!r->order2 && (rc = sr::dev_alloc(&r->order2, cap))) return rc;
    }
    // a ray is lost once: the straggler records never hold more entries than the bundle has rays
    if (!r->strag_rec && (rc = sr::dev_alloc(&r->strag_rec, 10 * cap))) return rc;
    if (r->strag_snap_cap < n_seg + 2) {  // one entry count per launch (the last segment may go in two) + the zero in front
      sr::dev_free(r->strag_snap);
      r->strag_snap = nullptr;
      r->strag_snap_cap = 0;
      if ((rc = sr::dev_alloc(&r->strag_snap, (size_t)std::max(n_seg + 2, 64)))) return rc;
      r->strag_snap_cap = std::max(n_seg + 2, 64);
    }
  }
  const unsigned nb = sr::grid_for(N, threads);
  unsigned long long *strag_count = r->counters + 8;
  SR_HIP(hipMemsetAsync(strag_count, 0, sizeof(unsigned long long), st));
  SR_HIP(hipMemsetAsync(r->strag_snap, 0, sizeof(unsigned long long), st));  // snap[0] = 0
  TileArgs T{};
  T.G = tp.g;
  {
    // Opt-in: measured SLOWER (profiles/r04_tile_variants.txt: 47.5 against 45.3 ms in the kernels on C3) -- the three producer
    // wavefronts of a CU were evidently not on one SIMD to begin with
    const char *e = getenv("SYNTHRAY_TILE_ROTATE");
    T.rot = (e && e[0] == '1') ? 1 : 0;
  }
  // SYNTHRAY_STRAGGLERS=serial: the stragglers' launches on the library stream itself, each behind its segment (for A/B timing)
  const char *se = getenv("SYNTHRAY_STRAGGLERS");
  const bool beside = !(se && se[0] == 's');
  // SYNTHRAY_TILE_LAST_SPLIT=f (opt-in, MEASURED NULL): the last segment in two launches, the first f of its workgroups and the
  // rest.  The thought: the stragglers of every other launch are carried beside the tile kernel's next launch, the last launch's
  // have nothing to run beside (1.6 ms alone with the producers' kernel and equal segments, profiles/r05_timeline_c3_producers_equal_
  // cuts.txt), and a quarter of the workgroups would leave one round of wavefronts, 0.4 ms.  Measured (profiles/r05_tile_variants.txt):
  // 48.66 ms per step in one launch, 48.78 - 48.98 cut at 0.5 / 0.65 / 0.75, whatever the side stream's priority -- beside a tile
  // launch the per-ray kernel only gets the CU slots the tile kernel's own start and end leave, so the first part's stragglers
  // finish AFTER the second part.  What shortened the tail instead: a shorter last segment (tile_cut_weights).
  double last_split = 0.0;
  if (const char *e = getenv("SYNTHRAY_TILE_LAST_SPLIT")) last_split = atof(e);
  unsigned nb_first = nb;
  if (beside && last_split > 0.0 && last_split < 1.0 && nb >= 2048) nb_first = std::min(nb - 8, std::max(8u, (unsigned)(nb * last_split) / 8 * 8));
  int launch = 0;  // launches so far = straggler snapshots taken
  // Segment boundaries.  SYNTHRAY_TILE_CUTS="w0,w1,..." (n_seg weights): the segments' shares of the node planes
  std::vector<int> cut((size_t)n_seg + 1, 0);
  {
    std::vector<double> w((size_t)n_seg, 1.0);
    const char *e = getenv("SYNTHRAY_TILE_CUTS");
    if (e && *e) {
      std::vector<double> given;
      for (const char *q = e; *q;) {
        char *end = nullptr;
        const double x = strtod(q, &end);
        if (end == q) break;
        given.push_back(x);
        q = *end == ',' ? end + 1 : end;
      }
      bool ok = (int)given.size() == n_seg;
      for (double x : given) ok = ok && x > 0;
      if (ok)
        w = given;
      else
        tile_cut_weights(n_seg, w);
    } else {
      tile_cut_weights(n_seg, w);
    }
    double tot = 0, acc = 0;
    for (double x : w) tot += x;
    for (int q = 0; q < n_seg; ++q) {
      acc += w[(size_t)q];
      cut[(size_t)q + 1] = q + 1 == n_seg ? steps : std::min(steps - (n_seg - 1 - q), std::max(cut[(size_t)q] + 1, (int)std::lround(steps * acc / tot)));
    }
  }
  for (int q = 0; q < n_seg; ++q) {
    T.k0 = cut[(size_t)q];
    T.k1 = cut[(size_t)q + 1];
    T.first = q == 0 && !ho_enter;
    T.last = q + 1 == n_seg;
    T.exit_rec = ho_exit ? 1 : 0;
    T.slab = (v->is_slab || p->handoff) ? 1 : 0;
    T.A = A;
    T.A.guard = r->guard;
    T.perm_out = r->perm;
    T.strag_rec = r->strag_rec;
    T.strag_cap = (int64_t)cap;
    T.strag_count = strag_count;
    if (T.first) {
      T.rec_in = nullptr;
      T.order = nullptr;
      T.A.rec = r->rec;
    } else {
      // bin the rays again by the cell they are in now (a slab's arrivals: the sender's order is its ENTRY cells'); the kernel
      // reads the records through the new order and writes the other buffer in place; the ray index rides in row 9
      int rc = bin_by_band(r, v, tp.g, r->rec, r->order2, st);
      if (rc) return rc;
      T.rec_in = r->rec;
      T.order = r->order2;
      T.A.rec = r->rec2;
    }
    const bool timed = n_seg <= sr::kMaxTileSegs;
    if (timed) SR_HIP(hipEventRecord(c.ev[4 + 2 * q], st));
    const int parts = (T.last && nb_first < nb) ? 2 : 1;
    for (int part = 0; part < parts; ++part) {
      T.block0 = part == 0 ? 0u : nb_first;
      T.A.n_blocks = parts == 1 ? nb : (part == 0 ? nb_first : nb - nb_first);
      const unsigned grid = ((T.A.n_blocks + 7) / 8) * 8;
      if (aux) {
        if (phase)
          hipLaunchKernelGGL((k_trace_tile<true, true>), dim3(grid), dim3(SR_TILE_THREADS), lds, st, T);
        else
          hipLaunchKernelGGL((k_trace_tile<false, true>), dim3(grid), dim3(SR_TILE_THREADS), lds, st, T);
      } else if (rec) {
        if (phase)
          hipLaunchKernelGGL((k_trace_tile<true, false, true>), dim3(grid), dim3(SR_TILE_THREADS), lds, st, T);
        else
          hipLaunchKernelGGL((k_trace_tile<false, false, true>), dim3(grid), dim3(SR_TILE_THREADS), lds, st, T);
      } else if (phase) {
        hipLaunchKernelGGL((k_trace_tile<true>), dim3(grid), dim3(SR_TILE_THREADS), lds, st, T);
      } else {
        hipLaunchKernelGGL((k_trace_tile<false>), dim3(grid), dim3(SR_TILE_THREADS), lds, st, T);
      }
      if (timed && part + 1 == parts) SR_HIP(hipEventRecord(c.ev[5 + 2 * q], st));
      hipLaunchKernelGGL(k_strag_snap, dim3(1), dim3(1), 0, st, (const unsigned long long *)strag_count, r->strag_snap, launch, r->counters + 3);
      // this launch's stragglers from node plane k0 to the end of the volume (slab), entry by entry in the straggler records
      TraceArgs S = A;
      S.N = (int64_t)cap;  // the records' row pitch
      S.rec = r->strag_rec;
      S.guard = nullptr;
      S.in_list = nullptr;
      S.in_iota = 1;
      S.in_first = r->strag_snap + launch;
      S.in_count = r->strag_snap + launch + 1;
      S.out_list = nullptr;  // what the plane form cannot finish comes back NaN in its entry: k_strag_finish sends it on
      S.out_count = nullptr;
      S.handoff = SR_HANDOFF_ENTER | SR_HANDOFF_EXIT;
      S.k_first = T.k0;
      S.k_last = -1;
      S.recover = 0;
      S.finish_later = ho_exit ? 0 : 1;
      hipStream_t ss = st;
      if (beside) {
        SR_HIP(hipEventRecord(ev_go, st));
        SR_HIP(hipStreamWaitEvent(side, ev_go, 0));
        ss = side;
      }
      launch_planes64(v, p, S, ss);
      ++launch;
    }
    if (!T.first) std::swap(r->rec, r->rec2);  // r->rec: the buffer this segment wrote
  }
  if (beside) {
    SR_HIP(hipEventRecord(ev_done, side));
    SR_HIP(hipStreamWaitEvent(st, ev_done, 0));
  }
  {
    StragFinish F{};
    F.A = A;
    F.A.rec = r->rec;
    F.A.guard = r->guard;
    F.strag = r->strag_rec;
    F.cap = (int64_t)cap;
    F.exit_rec = ho_exit ? 1 : 0;
    F.slab = (v->is_slab || p->handoff) ? 1 : 0;
    hipLaunchKernelGGL(k_strag_finish, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, F);
  }
  A.guard = r->guard;
  r->tile_segs = n_seg <= sr::kMaxTileSegs ? n_seg : 0;
  r->tile_segs_run = n_seg;
  r->tile_rec = rec;
  return SR_OK;
}

// ---- the launch of one trace: arguments, step table, and the chain of levels -------------------------------------------
// Shared by sr_rays_trace and the edge guard's re-trace (sr::retrace_f64).
int make_trace_args(const sr_rays *r, const sr_volume *v, const sr_trace_params *p, TraceArgs &A) {
  const int64_t N = r->n;
  A = TraceArgs{};
  A.V = vol_dev(v);
  A.s0 = r->s0;
  A.N = N;
  A.perm = r->perm;
  A.sf = r->sf;
  A.rf = r->rf;
  A.Jf = r->Jf;
  A.t_end = p->t_end;
  A.extent = p->extent;
  A.dt = p->dt > 0 ? p->dt : (v->hg[0][1] - v->hg[0][0]) / sr::kC;
  A.axis = v->axis;
  A.row_order = p->row_order;
  A.sub = p->substeps;
  A.counters = r->counters;
  A.in_list = nullptr;
  A.in_count = nullptr;
  A.out_list = r->fb_list;
  A.out_count = r->counters + 1;
  A.n_blocks = sr::grid_for(N, 256);
  A.rec = r->rec;
  A.handoff = p->handoff;
  A.guard = r->guard;
  A.step_stripe = 0;
  A.k_first = 0;
  A.k_last = -1;
  A.recover = 0;
  A.in_first = nullptr;
  A.in_iota = 0;
  A.finish_later = 0;
  {
    double hmax = 0;
    for (int k = 0; k + 1 < v->na; ++k) hmax = std::max(hmax, v->hg[0][k + 1] - v->hg[0][k]);
    A.guard_h6 = (float)(hmax / 6.0 * (1.0 + 1e-6));
    A.guard_ih6 = A.guard_h6 > 0 ? 1.f / A.guard_h6 : 0.f;
  }
  {  // step table: cached on the volume per `substeps` (built and copied once, synchronously: no host buffer outlives
     // this call and no in-flight kernel ever sees a table being rewritten)
    const int sub = p->substeps;
    const int64_t nt = (int64_t)(v->na - 1) * sub;
    auto it = v->step_tabs.find(sub);
    if (it == v->step_tabs.end()) {
      // first half: the float64 kernels' table (the mixed build runs k_trace_planes as its second level); second half:
      // the mixed kernel's
      std::vector<StepTab> tab((size_t)(2 * nt));
      const std::vector<double> &g = v->hg[0];
      for (int k = 0; k + 1 < v->na; ++k) {
        const double zk = g[k], zk1 = g[k + 1], rz = 1.0 / (zk1 - zk), dz = (zk1 - zk) / sub;
        for (int m = 0; m < sub; ++m) {
          const double za = zk + m * dz, zb = (m + 1 == sub) ? zk1 : zk + (m + 1) * dz;
          {  // the oracle's step arithmetic (trace_one_planes), plane weights by reciprocal cell width
            StepTab64 &D = reinterpret_cast<StepTab64 &>(tab[(size_t)k * sub + m]);
            D.h = zb - za;
            D.hh = 0.5 * D.h;
            D.h6 = D.h / 6.0;
            D.wa0 = (za - zk) * rz;
            D.waH = (za + D.hh - zk) * rz;
            D.wa1 = (m + 1 == sub) ? 1.0 : (zb - zk) * rz;
            D.h6w = D.h6 * v->omega;
            D.pad[0] = 0.0;
          }
          StepTab &T = tab[(size_t)nt + (size_t)k * sub + m];
          T.h = zb - za;
          T.hh = 0.5 * T.h;
          T.h6 = T.h / 6.0;
          T.h6w = T.h6 * v->omega;
          T.hf = (float)T.h;
          T.hhf = (float)T.hh;
          T.wa0 = (float)((za - zk) * rz);
          T.waH = (float)((za + T.hh - zk) * rz);
          T.wa1 = (m + 1 == sub) ? 1.f : (float)((zb - zk) * rz);
          T.pad[0] = T.pad[1] = T.pad[2] = 0.f;
        }
      }
      void *d = nullptr;
      SR_HIP(hipMalloc(&d, sizeof(StepTab) * (size_t)(2 * nt)));
      hipError_t e = hipMemcpy(d, tab.data(), sizeof(StepTab) * (size_t)(2 * nt), hipMemcpyHostToDevice);  // synchronous
      if (e != hipSuccess) {
        (void)hipFree(d);
        return sr::fail(SR_ERR_HIP, "step table upload failed: %s", hipGetErrorString(e));
      }
      it = v->step_tabs.emplace(sub, d).first;
    }
    A.tab64 = static_cast<const StepTab64 *>(it->second);
    A.tab = static_cast<const StepTab *>(it->second) + nt;
  }
  const size_t lds = sizeof(double) * 2 * (size_t)(v->nb + v->nc);
  SR_CHECK(lds <= 96 * 1024, "lateral grid too large for the LDS coordinate tables (%zu bytes)", lds);
  return SR_OK;
}

// the float64 plane level over every slot (A.in_list == nullptr) or over a queue: k_trace_f64 in the instantiation the
// volume and the parameters ask for (phase integral, optional terms, sub-steps)
void launch_planes64(const sr_volume *v, const sr_trace_params *p, TraceArgs &A, hipStream_t st) {
  const int64_t N = A.N;
  const size_t lds = sizeof(double) * 2 * (size_t)(v->nb + v->nc);
  const bool phase = v->L != nullptr;
  const bool aux = v->K != nullptr || v->Q != nullptr;  // amp / pol terms (A7)
  const bool subs = p->substeps != 1;
  const bool five = aux && (subs || v->Q != nullptr);  // the five-field kernel: one wavefront per SIMD
  const int block = small_block(lds, 4 * (five ? SR_F64K_AUX_WAVES : SR_F64K_WAVES));  // else 2 wavefronts per SIMD
  const unsigned nb64 = sr::grid_for(N, block);
  const unsigned grid = ((nb64 + 7) / 8) * 8;
  const unsigned saved = A.n_blocks;
  A.n_blocks = nb64;
#define SR_F64(PH, AX, SB) hipLaunchKernelGGL((k_trace_f64<PH, AX, SB>), dim3(grid), dim3(block), lds, st, A)
#define SR_F64_SEL(PH, SEL) hipLaunchKernelGGL((k_trace_f64<PH, true, false, SEL>), dim3(grid), dim3(block), lds, st, A)
  if (aux && !subs && v->K != nullptr && v->Q == nullptr && !(getenv("SYNTHRAY_AUX_ONE_PASS") && getenv("SYNTHRAY_AUX_ONE_PASS")[0] == '1')) {
    // inverse bremsstrahlung alone: the kernel that carries kappa and nothing of the Faraday term (trace_f64.inc, SEL = 1): two
    // wavefronts per SIMD.  (SYNTHRAY_AUX_ONE_PASS=1: the five-field kernel, for A/B timing.)
    if (phase) SR_F64_SEL(true, 1); else SR_F64_SEL(false, 1);
    A.n_blocks = saved;
    return;
  }
  switch ((phase ? 4 : 0) | (aux ? 2 : 0) | (subs ? 1 : 0)) {
    case 0: SR_F64(false, false, false); break;
    case 1: SR_F64(false, false, true); break;
    case 2: SR_F64(false, true, false); break;
    case 3: SR_F64(false, true, true); break;
    case 4: SR_F64(true, false, false); break;
    case 5: SR_F64(true, false, true); break;
    case 6: SR_F64(true, true, false); break;
    default: SR_F64(true, true, true); break;
  }
#undef SR_F64_SEL
#undef SR_F64
  A.n_blocks = saved;
}

// k_trace_mx over every slot (A.in_list == nullptr) or over a queue
void launch_mx(const sr_volume *v, TraceArgs &A, hipStream_t st) {
  const size_t ml = mixed_lds_bytes(v->nb, v->nc);
  const int block = std::max(128, small_block(ml, 16));  // 4 wavefronts per SIMD; 64 and 128 measure the same
  const unsigned nbx = sr::grid_for(A.N, block);
  const unsigned grid = ((nbx + 7) / 8) * 8;
  const unsigned saved = A.n_blocks;
  A.n_blocks = nbx;
  if (v->L != nullptr)
    hipLaunchKernelGGL((k_trace_mx<true>), dim3(grid), dim3(block), ml, st, A);
  else
    hipLaunchKernelGGL((k_trace_mx<false>), dim3(grid), dim3(block), ml, st, A);
  A.n_blocks = saved;
}

// time-stepping form for what the plane form cannot take: fixed small grid, strides over the device-side count
void launch_time(const sr_volume *v, TraceArgs &A, hipStream_t st) {
  const int block = 256;
  const unsigned nblk = sr::grid_for(A.N, block);
  const bool phase = v->L != nullptr;
  const bool aux = v->K != nullptr || v->Q != nullptr;
  A.in_list = A.out_list;
  A.in_count = A.out_count;
  A.out_list = nullptr;
  A.out_count = nullptr;
  const unsigned fgrid = (unsigned)std::min<int64_t>(nblk, (int64_t)sr::ctx().n_cu * 4);
  if (aux) {
    if (phase)
      hipLaunchKernelGGL((k_trace_time<true, true>), dim3(fgrid), dim3(block), 0, st, A);
    else
      hipLaunchKernelGGL((k_trace_time<false, true>), dim3(fgrid), dim3(block), 0, st, A);
  } else if (phase) {
    hipLaunchKernelGGL((k_trace_time<true, false>), dim3(fgrid), dim3(block), 0, st, A);
  } else {
    hipLaunchKernelGGL((k_trace_time<false, false>), dim3(fgrid), dim3(block), 0, st, A);
  }
}

}  // namespace

// Edge guard (deposit.hip): the launch slots list[0..*count) of a bundle the mixed build traced are traced again by the
// float64 levels, from s0, into the same slots (guard 0); their steps go to striped total 2, their own rejects to the
// time-stepping form through r->keys / counters[5].  Queued on the current stream.
int sr::retrace_f64(const sr_rays *r, const uint32_t *list, const unsigned long long *count) {
  const sr_volume *v = r->last_vol;
  SR_CHECK(v != nullptr && r->guard_live, "edge guard: no mixed-precision trace of a whole volume to refine");
  sr_trace_params p = r->last_p;
  p.precision = SR_PREC_F64;
  hipStream_t st = sr::ctx().stream;
  TraceArgs A;
  int rc = make_trace_args(r, v, &p, A);
  if (rc) return rc;
  A.step_stripe = 2;
  A.in_list = list;
  A.in_count = count;
  A.out_list = r->keys;
  A.out_count = r->counters + 5;
  launch_planes64(v, &p, A, st);
  launch_time(v, A, st);
  SR_HIP(hipGetLastError());
  return SR_OK;
}

extern "C" {

void sr_rays_destroy(sr_rays *r) {
  if (!r) return;
  sr::dev_free(r->s0);
  sr::dev_free(r->sf);
  sr::dev_free(r->rf);
  sr::dev_free(r->Jf);
  sr::dev_free(r->perm);
  sr::dev_free(r->keys);
  sr::dev_free(r->bins);
  sr::dev_free(r->sort_tmp);
  sr::dev_free(r->fb_list);
  sr::dev_free(r->counters);
  sr::dev_free(r->rec);
  sr::dev_free(r->rec2);
  sr::dev_free(r->order2);
  sr::dev_free(r->strag_rec);
  sr::dev_free(r->strag_snap);
  sr::dev_free(r->guard);
  sr::dev_free(r->guard_set);
  delete r;
}

int sr_rays_create(sr_rays **out, int64_t n) {
  SR_CHECK(out != nullptr, "sr_rays_create: NULL out");
  *out = nullptr;
  SR_CHECK(n >= 0 && n < (int64_t)0xFFFFFFFFll, "sr_rays_create: ray count %lld out of range", (long long)n);
  int rc = sr::ensure_init();
  if (rc) return rc;
  sr_rays *r = new sr_rays();
  r->n = n;
  r->cap = n;
  const size_t m = (size_t)(n > 0 ? n : 1);
  if ((rc = sr::dev_alloc(&r->s0, 9 * m)) || (rc = sr::dev_alloc(&r->sf, 9 * m)) || (rc = sr::dev_alloc(&r->rf, 4 * m)) ||
      (rc = sr::dev_alloc(&r->Jf, 4 * m)) || (rc = sr::dev_alloc(&r->perm, m)) || (rc = sr::dev_alloc(&r->keys, m)) ||
      (rc = sr::dev_alloc(&r->fb_list, m)) || (rc = sr::dev_alloc(&r->counters, sr::kCounterWords)) ||
      (rc = sr::dev_alloc(&r->guard, m))) {
    sr_rays_destroy(r);
    return rc;
  }
  *out = r;
  return SR_OK;
}

int64_t sr_rays_count(const sr_rays *r) { return r ? r->n : 0; }

int sr_rays_upload(sr_rays *r, const double *s0) {
  SR_CHECK(r && s0, "sr_rays_upload: NULL argument");
  r->have_bbox = r->bbox_given = false;
  if (r->n > 0) {
    hipStream_t st = sr::ctx().stream;
    int rc = sr::upload_sync(r->s0, s0, sizeof(double) * 9 * (size_t)r->n, st);
    if (rc) return rc;
    // the launch positions' bounding box (counters [10..15]: free between traces), read back with the wait the copy needs anyway
    unsigned long long *box = r->counters + 10, hb[6];
    SR_HIP(hipMemsetAsync(box, 0xff, 3 * sizeof(unsigned long long), st));
    SR_HIP(hipMemsetAsync(box + 3, 0, 3 * sizeof(unsigned long long), st));
    const unsigned grid = (unsigned)std::min<int64_t>(sr::grid_for(r->n, 256), (int64_t)sr::ctx().n_cu * 8);
    hipLaunchKernelGGL(k_bbox, dim3(grid), dim3(256), 0, st, (const double *)r->s0, r->n, box);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(hb, box, sizeof hb, hipMemcpyDeviceToHost, st));
    SR_HIP(hipStreamSynchronize(st));
    bool any = true;
    for (int q = 0; q < 3; ++q) any = any && hb[q] <= hb[3 + q];
    if (any) {
      for (int q = 0; q < 6; ++q) r->bbox[q] = from_ordered_bits(hb[q]);
      r->have_bbox = true;
    }
  }
  r->have_s0 = true;
  r->traced = false;
  return SR_OK;
}

int sr_rays_upload_part(sr_rays *r, const double *s0, int64_t n, int64_t first, int last) {
  SR_CHECK(r && (s0 || n == 0), "sr_rays_upload_part: NULL argument");
  SR_CHECK(n >= 0 && first >= 0 && first + n <= r->n, "sr_rays_upload_part: rays %lld .. %lld of a bundle of %lld", (long long)first,
           (long long)(first + n), (long long)r->n);
  hipStream_t st = sr::ctx().stream;
  r->have_bbox = r->bbox_given = false;
  r->have_s0 = false;
  r->traced = false;
  if (n > 0) {
    SR_HIP(hipMemcpy2DAsync(r->s0 + first, sizeof(double) * (size_t)r->n, s0, sizeof(double) * (size_t)n, sizeof(double) * (size_t)n, 9,
                            hipMemcpyHostToDevice, st));
    SR_HIP(hipStreamSynchronize(st));  // the caller's block may be drawn into again
  }
  if (!last) return SR_OK;
  if (r->n > 0) {  // as sr_rays_upload: the box of the whole bundle
    unsigned long long *box = r->counters + 10, hb[6];
    SR_HIP(hipMemsetAsync(box, 0xff, 3 * sizeof(unsigned long long), st));
    SR_HIP(hipMemsetAsync(box + 3, 0, 3 * sizeof(unsigned long long), st));
    const unsigned grid = (unsigned)std::min<int64_t>(sr::grid_for(r->n, 256), (int64_t)sr::ctx().n_cu * 8);
    hipLaunchKernelGGL(k_bbox, dim3(grid), dim3(256), 0, st, (const double *)r->s0, r->n, box);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(hb, box, sizeof hb, hipMemcpyDeviceToHost, st));
    SR_HIP(hipStreamSynchronize(st));
    bool any = true;
    for (int q = 0; q < 3; ++q) any = any && hb[q] <= hb[3 + q];
    if (any) {
      for (int q = 0; q < 6; ++q) r->bbox[q] = from_ordered_bits(hb[q]);
      r->have_bbox = true;
    }
  }
  r->have_s0 = true;
  return SR_OK;
}

int sr_rays_trace(sr_rays *r, const sr_volume *v, const sr_trace_params *p, sr_trace_stats *stats) {
  SR_CHECK(r && v && p, "sr_rays_trace: NULL argument");
  SR_CHECK((p->handoff & ~(SR_HANDOFF_ENTER | SR_HANDOFF_EXIT)) == 0, "handoff must be a combination of SR_HANDOFF_*");
  const bool ho_enter = (p->handoff & SR_HANDOFF_ENTER) != 0;
  if (!ho_enter && !r->have_s0) return sr::fail(SR_ERR_STATE, "sr_rays_trace: no rays uploaded");
  if (ho_enter && !r->have_rec) return sr::fail(SR_ERR_STATE, "sr_rays_trace: SR_HANDOFF_ENTER without hand-off records");

  SR_CHECK(ho_enter || v->k_lo == 0, "rays can only start (no SR_HANDOFF_ENTER) on the slab that holds node plane 0");
  SR_CHECK((p->handoff & SR_HANDOFF_EXIT) || v->k_hi == v->n_glob - 1,
           "rays can only finish (no SR_HANDOFF_EXIT) on the slab that holds the last node plane");
  SR_CHECK(p->probing_axis == v->axis, "probing_axis %d does not match the volume's layout axis %d", p->probing_axis, v->axis);
  SR_CHECK(p->substeps >= 1 && p->substeps <= 64, "substeps must be in 1..64, got %d", p->substeps);
  SR_CHECK(p->t_end > 0, "t_end must be positive");
  SR_CHECK(p->row_order == SR_ROWS_LEGACY || p->row_order == SR_ROWS_JAX, "row_order must be SR_ROWS_LEGACY or SR_ROWS_JAX");
  SR_CHECK(p->precision == SR_PREC_F64 || p->precision == SR_PREC_MIXED, "precision must be SR_PREC_F64 or SR_PREC_MIXED");
  sr::Context &c = sr::ctx();
  hipStream_t st = c.stream;
  const int64_t N = r->n;
  if (stats) *stats = sr_trace_stats{0, 0, 0.0, 0.0};
  if (N == 0) {
    r->traced = true;
    return SR_OK;
  }
  const int block = 256;
  const unsigned nblk = sr::grid_for(N, block);
  VolDev V = vol_dev(v);
  r->tile_segs = r->tile_segs_run = 0;
  r->tile_rec = false;
  TilePlan tplan;
  const bool tiled = tile_plan(r, v, p, N, tplan, st);
  const TileGeom &tile_geom = tplan.g;

  SR_HIP(hipEventRecord(c.ev[0], st));
  if (r->counters_carry)  // totals of earlier calls are still unread: only this call's queue lengths start at zero
    SR_HIP(hipMemsetAsync(r->counters + 1, 0, 2 * sizeof(unsigned long long), st));
  else
    SR_HIP(hipMemsetAsync(r->counters, 0, sr::kCounterWords * sizeof(unsigned long long), st));
  if (p->handoff && !r->rec) {
    int rc = sr::dev_alloc(&r->rec, (size_t)10 * (size_t)std::max(r->cap, N));
    if (rc) return rc;
  }
  if (ho_enter && tiled) {
    // trace_tiled bins the arrivals by the cell they are in NOW, as between two segments (records gathered, perm from row 9)
  } else if (ho_enter) {  // arrival order is the sender's launch order (already binned); the ray index rides in row 9
    hipLaunchKernelGGL(k_perm_from_rec, dim3(nblk), dim3(block), 0, st, (const double *)r->rec, N, r->perm);
  } else if (p->sort_rays && tiled) {  // the band order of trace_tile.inc
    int rc = bin_by_band(r, v, tile_geom, nullptr, r->perm, st);
    if (rc) return rc;
  } else if (p->sort_rays) {
    int bits = 1;
    while ((1 << bits) < std::max(v->nb - 1, v->nc - 1)) ++bits;
    SR_CHECK(bits <= 15, "lateral grid too large for the 32-bit Morton ray key");
    SR_CHECK(bits <= 11, "lateral grid too large for the LDS counters of the ray binning (2048 x 2048 cells)");
    const uint32_t oob_key = (uint32_t)1 << (2 * bits);  // out-of-volume / NaN rays: one more coarse group, after all cells
    hipLaunchKernelGGL(k_keys, dim3(nblk), dim3(block), 0, st, V, (const double *)r->s0, N, v->axis, oob_key, r->keys);
    int rc = bin_rays(r, N, bits, (1 << bits) + 1, r->perm, st);  // 2*bits key bits: the coarser Morton cell, then the cell inside it
    if (rc) return rc;
  } else {
    hipLaunchKernelGGL(k_iota, dim3(nblk), dim3(block), 0, st, r->perm, N);
  }

  TraceArgs A;
  {
    int rc = make_trace_args(r, v, p, A);
    if (rc) return rc;
  }
  SR_HIP(hipEventRecord(c.ev[1], st));
  const bool aux = v->K != nullptr || v->Q != nullptr;  // amp / pol terms (A7)
  // Levels: [mixed kernel ->] float64 plane kernel -> time-stepping form.  Each level takes the launch slots the one
  // before it queued (device-side counts, no host round trip) and queues what it cannot finish itself.
  // The mixed build is k_trace_mx: one step per cell, no optional terms -- the common case, with the per-ray error bound the
  // exact-counts deposit works from.  SR_PREC_MIXED with sub-steps or with kappa / Faraday fields runs the FLOAT64 kernels
  // (k_trace_f64<., AUX, SUBS>): round 1's float32 kernel for those cases kept no error bound, so an exact-counts deposit had to
  // trace every ray again in float64 anyway, and it spilled registers; the float64 result needs no guard (bound 0).
  const bool mixed = p->precision == SR_PREC_MIXED && !aux && p->substeps == 1;
  if (mixed) {
    launch_mx(v, A, st);
    SR_HIP(hipEventRecord(c.ev[2], st));
    // second level: the queue of the mixed kernel; its own rejects go to a second list (the sort keys' buffer,
    // free once the permutation exists)
    A.in_list = r->fb_list;
    A.in_count = r->counters + 1;
    A.out_list = r->keys;
    A.out_count = r->counters + 2;
    launch_planes64(v, p, A, st);
  } else if (tiled) {
    // the tile kernel over every ray, in segments of node planes; what it loses (rays leaving their workgroup's tile or the
    // volume, rays that are not plane-form rays) is queued for k_trace_f64, from s0
    int rc = trace_tiled(r, v, p, tplan, A, st);
    if (rc) return rc;
    A.rec = r->rec;  // the record buffers may have changed places
    A.in_list = r->fb_list;
    A.in_count = r->counters + 1;
    A.out_list = r->keys;
    A.out_count = r->counters + 2;
    launch_planes64(v, p, A, st);
    SR_HIP(hipEventRecord(c.ev[2], st));
  } else {
    launch_planes64(v, p, A, st);
    SR_HIP(hipEventRecord(c.ev[2], st));
  }
  // Not on a slab, which holds only its own planes (the plane kernel has written NaN for such rays).
  if (!p->handoff) launch_time(v, A, st);
  hipLaunchKernelGGL(k_carry, dim3(1), dim3(1), 0, st, r->counters);
  SR_HIP(hipGetLastError());
  SR_HIP(hipEventRecord(c.ev[3], st));
  r->counters_carry = true;
  r->traced = (p->handoff & SR_HANDOFF_EXIT) == 0;
  r->have_rec = (p->handoff & SR_HANDOFF_EXIT) != 0;
  r->sorted = p->sort_rays != 0 || ho_enter;
  // what an exact-counts deposit needs to refine this trace (deposit.hip): only a mixed-precision trace of a WHOLE volume
  // from s0 can be repeated in float64 (a slab holds neither the other planes nor the rays' start)
  r->last_vol = v;
  r->last_p = *p;
  r->guard_live = mixed && !p->handoff;
  // Position bound of rf = guard_len * angle bound.  rf is the BACK-PROJECTION of the final state onto the plane `extent` of the
  // probing axis (project(); full_solver.py:856-881), and the flight from the last node plane to t_end and back is one exact
  // straight line, so t_end does not enter: the lever is the volume's length (the bound on the position ON the last node plane,
  // trace_mx.inc) plus the distance from that plane to the plane the rays are projected onto (0 to rounding for the reference's
  // extent = the grid's half-length, but a caller may ask for any plane).
  r->guard_len = v->hg[0].empty() ? 0.0 : (v->hg[0].back() - v->hg[0].front()) + std::fabs(p->extent - v->hg[0].back());
  if (stats) return sr_rays_trace_stats(r, stats);
  return SR_OK;
}

int sr_rays_trace_stats(sr_rays *r, sr_trace_stats *stats) {
  SR_CHECK(r != nullptr && stats != nullptr, "sr_rays_trace_stats: NULL argument");
  *stats = sr_trace_stats{0, 0, 0.0, 0.0};
  if (!r->counters_carry) return SR_OK;  // nothing traced since the counters were last read
  sr::Context &c = sr::ctx();
  hipStream_t st = c.stream;
  std::vector<unsigned long long> h(sr::kCounterWords, 0ull);
  SR_HIP(hipMemcpyAsync(h.data(), r->counters, sizeof(unsigned long long) * sr::kCounterWords, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  float ms = 0.f;  // the library's events belong to the last trace on the stream
  if (hipEventElapsedTime(&ms, c.ev[1], c.ev[2]) == hipSuccess) stats->trace_kernel_ms = ms;
  if (r->tile_segs > 0) {  // the tile path: its kernels alone (the span above also holds the binning between segments and the recovery)
    double sum = 0;
    bool ok = true;
    for (int q = 0; q < r->tile_segs; ++q) {
      ok = ok && hipEventElapsedTime(&ms, c.ev[4 + 2 * q], c.ev[5 + 2 * q]) == hipSuccess;
      sum += ms;
    }
    if (ok) stats->trace_kernel_ms = sum;
  }
  if (hipEventElapsedTime(&ms, c.ev[0], c.ev[3]) == hipSuccess) stats->total_ms = ms;
  stats->ray_steps = (int64_t)sr::stripe_sum(h.data(), 0);
  stats->fallback_rays = (int64_t)h[3];
  r->counters_carry = false;
  return SR_OK;
}

int sr_rays_tile_segments(const sr_rays *r) { return r ? r->tile_segs_run : 0; }

int sr_rays_tile_records(const sr_rays *r) { return r && r->tile_rec ? 1 : 0; }

double sr_tile_min_density(void) { return kTileMinDensity; }

int sr_rays_download(const sr_rays *r, double *sf, double *rf, double *Jf) {
  SR_CHECK(r != nullptr, "sr_rays_download: NULL rays");
  return download_rows(r, sf, rf, Jf, r->n, 0, nullptr);
}

__global__ void k_unpermute_f32(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ perm, int64_t N) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j < N) dst[perm[j]] = src[j];
}

int sr_rays_error_bound(const sr_rays *r, float *bound) {
  SR_CHECK(r && bound, "sr_rays_error_bound: NULL argument");
  if (!r->traced) return sr::fail(SR_ERR_STATE, "sr_rays_error_bound: rays have not been traced");
  const int64_t N = r->n;
  if (N == 0) return SR_OK;
  if (!r->guard_live) {  // float64 build (or a slab): every ray is as exact as the library gets
    std::fill(bound, bound + N, 0.f);
    return SR_OK;
  }
  hipStream_t st = sr::ctx().stream;
  float *tmp = nullptr;
  int rc = sr::dev_alloc(&tmp, (size_t)N);
  if (rc) return rc;
  hipLaunchKernelGGL(k_unpermute_f32, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, (const float *)r->guard, tmp, (const uint32_t *)r->perm, N);
  hipError_t e = hipMemcpyAsync(bound, tmp, sizeof(float) * (size_t)N, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  sr::dev_free(tmp);
  if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_rays_error_bound: %s", hipGetErrorString(e));
  return SR_OK;
}

int sr_rays_download_s0(const sr_rays *r, double *s0) {
  SR_CHECK(r && s0, "sr_rays_download_s0: NULL argument");
  if (!r->have_s0) return sr::fail(SR_ERR_STATE, "sr_rays_download_s0: no rays uploaded or generated");
  if (r->n == 0) return SR_OK;
  hipStream_t st = sr::ctx().stream;
  SR_HIP(hipMemcpyAsync(s0, r->s0, sizeof(double) * 9 * (size_t)r->n, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int sr_rays_handoff_download(const sr_rays *r, double *rec) {
  SR_CHECK(r && rec, "sr_rays_handoff_download: NULL argument");
  if (!r->have_rec) return sr::fail(SR_ERR_STATE, "sr_rays_handoff_download: no hand-off records (trace with SR_HANDOFF_EXIT first)");
  if (r->n == 0) return SR_OK;
  hipStream_t st = sr::ctx().stream;
  SR_HIP(hipMemcpyAsync(rec, r->rec, sizeof(double) * 10 * (size_t)r->n, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int sr_rays_handoff_upload(sr_rays *r, const double *rec) {
  SR_CHECK(r && rec, "sr_rays_handoff_upload: NULL argument");
  if (r->n > 0) {
    if (!r->rec) {
      int rc = sr::dev_alloc(&r->rec, (size_t)10 * (size_t)std::max(r->cap, r->n));
      if (rc) return rc;
    }
    hipStream_t st = sr::ctx().stream;
    SR_HIP(hipMemcpyAsync(r->rec, rec, sizeof(double) * 10 * (size_t)r->n, hipMemcpyHostToDevice, st));
    SR_HIP(hipStreamSynchronize(st));
  }
  r->have_rec = true;
  r->traced = false;
  if (!r->bbox_given) r->have_bbox = false;  // other rays than the bundle's last upload: judged by the whole lateral grid, unless the caller named their beam
  return SR_OK;
}

int sr_rays_set_bbox(sr_rays *r, const double *bbox) {
  SR_CHECK(r != nullptr, "sr_rays_set_bbox: NULL rays");
  if (!bbox) {
    r->have_bbox = r->bbox_given = false;
    return SR_OK;
  }
  for (int q = 0; q < 3; ++q) SR_CHECK(bbox[q] <= bbox[3 + q], "sr_rays_set_bbox: min > max (or NaN) on axis %d", q);
  for (int q = 0; q < 6; ++q) r->bbox[q] = bbox[q];
  r->have_bbox = r->bbox_given = true;
  return SR_OK;
}

int sr_rays_get_bbox(const sr_rays *r, double *bbox, int *known) {
  SR_CHECK(r && bbox && known, "sr_rays_get_bbox: NULL argument");
  *known = r->have_bbox ? 1 : 0;
  for (int q = 0; q < 6; ++q) bbox[q] = r->have_bbox ? r->bbox[q] : 0.0;
  return SR_OK;
}

int sr_volume_sample(const sr_volume *v, const double *pts, int64_t N, double *out) {
  SR_CHECK(v && N >= 0 && (N == 0 || (pts && out)), "sr_volume_sample: bad argument");
  if (N == 0) return SR_OK;
  hipStream_t st = sr::ctx().stream;
  double *d = nullptr;
  int rc = sr::dev_alloc(&d, (size_t)7 * N);
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync(d, pts, sizeof(double) * 3 * N, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_sample, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, vol_dev(v), v->axis, v->L != nullptr,
                       (const double *)d, N, d + 3 * N);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d + 3 * N, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  sr::dev_free(d);
  if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_volume_sample: %s", hipGetErrorString(e));
  return SR_OK;
}

int sr_volume_sample_aux(const sr_volume *v, const double *pts, int64_t N, double *out) {
  SR_CHECK(v && N >= 0 && (N == 0 || (pts && out)), "sr_volume_sample_aux: bad argument");
  if (N == 0) return SR_OK;
  hipStream_t st = sr::ctx().stream;
  double *d = nullptr;
  int rc = sr::dev_alloc(&d, (size_t)8 * N);
  if (rc) return rc;
  hipError_t e = hipMemcpyAsync(d, pts, sizeof(double) * 3 * N, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_sample_aux, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, vol_dev(v), v->axis, (const double *)d, N,
                       d + 3 * N);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(out, d + 3 * N, sizeof(double) * 5 * N, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  sr::dev_free(d);
  if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_volume_sample_aux: %s", hipGetErrorString(e));
  return SR_OK;
}

int sr_ray_to_jones(const double *sf, int64_t N, double extent, int probing_axis, int row_order, double *rf, double *Jf) {
  SR_CHECK(N >= 0 && (N == 0 || (sf && rf)), "sr_ray_to_jones: bad argument");
  SR_CHECK(probing_axis >= 0 && probing_axis <= 2, "probing_axis must be 0, 1 or 2, got %d", probing_axis);
  if (N == 0) return SR_OK;
  int rc = sr::ensure_init();
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;
  double *d = nullptr;
  if ((rc = sr::dev_alloc(&d, (size_t)(9 + 4 + 4) * N))) return rc;
  double *drf = d + 9 * N, *dJf = drf + 4 * N;
  hipError_t e = hipMemcpyAsync(d, sf, sizeof(double) * 9 * N, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_ray_to_jones, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, (const double *)d, N, extent,
                       probing_axis, row_order, drf, Jf ? dJf : (double *)nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(rf, drf, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess && Jf) e = hipMemcpyAsync(Jf, dJf, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  sr::dev_free(d);
  if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_ray_to_jones: %s", hipGetErrorString(e));
  return SR_OK;
}

// The host-buffer entry point on a large bundle: the rays go through in chunks that alternate between the library's two
// streams, so that the upload of chunk i+1 and the download of chunk i-1 (host-synchronous copies of pageable memory)
// run while chunk i is traced (the traces themselves one after the other: see `serial` below).  Rays are independent and every output row is written at its own rays' columns: the
// arrays are those of the single pass, bit for bit.  No hipMalloc / hipFree inside the loop after the first two chunks
// (hipFree waits for every stream).
static int64_t pipeline_chunk() {
  const char *e = getenv("SYNTHRAY_TRACE_CHUNK");  // most rays per chunk; 0 = never pipeline
  // 2.5 * 2^20: 1e7 rays in four chunks of 2.5e6, dense enough for the tile path's records kernel (15 rays per cell of a 4 mm
  // beam on 512^3) -- 77.7 ms per call with page-locked result arrays; 2^20: 79.5, 1.5 * 2^20: 84 (chunks at the tile path's
  // threshold, where it is no faster than the per-ray kernel), 2^21: 80, 3.4e6: 77.7, 5e6: 80.5 (profiles/r05_pcie_host_arrays.txt)
  return e ? atoll(e) : (int64_t)5 << 19;
}

// Result arrays that are ordinary (pageable, never written) NumPy memory cost a page fault per 4 KB when the copy engine's
// staging thread first writes them: 1.36 GB of sf / rf / Jf for 1e7 rays, more time than the trace.  MADV_POPULATE_WRITE
// maps the pages WITHOUT touching their contents (safe beside copies already landing), and several threads do it side by
// side while the first chunks are uploaded and traced.  Page-locked arrays (sr_host_alloc) are left alone.
static void populate_pages(double *p, size_t bytes, std::vector<std::thread> &pool) {
  if (!p || bytes < ((size_t)8 << 20)) return;
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type != hipMemoryTypeUnregistered) return;  // page-locked already
  (void)hipGetLastError();
  const uintptr_t page = 4096, lo = ((uintptr_t)p + page - 1) & ~(page - 1), hi = ((uintptr_t)p + bytes) & ~(page - 1);
  if (hi <= lo) return;
  const int n_thr = 4;
  const uintptr_t per = (((hi - lo) / n_thr) + page - 1) & ~(page - 1);
  for (int t = 0; t < n_thr; ++t) {
    const uintptr_t a = lo + (uintptr_t)t * per, b = std::min(hi, a + per);
    if (a < b) pool.emplace_back([a, b]() { (void)madvise((void *)a, b - a, 23 /* MADV_POPULATE_WRITE */); });
  }
}

// The device side of trace_pipelined -- three chunk-sized ray bundles and two staging blocks, ~2.5 GB of HBM at the default
// chunk -- is KEPT between calls (a loop of solve() calls: ~40 hipMalloc + ~40 hipFree, each of which waits for the device,
// were 6 of a call's 85 ms).  Released by sr_release_caches(), by a call with another chunk size or device, and not kept at
// all with SYNTHRAY_TRACE_CACHE=0.
namespace {
struct PipelineCache {
  int64_t chunk = 0;
  int device = -1;
  sr_rays *ring[3] = {nullptr, nullptr, nullptr};
  double *staging[2] = {nullptr, nullptr};
} g_pipe;
void release_pipeline_cache() {
  for (auto &r : g_pipe.ring) {
    if (r) sr_rays_destroy(r);
    r = nullptr;
  }
  for (auto &q : g_pipe.staging) {
    sr::dev_free(q);
    q = nullptr;
  }
  g_pipe.chunk = 0;
  g_pipe.device = -1;
}
}  // namespace

int sr_release_caches(void) {
  if (sr::ctx().stream) (void)sr_synchronize();
  release_pipeline_cache();
  sr::scratch_release();
  return SR_OK;
}

static int trace_pipelined(const sr_volume *v, const double *s0, int64_t N, const sr_trace_params *p, double *sf, double *rf,
                           double *Jf, sr_trace_stats *stats, int64_t cap) {
  sr::Context &c = sr::ctx();
  const int saved = c.current;
  // `cap`: the most rays of a chunk (what the bundles, staging blocks and bounce buffers are sized for, and what the cache is
  // kept by); the chunks themselves are EQUAL parts of this call's rays -- 1e7 rays: 4 x 2.5e6, not 3 x 2.62e6 and a rest at a
  // lower ray density (the density chooses the kernel)
  const int64_t n_chunks = std::max<int64_t>(2, (N + cap - 1) / cap);
  int64_t chunk = std::min(cap, (N + n_chunks - 1) / n_chunks);
  if ((n_chunks - 1) * chunk >= N) chunk = cap;  // (only chunks of a few rays: n_chunks^2 > N)
  const int64_t last = N - (n_chunks - 1) * chunk;
  constexpr int kRing = 3;  // bundles in flight: one being traced on each of the two streams, one being uploaded
  sr_rays *ring[kRing] = {nullptr, nullptr, nullptr};
  double *staging[2] = {nullptr, nullptr};
  sr_trace_stats tot{0, 0, 0.0, 0.0};
  int rc = SR_OK;
  std::vector<std::thread> faulters;
  populate_pages(sf, sizeof(double) * 9 * (size_t)N, faulters);
  populate_pages(rf, sizeof(double) * 4 * (size_t)N, faulters);
  populate_pages(Jf, sizeof(double) * 4 * (size_t)N, faulters);
  const char *ce = getenv("SYNTHRAY_TRACE_CACHE");
  const bool keep = !(ce && ce[0] == '0');
  if (g_pipe.chunk != cap || g_pipe.device != c.device) release_pipeline_cache();
  for (int q = 0; q < kRing; ++q) {  // from the cache (whole set or nothing)
    ring[q] = g_pipe.ring[q];
    g_pipe.ring[q] = nullptr;
  }
  for (int q = 0; q < 2; ++q) {
    staging[q] = g_pipe.staging[q];
    g_pipe.staging[q] = nullptr;
  }
  g_pipe.chunk = 0;
  for (int q = 0; q < kRing && q < n_chunks && !rc; ++q)
    if (!ring[q]) rc = sr_rays_create(&ring[q], cap);  // a chunk may use part of one
  for (int q = 0; q < 2 && !rc; ++q) {
    rc = sr_stream_select(q);
    if (!rc && !staging[q]) rc = sr::dev_alloc(&staging[q], (size_t)17 * (size_t)cap);
  }
  // events: uploaded[ci] (recorded by the uploader on its own stream), traced[ci] (recorded after chunk ci's trace)
  std::vector<hipEvent_t> uploaded((size_t)n_chunks, nullptr), traced((size_t)n_chunks, nullptr);
  // per chunk: the launch positions' bounding box (min x, y, z, max x, y, z), written by the uploader's copier threads before the
  // chunk is announced (n_uploaded, under the mutex); lo > hi: not known
  std::vector<std::array<double, 6>> boxes((size_t)n_chunks, std::array<double, 6>{1, 1, 1, 0, 0, 0});
  for (int64_t ci = 0; ci < n_chunks && !rc; ++ci) {
    if (hipEventCreateWithFlags(&uploaded[ci], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&traced[ci], hipEventDisableTiming) != hipSuccess)
      rc = sr::fail(SR_ERR_HIP, "sr_trace: hipEventCreate failed");
  }
  // ---- the uploader: a host thread of its own, because a copy FROM pageable memory holds the thread that asked for it
  // (staged through the runtime's bounce buffers at ~15 GB/s): on the caller's thread every chunk's upload delayed the
  // download of the chunk before it and the launch of the chunk after it.
  std::mutex mu;
  std::condition_variable cv;
  int64_t n_uploaded = 0, n_traced = 0;  // chunks whose `uploaded` / `traced` event has been RECORDED (guarded by mu)
  bool abort_upload = false;
  hipError_t up_err = hipSuccess;
  const int device = c.device;
  // Page-locked bounce buffers of the library's own (two chunks' worth, allocated at the first large sr_trace and kept): the
  // runtime's copy from pageable memory runs on ONE thread at ~8 GB/s -- 90 ms for the 0.72 GB of 1e7 rays, more than their
  // trace; four threads copying into a page-locked buffer and a DMA from there move them in a quarter of that.
  static double *bounce[2] = {nullptr, nullptr};
  static size_t bounce_rays = 0;
  if (!rc && bounce_rays < (size_t)cap) {
    for (auto &b : bounce) {
      if (b) (void)hipHostFree(b);
      b = nullptr;
    }
    bounce_rays = 0;
    if (hipHostMalloc(reinterpret_cast<void **>(&bounce[0]), sizeof(double) * 9 * (size_t)cap, hipHostMallocDefault) == hipSuccess &&
        hipHostMalloc(reinterpret_cast<void **>(&bounce[1]), sizeof(double) * 9 * (size_t)cap, hipHostMallocDefault) == hipSuccess)
      bounce_rays = (size_t)cap;
    else
      (void)hipGetLastError();  // no page-locked memory to be had: the runtime's own staging does (slower)
  }
  const bool use_bounce = bounce_rays >= (size_t)cap;
  std::thread uploader;
  if (!rc) uploader = std::thread([&]() {
    hipStream_t us = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&us, hipStreamNonBlocking);
    hipEvent_t bounce_free[2] = {nullptr, nullptr};
    for (auto &b : bounce_free)
      if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
    for (int64_t ci = 0; ci < n_chunks && e == hipSuccess; ++ci) {
      if (ci >= kRing) {  // the bundle is free again once the trace that read it is done
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return abort_upload || n_traced > ci - kRing; });
          if (abort_upload) break;
        }
        e = hipEventSynchronize(traced[ci - kRing]);
        if (e != hipSuccess) break;
      }
      sr_rays *r = ring[ci % kRing];
      const int64_t off = ci * chunk, n = ci + 1 < n_chunks ? chunk : last;
      if (use_bounce) {
        double *bb = bounce[ci & 1];
        if (ci >= 2) e = hipEventSynchronize(bounce_free[ci & 1]);  // the DMA that last read this buffer
        std::thread copiers[3];
        // the three position rows are read for their bounding box while they are copied (the chunk's rays per lateral cell of
        // the BEAM choose its kernel, as for a bundle uploaded whole: tile_plan); NaN positions compare false and are left out
        auto copy_rows = [&](int q0, int q1) {
          for (int q = q0; q < q1; ++q) {
            const double *src = s0 + (size_t)q * N + off;
            double *dst = bb + (size_t)q * n;
            if (q < 3) {
              double lo = __builtin_inf(), hi = -__builtin_inf();
              for (int64_t t = 0; t < n; ++t) {
                const double x = src[t];
                dst[t] = x;
                lo = x < lo ? x : lo;
                hi = x > hi ? x : hi;
              }
              boxes[(size_t)ci][q] = lo;
              boxes[(size_t)ci][3 + q] = hi;
            } else {
              memcpy(dst, src, sizeof(double) * (size_t)n);
            }
          }
        };
        copiers[0] = std::thread(copy_rows, 2, 4);
        copiers[1] = std::thread(copy_rows, 4, 6);
        copiers[2] = std::thread(copy_rows, 6, 9);
        copy_rows(0, 2);
        for (auto &t : copiers) t.join();
        if (e == hipSuccess) e = hipMemcpyAsync(r->s0, bb, sizeof(double) * 9 * (size_t)n, hipMemcpyHostToDevice, us);  // rows at pitch n
        if (e == hipSuccess) e = hipEventRecord(bounce_free[ci & 1], us);
      } else {
        for (int q = 0; q < 9 && e == hipSuccess; ++q)  // rows of n rays at pitch n: a shorter last chunk uses the front of a full-size bundle
          e = hipMemcpyAsync(r->s0 + (size_t)q * n, s0 + (size_t)q * N + off, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, us);
      }
      if (e == hipSuccess) e = hipEventRecord(uploaded[ci], us);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (e == hipSuccess) n_uploaded = ci + 1;
      }
      cv.notify_all();
    }
    if (us) {
      (void)hipStreamSynchronize(us);
      (void)hipStreamDestroy(us);
    }
    for (auto &b : bounce_free)
      if (b) (void)hipEventDestroy(b);
    {
      std::lock_guard<std::mutex> lk(mu);
      up_err = e;
      if (e != hipSuccess) abort_upload = true;
    }
    cv.notify_all();
  });
  struct Pending {
    sr_rays *r;
    int64_t off, n;
    int sid;
  } prev{nullptr, 0, 0, 0};
  auto finish = [&](const Pending &q) -> int {  // waits for the chunk's trace (its stream), copies its rows out, adds its totals
    int e = sr_stream_select(q.sid);
    if (!e) e = download_rows(q.r, sf, rf, Jf, N, q.off, staging[q.sid]);
    sr_trace_stats st{0, 0, 0.0, 0.0};
    if (!e) e = sr_rays_trace_stats(q.r, &st);
    tot.ray_steps += st.ray_steps;
    tot.fallback_rays += st.fallback_rays;
    tot.trace_kernel_ms += st.trace_kernel_ms;
    tot.total_ms += st.total_ms;
    return e;
  };
  const bool dbg = getenv("SYNTHRAY_TRACE_DEBUG") != nullptr;
  // The chunks' TRACES run one after the other, each behind the one before (an event wait; the streams still alternate, so
  // the download of chunk i runs beside the trace of chunk i+1): two traces side by side finish together, and the first one's
  // download then overlaps nothing -- 104 -> 91 ms per 1e7 rays at 2^21-ray chunks, 119 -> 104 at 5e6.  SYNTHRAY_TRACE_SERIAL=0:
  // side by side, as before.
  const bool serial = getenv("SYNTHRAY_TRACE_SERIAL") ? atoi(getenv("SYNTHRAY_TRACE_SERIAL")) != 0 : true;
  auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now_ms();
  for (int64_t ci = 0; ci < n_chunks && !rc; ++ci) {
    const int sid = (int)(ci & 1);
    const int64_t off = ci * chunk, n = ci + 1 < n_chunks ? chunk : last;
    rc = sr_stream_select(sid);
    if (rc) break;
    const double t_a = now_ms();
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return abort_upload || n_uploaded > ci; });
      if (abort_upload) {
        rc = sr::fail(SR_ERR_HIP, "sr_trace: upload of rays %lld..: %s", (long long)off, hipGetErrorString(up_err));
        break;
      }
    }
    sr_rays *r = ring[ci % kRing];
    if (hipStreamWaitEvent(c.stream, uploaded[ci], 0) != hipSuccess ||
        (serial && ci > 0 && hipStreamWaitEvent(c.stream, traced[ci - 1], 0) != hipSuccess)) {
      rc = sr::fail(SR_ERR_HIP, "sr_trace: hipStreamWaitEvent failed");
      break;
    }
    r->n = n;  // a shorter last chunk: the front of a full-size bundle (its rows were uploaded at pitch n)
    r->have_bbox = true;   // found by the uploader's copier threads while they copied the position rows
    for (int q = 0; q < 6; ++q) r->bbox[q] = boxes[(size_t)ci][q];
    for (int q = 0; q < 3; ++q)
      if (!(r->bbox[q] <= r->bbox[3 + q])) r->have_bbox = false;  // no bounce buffers (the runtime staged the rows), or every position NaN
    r->have_s0 = true;
    r->traced = false;
    rc = sr_rays_trace(r, v, p, nullptr);  // queued; returns at once
    if (!rc && hipEventRecord(traced[ci], c.stream) != hipSuccess) rc = sr::fail(SR_ERR_HIP, "sr_trace: hipEventRecord failed");
    {
      std::lock_guard<std::mutex> lk(mu);
      n_traced = ci + 1;
    }
    cv.notify_all();
    if (rc) break;
    const double t_b = now_ms();
    if (prev.r) rc = finish(prev);  // the chunk before this one, on the other stream
    if (dbg) fprintf(stderr, "sr_trace chunk %lld: at %.1f ms waited %.1f ms for its upload, queued in %.1f ms, finish(prev) %.1f ms\n", (long long)ci,
                     t_a - t_begin, t_b - t_a, 0.0, now_ms() - t_b);
    prev = Pending{r, off, n, sid};
  }
  {
    const double t_b = now_ms();
    if (!rc && prev.r) rc = finish(prev);
    if (dbg) fprintf(stderr, "sr_trace last finish at %.1f ms: %.1f ms\n", t_b - t_begin, now_ms() - t_b);
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    if (rc) abort_upload = true;
  }
  cv.notify_all();
  if (uploader.joinable()) uploader.join();
  for (auto &t : faulters) t.join();
  (void)sr_synchronize();
  for (int q = 0; q < kRing; ++q)
    if (ring[q]) ring[q]->n = cap;
  if (keep && !rc) {  // for the next call
    for (int q = 0; q < kRing; ++q) g_pipe.ring[q] = ring[q];
    for (int q = 0; q < 2; ++q) g_pipe.staging[q] = staging[q];
    g_pipe.chunk = cap;
    g_pipe.device = c.device;
  } else {
    for (int q = 0; q < kRing; ++q) sr_rays_destroy(ring[q]);
    for (int q = 0; q < 2; ++q) sr::dev_free(staging[q]);
  }
  for (auto e : uploaded)
    if (e) (void)hipEventDestroy(e);
  for (auto e : traced)
    if (e) (void)hipEventDestroy(e);
  (void)sr_stream_select(saved);
  if (stats) *stats = tot;
  return rc;
}

int sr_trace(const sr_volume *v, const double *s0, int64_t n_rays, const sr_trace_params *p, double *sf, double *rf,
             double *Jf, sr_trace_stats *stats) {
  SR_CHECK(v && s0 && p, "sr_trace: NULL argument");
  const int64_t chunk = pipeline_chunk();
  // from 1.2 chunks' worth of rays: two equal chunks (3.1e6 rays and more at the default, as before the chunks grew)
  if (chunk > 0 && n_rays >= chunk + chunk / 5 && n_rays >= 2 && !p->handoff) return trace_pipelined(v, s0, n_rays, p, sf, rf, Jf, stats, chunk);
  sr_rays *r = nullptr;
  int rc = sr_rays_create(&r, n_rays);
  if (rc) return rc;
  rc = sr_rays_upload(r, s0);
  if (!rc) rc = sr_rays_trace(r, v, p, stats);
  if (!rc) rc = sr_rays_download(r, sf, rf, Jf);
  sr_rays_destroy(r);
  return rc;
}

}  // extern "C"
