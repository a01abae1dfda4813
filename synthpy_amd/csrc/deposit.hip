// A7-A11 on the device: ray-transfer-matrix optics (rtm_solver.py:48-136 and the chains
// :197-286, :376-422), Rays.histogram = np.histogram2d (:156-178), Interferometry.interferogram
// (:424-453) and interfere_ref_beam (diagnostics.py:559-581).
//
// Detector arithmetic is integer work downstream of a float64 coordinate, so the coordinate
// has to round exactly as the reference's: this file is compiled with -ffp-contract=off and
// fuses only where OpenBLAS' dgemm does (distance: x' = fma(d, theta, x); see the oracle, A7).
// Bin edges are numpy's linspace values i*step + lo (last edge = hi), recomputed per ray.
#include <algorithm>

#include "common.hpp"

namespace {

struct Chain {
  sr_optic op[SR_MAX_OPTICS];
  int n;
  double kwave;
};

struct RefBeam {
  double f, xw, yw;  // exp(1j * f * (xw*x + yw*y)) added to E_y
};
// the reference beams of one deposit, added in order (diagnostics.py:559-581; two_lens_solve adds (10, 20) after the caller's)
struct RefSet {
  int n;
  RefBeam b[SR_MAX_REF_BEAMS];
  __device__ __forceinline__ void add_to(double xm, double ym, double &er, double &ei) const {
    for (int q = 0; q < n; ++q) {  // uses self.rf, still in metres (diagnostics.py:579-581)
      double s, c;
      sincos(b[q].f * (b[q].xw * xm + b[q].yw * ym), &s, &c);
      er += c;
      ei += s;
    }
  }
};

struct Edges {
  double lo, hi, step;
  int n;  // number of bins
};

__device__ __forceinline__ double edge_at(const Edges &e, int i) { return i == e.n ? e.hi : (double)i * e.step + e.lo; }

__device__ __forceinline__ int walk(const Edges &e, double v) {
  int j = (int)((v - e.lo) / e.step);
  j = j < 0 ? 0 : (j > e.n - 1 ? e.n - 1 : j);
  while (j > 0 && v < edge_at(e, j)) --j;
  while (j < e.n - 1 && v >= edge_at(e, j + 1)) ++j;
  return j;
}
// np.histogramdd: searchsorted(edges, v, 'right'), last edge closed, outliers and NaN dropped
__device__ __forceinline__ int bin_hist(const Edges &e, double v) {
  if (!(v >= e.lo && v <= e.hi)) return -1;
  if (v == e.hi) return e.n - 1;
  return walk(e, v);
}
// np.digitize(v, edges) - 1 with the right edge open (rtm_solver.py:442-446)
__device__ __forceinline__ int bin_digitize(const Edges &e, double v) {
  if (!(v >= e.lo && v < e.hi)) return -1;
  return walk(e, v);
}

// could a coordinate within `err` of v fall in another bin (or on the other side of the detector's edge)?
__device__ __forceinline__ bool near_bin_edge(const Edges &e, double v, double err) {
  if (!(v == v)) return false;
  if (v < e.lo) return e.lo - v <= err;
  if (v > e.hi) return v - e.hi <= err;
  const int j = v == e.hi ? e.n - 1 : walk(e, v);
  return v - edge_at(e, j) <= err || edge_at(e, j + 1) - v <= err;
}

struct Ray4 {
  double x, th, y, ph;
  double e0r, e0i, e1r, e1i;
};

// The field factors exp(1j*k*|dr|) of the distance ops commute with everything else in a chain (lenses and masks do
// not touch a surviving ray's E), so their arguments are summed and ONE rotation is applied at the end: one sincos of
// a ~3e8 rad argument per ray instead of one per leg.  (Sum rounding ~6e-8 rad, the size of the reference's own
// rounding of each leg's argument.)
// EDGE GUARD.  The mixed build's exit ray differs from the float64 build's by at most (e_pos, e_ang) in position [mm] and
// angle [rad] on either axis (from the tracer's per-ray bound: trace_mx.inc).  Between its masks the chain is LINEAR,
// x = A*x0 + B*theta0 per axis, so the 2x2 ray-transfer matrices from the chain's start to the current optic are carried
// along (they depend on the chain only: wave-uniform) and the half-width of a coordinate there is |A|*e_pos + |B|*e_ang --
// no interval blow-up: an imaging chain has B = 0 on the detector whatever its legs.  `near` is set when a mask's decision
// or (after the chain) the bin could differ inside that half-width: such rays are traced again in float64 before they are
// counted (sr_rays_deposit).
struct Err4 {
  double e_pos, e_ang;                     // the tracer's bound for this ray
  double ax, bx, cx, dx, ay, by, cy, dy;   // x = ax*x0 + bx*th0, th = cx*x0 + dx*th0; the same for (y, phi)
  bool near;
  // a zero transfer coefficient contributes nothing even when the bound is +inf (0 * inf = NaN would make every compare false)
  static __device__ __forceinline__ double term(double coef, double e) { return coef != 0.0 ? fabs(coef) * e : 0.0; }
  __device__ __forceinline__ double hx() const { return term(ax, e_pos) + term(bx, e_ang); }
  __device__ __forceinline__ double hy() const { return term(ay, e_pos) + term(by, e_ang); }
};

template <bool WITH_E, bool GUARD = false>
__device__ __forceinline__ void apply_chain(const Chain &C, Ray4 &r, Err4 *g = nullptr) {
  double turn = 0.0;
  for (int o = 0; o < C.n; ++o) {
    const sr_optic q = C.op[o];
    bool kill = false;
    if (GUARD) {  // the decision of a mask is looked at BEFORE it is applied; a NaN ray compares false everywhere
      switch (q.op) {
        case SR_OP_DIST:
          g->ax = fma(q.a, g->cx, g->ax);
          g->bx = fma(q.a, g->dx, g->bx);
          g->ay = fma(q.a, g->cy, g->ay);
          g->by = fma(q.a, g->dy, g->by);
          break;
        case SR_OP_LENS:
          g->cx = fma(-1.0 / q.a, g->ax, g->cx);
          g->dx = fma(-1.0 / q.a, g->bx, g->dx);
          g->cy = fma(-1.0 / q.b, g->ay, g->cy);
          g->dy = fma(-1.0 / q.b, g->by, g->dy);
          break;
        case SR_OP_CIRC_AP:
        case SR_OP_CIRC_STOP:
          if (fabs(sqrt(r.x * r.x + r.y * r.y) - fabs(q.a)) <= g->hx() + g->hy()) g->near = true;
          break;
        case SR_OP_RECT_AP:
          if (fabs(fabs(r.x) - fabs(q.a)) <= g->hx() || fabs(fabs(r.y) - fabs(q.b)) <= g->hy()) g->near = true;
          break;
        case SR_OP_KNIFE:
          if (fabs((q.iarg == 0 ? r.x : r.y) - q.a) <= (q.iarg == 0 ? g->hx() : g->hy())) g->near = true;
          break;
        case SR_OP_SCALE:
          g->ax *= q.a;
          g->bx *= q.a;
          g->ay *= q.a;
          g->by *= q.a;
          break;
        default:
          break;
      }
    }
    switch (q.op) {
      case SR_OP_PHASE:
      case SR_OP_DIST: {
        const double xn = fma(q.a, r.th, r.x), yn = fma(q.a, r.ph, r.y);
        if (WITH_E && C.kwave > 0 && q.iarg == 0) {
          const double dx = xn - r.x, dy = yn - r.y;
          turn += C.kwave * sqrt(dx * dx + dy * dy);
        }
        if (q.op == SR_OP_DIST) {
          r.x = xn;
          r.y = yn;
        }
      } break;
      case SR_OP_LENS: {
        const double m1 = -1.0 / q.a, m2 = -1.0 / q.b;
        r.th = m1 * r.x + r.th;
        r.ph = m2 * r.y + r.ph;
      } break;
      case SR_OP_CIRC_AP:
        kill = (r.x * r.x + r.y * r.y > q.a * q.a);
        break;
      case SR_OP_CIRC_STOP:
        kill = (r.x * r.x + r.y * r.y < q.a * q.a);
        break;
      case SR_OP_RECT_AP:
        kill = (r.x * r.x > q.a * q.a) && (r.y * r.y > q.b * q.b);
        break;
      case SR_OP_KNIFE: {
        const double v = q.iarg == 0 ? r.x : r.y;
        kill = q.b > 0 ? (v > q.a) : (q.b < 0 ? (v < q.a) : false);
      } break;
      case SR_OP_SCALE:
        r.x = r.x * q.a;
        r.y = r.y * q.a;
        break;
      default:
        break;
    }
    if (kill) {
      const double nan = __builtin_nan("");
      r.x = r.th = r.y = r.ph = nan;
      if (WITH_E) r.e0r = r.e0i = r.e1r = r.e1i = nan;
    }
  }
  if (WITH_E && turn != 0.0) {
    double s, c;
    sincos(turn, &s, &c);
    double tr = r.e0r * c - r.e0i * s, ti = r.e0r * s + r.e0i * c;
    r.e0r = tr;
    r.e0i = ti;
    tr = r.e1r * c - r.e1i * s;
    ti = r.e1r * s + r.e1i * c;
    r.e1r = tr;
    r.e1i = ti;
  }
}

// host-buffer optics: r (4,N) mm in/out, E (2,N) complex in/out
template <bool WITH_E>
__global__ void k_optics(Chain C, int64_t N, const double *__restrict__ rin, const double *__restrict__ Ein,
                         double *__restrict__ rout, double *__restrict__ Eout) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  Ray4 r{rin[i], rin[N + i], rin[2 * N + i], rin[3 * N + i], 0, 0, 0, 0};
  if (WITH_E) {
    r.e0r = Ein[2 * i];
    r.e0i = Ein[2 * i + 1];
    r.e1r = Ein[2 * (N + i)];
    r.e1i = Ein[2 * (N + i) + 1];
  }
  apply_chain<WITH_E>(C, r);
  rout[i] = r.x;
  rout[N + i] = r.th;
  rout[2 * N + i] = r.y;
  rout[3 * N + i] = r.ph;
  if (WITH_E) {
    Eout[2 * i] = r.e0r;
    Eout[2 * i + 1] = r.e0i;
    Eout[2 * (N + i)] = r.e1r;
    Eout[2 * (N + i) + 1] = r.e1i;
  }
}

__global__ void k_hist2d(const double *__restrict__ x, const double *__restrict__ y, int64_t N, Edges ex, Edges ey,
                         uint32_t *__restrict__ H) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double xv = x[i], yv = y[i];
  if (xv != xv || yv != yv) return;
  const int bx = bin_hist(ex, xv), by = bin_hist(ey, yv);
  if (bx < 0 || by < 0) return;
  atomicAdd(&H[(int64_t)by * ex.n + bx], 1u);
}

__global__ void k_interferogram(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ E,
                                int64_t N, Edges ex, Edges ey, double *__restrict__ amp) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const int bx = bin_digitize(ex, x[i]), by = bin_digitize(ey, y[i]);
  if (bx < 0 || by < 0) return;
  const int64_t plane = (int64_t)ex.n * ey.n, p = (int64_t)by * ex.n + bx;
  unsafeAtomicAdd(&amp[2 * p], E[2 * i]);
  unsafeAtomicAdd(&amp[2 * p + 1], E[2 * i + 1]);
  unsafeAtomicAdd(&amp[2 * (plane + p)], E[2 * (N + i)]);
  unsafeAtomicAdd(&amp[2 * (plane + p) + 1], E[2 * (N + i) + 1]);
}

// H = sqrt(Re(Ax)^2 + Re(Ay)^2)  (rtm_solver.py:450)
__global__ void k_amplitude(const double *__restrict__ amp, int64_t plane, double *__restrict__ H) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const double ax = amp[2 * p], ay = amp[2 * (plane + p)];
  H[p] = sqrt(ax * ax + ay * ay);
}

__global__ void k_ref_beam(const double *__restrict__ x, const double *__restrict__ y, int64_t N, RefBeam R,
                           double *__restrict__ E) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  const double arg = R.f * (R.xw * x[i] + R.yw * y[i]);
  double s, c;
  sincos(arg, &s, &c);
  E[2 * (N + i)] += c;
  E[2 * (N + i) + 1] += s;
}

// sr_rays_optics: the deposit's front end (m_to_mm -> reference beams -> chain) on a resident bundle, written in the ORIGINAL
// ray order (perm[j] = original index of launch slot j) for the host
template <bool WITH_E>
__global__ __launch_bounds__(256) void k_rays_optics(Chain C, RefSet R, int64_t N, const double *__restrict__ rf, const double *__restrict__ Jf,
                                                     const uint32_t *__restrict__ perm, double *__restrict__ rout, double *__restrict__ Eout) {
  const int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (j >= N) return;
  const double xm = rf[j], ym = rf[2 * N + j];
  Ray4 r{xm * 1e3, rf[N + j], ym * 1e3, rf[3 * N + j], 0, 0, 0, 0};
  if (WITH_E) {
    r.e0r = Jf[2 * j];
    r.e0i = Jf[2 * j + 1];
    r.e1r = Jf[2 * (N + j)];
    r.e1i = Jf[2 * (N + j) + 1];
    R.add_to(xm, ym, r.e1r, r.e1i);
  }
  apply_chain<WITH_E>(C, r);
  const int64_t i = perm[j];
  rout[i] = r.x;
  rout[N + i] = r.th;
  rout[2 * N + i] = r.y;
  rout[3 * N + i] = r.ph;
  if (WITH_E) {
    Eout[2 * i] = r.e0r;
    Eout[2 * i + 1] = r.e0i;
    Eout[2 * (N + i)] = r.e1r;
    Eout[2 * (N + i) + 1] = r.e1i;
  }
}

__global__ void k_counts_f64(const uint32_t *__restrict__ cnt, int64_t n, double *__restrict__ H) {
  const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (p < n) H[p] = (double)cnt[p];
}

// Fused deposit: exit-plane rays in HBM (metres) -> m_to_mm -> [reference beam] -> chain -> detector.
//
// The rays arrive in launch order, i.e. binned by entry cell, and the imaging chains map neighbouring rays
// to neighbouring pixels, so the 256 hits of a workgroup fall in a compact patch of the detector.  With
// TILED the workgroup privatises that patch in LDS: the patch origin is the minimum (bx, by) over the
// workgroup (LDS atomicMin), hits inside the TW x TH tile are LDS atomics, the few outside it go straight
// to HBM, and the tile is flushed with one global atomic per NON-EMPTY bin, row by row (coalesced).
// Integer counts are order-independent, so the image is bit-identical with and without tiles; the complex
// image sums in a different order (float64 atomics either way).
constexpr int kTileW = 64, kTileH = 32;   // counts: 2048 bins, 8 KiB of LDS
constexpr int kCTileW = 32, kCTileH = 16;  // complex: 512 bins x 4 doubles, 16 KiB of LDS

struct Guard {
  const float *bound;         // per launch slot: angle bound of the mixed build [rad]; 0 = a float64 result
  double len;                 // position bound = len * angle bound [m]
  uint32_t *list;             // the slots to trace again
  unsigned long long *count;
};

template <int KIND, bool TILED, bool GUARD = false>
__global__ __launch_bounds__(256) void k_deposit(Chain C, RefSet R, int64_t N, const double *__restrict__ rf,
                                                 const double *__restrict__ Jf, Edges ex, Edges ey, void *__restrict__ img,
                                                 unsigned long long *__restrict__ counter, Guard G) {
  constexpr int TW = KIND == SR_IMG_COMPLEX ? kCTileW : kTileW, TH = KIND == SR_IMG_COMPLEX ? kCTileH : kTileH;
  __shared__ int org[2];
  __shared__ double tile_store[TILED ? (KIND == SR_IMG_COMPLEX ? TW * TH * 4 : TW * TH / 2) : 1];
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (TILED) {
    if (threadIdx.x < 2) org[threadIdx.x] = 0x7fffffff;
    if (KIND == SR_IMG_COMPLEX) {
      for (int t = threadIdx.x; t < TW * TH * 4; t += blockDim.x) tile_store[t] = 0.0;
    } else {
      uint32_t *tc = reinterpret_cast<uint32_t *>(tile_store);
      for (int t = threadIdx.x; t < TW * TH; t += blockDim.x) tc[t] = 0u;
    }
    __syncthreads();
  }
  int bx = -1, by = -1;
  bool again = false;
  Ray4 r{0, 0, 0, 0, 0, 0, 0, 0};
  if (i < N) {
    const double xm = rf[i], ym = rf[2 * N + i];
    r = Ray4{xm * 1e3, rf[N + i], ym * 1e3, rf[3 * N + i], 0, 0, 0, 0};  // m_to_mm (rtm_solver.py:48-51)
    if (KIND == SR_IMG_COMPLEX) {
      r.e0r = Jf[2 * i];
      r.e0i = Jf[2 * i + 1];
      r.e1r = Jf[2 * (N + i)];
      r.e1i = Jf[2 * (N + i) + 1];
      R.add_to(xm, ym, r.e1r, r.e1i);
      apply_chain<true>(C, r);
      bx = bin_digitize(ex, r.x);
      by = bin_digitize(ey, r.y);
    } else if (GUARD) {
      const double ea = (double)G.bound[i];
      if (!(ea < INFINITY)) {  // a kernel that keeps no bound (sub-steps, optional terms): every such ray is traced again
        again = true;
      } else if (ea > 0.0) {  // a mixed-precision result: is its pixel (and every mask's decision) the float64 result's as well?
        // 1.0000001: the few roundings of the half-widths themselves
        Err4 g{1e3 * (G.len * ea) * 1.0000001, ea * 1.0000001, 1, 0, 0, 1, 1, 0, 0, 1, false};
        apply_chain<false, true>(C, r, &g);
        again = g.near || near_bin_edge(ex, r.x, g.hx()) || near_bin_edge(ey, r.y, g.hy());
      } else {
        apply_chain<false>(C, r);
      }
      if (!again && r.x == r.x && r.y == r.y) {
        bx = bin_hist(ex, r.x);
        by = bin_hist(ey, r.y);
      }
    } else {
      apply_chain<false>(C, r);
      if (r.x == r.x && r.y == r.y) {
        bx = bin_hist(ex, r.x);
        by = bin_hist(ey, r.y);
      }
    }
  }
  if (GUARD) sr::queue_push(G.count, G.list, again, (uint32_t)i);  // counted after their float64 re-trace (k_deposit_list)
  const bool hit = bx >= 0 && by >= 0;
  int tx = -1, ty = -1;
  if (TILED) {
    if (hit) {
      atomicMin(&org[0], bx);
      atomicMin(&org[1], by);
    }
    __syncthreads();
    tx = bx - org[0];
    ty = by - org[1];
  }
  const bool in_tile = TILED && hit && tx < TW && ty < TH;  // tx, ty >= 0 by construction of the origin
  if (hit) {
    if (KIND == SR_IMG_COMPLEX) {
      // a zero contribution is skipped (x + 0 = x): E_x of an unrotated ray is exactly -0, half of the atomics
      if (in_tile) {
        double *t = tile_store + (size_t)(ty * TW + tx) * 4;
        if (r.e0r != 0.0) unsafeAtomicAdd(&t[0], r.e0r);
        if (r.e0i != 0.0) unsafeAtomicAdd(&t[1], r.e0i);
        if (r.e1r != 0.0) unsafeAtomicAdd(&t[2], r.e1r);
        if (r.e1i != 0.0) unsafeAtomicAdd(&t[3], r.e1i);
      } else {
        double *amp = (double *)img;
        const int64_t plane = (int64_t)ex.n * ey.n, p = (int64_t)by * ex.n + bx;
        if (r.e0r != 0.0) unsafeAtomicAdd(&amp[2 * p], r.e0r);
        if (r.e0i != 0.0) unsafeAtomicAdd(&amp[2 * p + 1], r.e0i);
        if (r.e1r != 0.0) unsafeAtomicAdd(&amp[2 * (plane + p)], r.e1r);
        if (r.e1i != 0.0) unsafeAtomicAdd(&amp[2 * (plane + p) + 1], r.e1i);
      }
    } else {
      if (in_tile)
        atomicAdd(&reinterpret_cast<uint32_t *>(tile_store)[ty * TW + tx], 1u);
      else
        atomicAdd(&((uint32_t *)img)[(int64_t)by * ex.n + bx], 1u);
    }
  }
  if (TILED) {
    __syncthreads();
    const int ox = org[0], oy = org[1];
    if (ox != 0x7fffffff) {  // at least one hit in this workgroup
      for (int t = threadIdx.x; t < TW * TH; t += blockDim.x) {
        const int gx = ox + t % TW, gy = oy + t / TW;
        if (gx >= ex.n || gy >= ey.n) continue;
        if (KIND == SR_IMG_COMPLEX) {
          const double *s = tile_store + (size_t)t * 4;
          if (s[0] != 0.0 || s[1] != 0.0 || s[2] != 0.0 || s[3] != 0.0) {
            double *amp = (double *)img;
            const int64_t plane = (int64_t)ex.n * ey.n, p = (int64_t)gy * ex.n + gx;
            if (s[0] != 0.0) unsafeAtomicAdd(&amp[2 * p], s[0]);
            if (s[1] != 0.0) unsafeAtomicAdd(&amp[2 * p + 1], s[1]);
            if (s[2] != 0.0) unsafeAtomicAdd(&amp[2 * (plane + p)], s[2]);
            if (s[3] != 0.0) unsafeAtomicAdd(&amp[2 * (plane + p) + 1], s[3]);
          }
        } else {
          const uint32_t cnt = reinterpret_cast<const uint32_t *>(tile_store)[t];
          if (cnt) atomicAdd(&((uint32_t *)img)[(int64_t)gy * ex.n + gx], cnt);
        }
      }
    }
  }
  unsigned long long tot = hit ? 1ull : 0ull;
  for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, 64);
  if ((threadIdx.x & 63) == 0 && tot) atomicAdd(sr::stripe(counter, 1), tot);
}

// counts deposit of the launch slots list[0..*count): the rays the edge guard had traced again (few; global atomics)
__global__ __launch_bounds__(256) void k_deposit_list(Chain C, int64_t N, const double *__restrict__ rf, Edges ex, Edges ey,
                                                      uint32_t *__restrict__ img, unsigned long long *__restrict__ counter,
                                                      const uint32_t *__restrict__ list, const unsigned long long *__restrict__ count) {
  const unsigned long long n = *count;
  unsigned long long hits = 0;
  for (unsigned long long t = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; t < n; t += (unsigned long long)gridDim.x * blockDim.x) {
    const int64_t i = list[t];
    Ray4 r{rf[i] * 1e3, rf[N + i], rf[2 * N + i] * 1e3, rf[3 * N + i], 0, 0, 0, 0};
    apply_chain<false>(C, r);
    if (r.x == r.x && r.y == r.y) {
      const int bx = bin_hist(ex, r.x), by = bin_hist(ey, r.y);
      if (bx >= 0 && by >= 0) {
        atomicAdd(&img[(int64_t)by * ex.n + bx], 1u);
        ++hits;
      }
    }
  }
  if (hits) atomicAdd(sr::stripe(counter, 1), hits);
}

// Edge guard for SEVERAL counts diagnostics at once (sr_rays_refine): a ray is queued when its pixel or a mask's decision is
// not certain for ANY of them; one float64 re-trace then serves every deposit that follows.
struct GuardSet {
  Chain C[SR_MAX_REFINE];
  Edges ex[SR_MAX_REFINE], ey[SR_MAX_REFINE];
  int n;
};
__global__ __launch_bounds__(256) void k_guard_flags(const GuardSet *__restrict__ S, int64_t N, const double *__restrict__ rf, Guard G) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  bool again = false;
  if (i < N) {
    const double ea = (double)G.bound[i];
    if (!(ea < INFINITY)) {
      again = true;
    } else if (ea > 0.0) {
      for (int q = 0; q < S->n && !again; ++q) {
        Ray4 r{rf[i] * 1e3, rf[N + i], rf[2 * N + i] * 1e3, rf[3 * N + i], 0, 0, 0, 0};
        Err4 g{1e3 * (G.len * ea) * 1.0000001, ea * 1.0000001, 1, 0, 0, 1, 1, 0, 0, 1, false};
        apply_chain<false, true>(S->C[q], r, &g);
        again = g.near || near_bin_edge(S->ex[q], r.x, g.hx()) || near_bin_edge(S->ey[q], r.y, g.hy());
      }
    }
  }
  sr::queue_push(G.count, G.list, again, (uint32_t)i);
}

int make_chain(const sr_optic *chain, int n_ops, double kwave, Chain &C) {
  SR_CHECK(n_ops >= 0 && n_ops <= SR_MAX_OPTICS, "optic chain length %d out of range (0..%d)", n_ops, SR_MAX_OPTICS);
  SR_CHECK(n_ops == 0 || chain != nullptr, "optic chain is NULL");
  C.n = n_ops;
  C.kwave = kwave;
  for (int i = 0; i < n_ops; ++i) {
    SR_CHECK(chain[i].op >= SR_OP_DIST && chain[i].op <= SR_OP_PHASE, "unknown optic op %d at position %d", chain[i].op, i);
    if (chain[i].op == SR_OP_LENS) SR_CHECK(chain[i].a != 0 && chain[i].b != 0, "lens focal length must be non-zero");
    C.op[i] = chain[i];
  }
  return SR_OK;
}

RefBeam make_ref(double n_fringes, double deg) {
  RefBeam R{0, 0, 0};
  if (deg >= 45) deg = -fabs(deg - 90);
  const double rad = deg * M_PI / 180;
  R.yw = atan(rad);
  R.xw = sqrt(1 - R.yw * R.yw);
  R.f = 2 * n_fringes / 3;
  return R;
}
int make_refs(const sr_deposit_params *p, RefSet &R) {
  R.n = 0;
  for (auto &b : R.b) b = RefBeam{0, 0, 0};
  if (!p) return SR_OK;
  SR_CHECK(p->ref_on >= 0 && p->ref_on <= SR_MAX_REF_BEAMS, "ref_on = %d: at most %d reference beams", p->ref_on, SR_MAX_REF_BEAMS);
  R.n = p->ref_on;
  for (int q = 0; q < R.n; ++q) R.b[q] = make_ref(p->ref_n_fringes[q], p->ref_deg[q]);
  return SR_OK;
}

Edges make_edges(double lo, double hi, int nbins) {
  Edges e;
  e.lo = lo;
  e.hi = hi;
  e.n = nbins;
  e.step = (hi - lo) / nbins;  // np.linspace: delta / div
  return e;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { sr::dev_free(p); }
  int alloc(size_t bytes) {
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SR_OK;
  }
};

}  // namespace

extern "C" {

int sr_optics(const sr_optic *chain, int n_ops, double kwave, int64_t N, const double *r_in, const double *E_in,
              double *r_out, double *E_out) {
  SR_CHECK(N >= 0 && r_in && r_out, "sr_optics: bad argument");
  SR_CHECK((E_in == nullptr) == (E_out == nullptr), "sr_optics: E_in and E_out must both be given or both NULL");
  Chain C;
  int rc = make_chain(chain, n_ops, kwave, C);
  if (rc) return rc;
  if (N == 0) return SR_OK;
  if ((rc = sr::ensure_init())) return rc;
  hipStream_t st = sr::ctx().stream;
  DevBuf dr, dE;
  if ((rc = dr.alloc(sizeof(double) * 4 * N))) return rc;
  SR_HIP(hipMemcpyAsync(dr.p, r_in, sizeof(double) * 4 * N, hipMemcpyHostToDevice, st));
  if (E_in) {
    if ((rc = dE.alloc(sizeof(double) * 4 * N))) return rc;
    SR_HIP(hipMemcpyAsync(dE.p, E_in, sizeof(double) * 4 * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL((k_optics<true>), dim3(sr::grid_for(N, 256)), dim3(256), 0, st, C, N, (const double *)dr.p,
                       (const double *)dE.p, (double *)dr.p, (double *)dE.p);
  } else {
    hipLaunchKernelGGL((k_optics<false>), dim3(sr::grid_for(N, 256)), dim3(256), 0, st, C, N, (const double *)dr.p,
                       (const double *)nullptr, (double *)dr.p, (double *)nullptr);
  }
  SR_HIP(hipGetLastError());
  SR_HIP(hipMemcpyAsync(r_out, dr.p, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st));
  if (E_out) SR_HIP(hipMemcpyAsync(E_out, dE.p, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int sr_hist2d(const double *x, const double *y, int64_t N, int nxb, int nyb, double x_lo, double x_hi, double y_lo,
              double y_hi, uint32_t *H) {
  SR_CHECK(N >= 0 && (N == 0 || (x && y)) && H, "sr_hist2d: bad argument");
  SR_CHECK(nxb >= 1 && nyb >= 1, "sr_hist2d: bins must be positive");
  SR_CHECK(x_hi > x_lo && y_hi > y_lo, "sr_hist2d: empty range");
  int rc = sr::ensure_init();
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;
  DevBuf dx, dy, dH;
  const size_t hb = sizeof(uint32_t) * (size_t)nxb * nyb;
  if ((rc = dH.alloc(hb))) return rc;
  SR_HIP(hipMemsetAsync(dH.p, 0, hb, st));
  if (N > 0) {
    if ((rc = dx.alloc(sizeof(double) * N)) || (rc = dy.alloc(sizeof(double) * N))) return rc;
    SR_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * N, hipMemcpyHostToDevice, st));
    SR_HIP(hipMemcpyAsync(dy.p, y, sizeof(double) * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_hist2d, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, (const double *)dx.p, (const double *)dy.p, N,
                       make_edges(x_lo, x_hi, nxb), make_edges(y_lo, y_hi, nyb), (uint32_t *)dH.p);
    SR_HIP(hipGetLastError());
  }
  SR_HIP(hipMemcpyAsync(H, dH.p, hb, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int sr_interferogram(const double *x, const double *y, const double *E, int64_t N, int nxe, int nye, double x_lo,
                     double x_hi, double y_lo, double y_hi, double *amp, double *H) {
  SR_CHECK(N >= 0 && (N == 0 || (x && y && E)) && (amp || H), "sr_interferogram: bad argument");
  SR_CHECK(nxe >= 2 && nye >= 2, "sr_interferogram: need at least 2 edges per axis");
  SR_CHECK(x_hi > x_lo && y_hi > y_lo, "sr_interferogram: empty range");
  int rc = sr::ensure_init();
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;
  DevBuf dx, dy, dE, dA;
  const size_t ab = sizeof(double) * 4 * (size_t)(nxe - 1) * (nye - 1);
  if ((rc = dA.alloc(ab))) return rc;
  SR_HIP(hipMemsetAsync(dA.p, 0, ab, st));
  if (N > 0) {
    if ((rc = dx.alloc(sizeof(double) * N)) || (rc = dy.alloc(sizeof(double) * N)) || (rc = dE.alloc(sizeof(double) * 4 * N)))
      return rc;
    SR_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * N, hipMemcpyHostToDevice, st));
    SR_HIP(hipMemcpyAsync(dy.p, y, sizeof(double) * N, hipMemcpyHostToDevice, st));
    SR_HIP(hipMemcpyAsync(dE.p, E, sizeof(double) * 4 * N, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_interferogram, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, (const double *)dx.p,
                       (const double *)dy.p, (const double *)dE.p, N, make_edges(x_lo, x_hi, nxe - 1),
                       make_edges(y_lo, y_hi, nye - 1), (double *)dA.p);
    SR_HIP(hipGetLastError());
  }
  if (amp) SR_HIP(hipMemcpyAsync(amp, dA.p, ab, hipMemcpyDeviceToHost, st));
  if (H) {
    DevBuf dH;
    const int64_t plane = (int64_t)(nxe - 1) * (nye - 1);
    if ((rc = dH.alloc(sizeof(double) * plane))) return rc;
    hipLaunchKernelGGL(k_amplitude, dim3(sr::grid_for(plane, 256)), dim3(256), 0, st, (const double *)dA.p, plane, (double *)dH.p);
    SR_HIP(hipGetLastError());
    SR_HIP(hipMemcpyAsync(H, dH.p, sizeof(double) * plane, hipMemcpyDeviceToHost, st));
    SR_HIP(hipStreamSynchronize(st));
    return SR_OK;
  }
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int sr_interfere_ref_beam(const double *x, const double *y, int64_t N, double n_fringes, double deg, double *E) {
  SR_CHECK(N >= 0 && (N == 0 || (x && y && E)), "sr_interfere_ref_beam: bad argument");
  if (N == 0) return SR_OK;
  int rc = sr::ensure_init();
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;
  DevBuf dx, dy, dE;
  if ((rc = dx.alloc(sizeof(double) * N)) || (rc = dy.alloc(sizeof(double) * N)) || (rc = dE.alloc(sizeof(double) * 4 * N)))
    return rc;
  SR_HIP(hipMemcpyAsync(dx.p, x, sizeof(double) * N, hipMemcpyHostToDevice, st));
  SR_HIP(hipMemcpyAsync(dy.p, y, sizeof(double) * N, hipMemcpyHostToDevice, st));
  SR_HIP(hipMemcpyAsync(dE.p, E, sizeof(double) * 4 * N, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_ref_beam, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, (const double *)dx.p, (const double *)dy.p, N,
                     make_ref(n_fringes, deg), (double *)dE.p);
  SR_HIP(hipGetLastError());
  SR_HIP(hipMemcpyAsync(E, dE.p, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

// ---- device-resident images -------------------------------------------------------------
void sr_image_destroy(sr_image *img) {
  if (!img) return;
  sr::dev_free(img->d);
  delete img;
}

int sr_image_create(sr_image **out, int kind, int nx, int ny, double x_lo, double x_hi, double y_lo, double y_hi) {
  SR_CHECK(out != nullptr, "sr_image_create: NULL out");
  *out = nullptr;
  SR_CHECK(kind == SR_IMG_COUNTS || kind == SR_IMG_COMPLEX, "sr_image_create: unknown kind %d", kind);
  SR_CHECK(x_hi > x_lo && y_hi > y_lo, "sr_image_create: empty range");
  if (kind == SR_IMG_COUNTS)
    SR_CHECK(nx >= 1 && ny >= 1, "sr_image_create: bins must be positive");
  else
    SR_CHECK(nx >= 2 && ny >= 2, "sr_image_create: need at least 2 edges per axis");
  int rc = sr::ensure_init();
  if (rc) return rc;
  sr_image *img = new sr_image();
  img->kind = kind;
  img->nx = nx;
  img->ny = ny;
  img->x_lo = x_lo;
  img->x_hi = x_hi;
  img->y_lo = y_lo;
  img->y_hi = y_hi;
  img->bytes = kind == SR_IMG_COUNTS ? (int64_t)sizeof(uint32_t) * nx * ny : (int64_t)sizeof(double) * 4 * (nx - 1) * (ny - 1);
  hipError_t e = hipMalloc(&img->d, (size_t)img->bytes);
  if (e != hipSuccess) {
    delete img;
    return sr::fail(SR_ERR_HIP, "sr_image_create: hipMalloc(%lld) failed: %s", (long long)img->bytes, hipGetErrorString(e));
  }
  e = hipMemsetAsync(img->d, 0, (size_t)img->bytes, sr::ctx().stream);
  if (e != hipSuccess) {
    sr_image_destroy(img);
    return sr::fail(SR_ERR_HIP, "sr_image_create: memset failed: %s", hipGetErrorString(e));
  }
  *out = img;
  return SR_OK;
}

int sr_image_zero(sr_image *img) {
  SR_CHECK(img != nullptr, "sr_image_zero: NULL image");
  SR_HIP(hipMemsetAsync(img->d, 0, (size_t)img->bytes, sr::ctx().stream));
  return SR_OK;
}

int sr_image_download(const sr_image *img, void *host) {
  SR_CHECK(img && host, "sr_image_download: NULL argument");
  SR_HIP(hipMemcpyAsync(host, img->d, (size_t)img->bytes, hipMemcpyDeviceToHost, sr::ctx().stream));
  SR_HIP(hipStreamSynchronize(sr::ctx().stream));
  return SR_OK;
}

int sr_image_amplitude(const sr_image *img, double *H) {
  SR_CHECK(img && H, "sr_image_amplitude: NULL argument");
  SR_CHECK(img->kind == SR_IMG_COMPLEX, "sr_image_amplitude: image does not hold a complex field");
  hipStream_t st = sr::ctx().stream;
  const int64_t plane = (int64_t)(img->nx - 1) * (img->ny - 1);
  double *dH = static_cast<double *>(sr::scratch(sizeof(double) * plane));
  if (!dH) return SR_ERR_HIP;
  hipLaunchKernelGGL(k_amplitude, dim3(sr::grid_for(plane, 256)), dim3(256), 0, st, (const double *)img->d, plane, dH);
  SR_HIP(hipGetLastError());
  SR_HIP(hipMemcpyAsync(H, dH, sizeof(double) * plane, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int64_t sr_image_bytes(const sr_image *img) { return img ? img->bytes : 0; }

int sr_image_counts_f64(const sr_image *img, double *H) {
  SR_CHECK(img && H, "sr_image_counts_f64: NULL argument");
  SR_CHECK(img->kind == SR_IMG_COUNTS, "sr_image_counts_f64: image does not hold counts");
  hipStream_t st = sr::ctx().stream;
  const int64_t n = (int64_t)img->nx * img->ny;
  double *dH = static_cast<double *>(sr::scratch(sizeof(double) * n));
  if (!dH) return SR_ERR_HIP;
  hipLaunchKernelGGL(k_counts_f64, dim3(sr::grid_for(n, 256)), dim3(256), 0, st, (const uint32_t *)img->d, n, dH);
  SR_HIP(hipGetLastError());
  SR_HIP(hipMemcpyAsync(H, dH, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  return SR_OK;
}

int sr_rays_optics(const sr_rays *r, const sr_optic *chain, int n_ops, const sr_deposit_params *p, double *rf_out, double *E_out) {
  SR_CHECK(r && rf_out, "sr_rays_optics: NULL argument");
  if (!r->traced) return sr::fail(SR_ERR_STATE, "sr_rays_optics: rays have not been traced");
  Chain C;
  int rc = make_chain(chain, n_ops, p ? p->kwave : 0.0, C);
  if (rc) return rc;
  RefSet R;
  if ((rc = make_refs(p, R))) return rc;
  const int64_t N = r->n;
  if (N == 0) return SR_OK;
  hipStream_t st = sr::ctx().stream;
  double *dr = static_cast<double *>(sr::scratch(sizeof(double) * (E_out ? 8 : 4) * (size_t)N));
  if (!dr) return SR_ERR_HIP;
  double *dE = E_out ? dr + 4 * (size_t)N : nullptr;
  if (E_out)
    hipLaunchKernelGGL((k_rays_optics<true>), dim3(sr::grid_for(N, 256)), dim3(256), 0, st, C, R, N, (const double *)r->rf,
                       (const double *)r->Jf, (const uint32_t *)r->perm, dr, dE);
  else
    hipLaunchKernelGGL((k_rays_optics<false>), dim3(sr::grid_for(N, 256)), dim3(256), 0, st, C, R, N, (const double *)r->rf,
                       (const double *)nullptr, (const uint32_t *)r->perm, dr, (double *)nullptr);
  SR_HIP(hipGetLastError());
  SR_HIP(hipMemcpyAsync(rf_out, dr, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st));
  if (E_out) SR_HIP(hipMemcpyAsync(E_out, dE, sizeof(double) * 4 * N, hipMemcpyDeviceToHost, st));
  SR_HIP(hipStreamSynchronize(st));
  sr::scratch_trim();
  return SR_OK;
}

int sr_rays_refine(const sr_rays *r, int n_diag, const sr_optic *const *chains, const int *n_ops, sr_image *const *imgs,
                   int64_t *retraced) {
  SR_CHECK(r != nullptr && n_diag >= 0 && n_diag <= SR_MAX_REFINE, "sr_rays_refine: bad argument (at most %d diagnostics)", SR_MAX_REFINE);
  if (!r->traced) return sr::fail(SR_ERR_STATE, "sr_rays_refine: rays have not been traced");
  if (retraced) *retraced = 0;
  if (!r->guard_live || r->n == 0 || n_diag == 0) return SR_OK;  // float64 results already: nothing to refine
  GuardSet S;
  S.n = 0;
  for (int q = 0; q < n_diag; ++q) {
    SR_CHECK(chains && n_ops && imgs && imgs[q], "sr_rays_refine: NULL diagnostic %d", q);
    if (imgs[q]->kind != SR_IMG_COUNTS) continue;  // complex images are not counts: no guard (engine.resolve_precision)
    int rc = make_chain(chains[q], n_ops[q], 0.0, S.C[S.n]);
    if (rc) return rc;
    S.ex[S.n] = make_edges(imgs[q]->x_lo, imgs[q]->x_hi, imgs[q]->nx);
    S.ey[S.n] = make_edges(imgs[q]->y_lo, imgs[q]->y_hi, imgs[q]->ny);
    ++S.n;
  }
  if (S.n == 0) return SR_OK;
  sr::Context &c = sr::ctx();
  hipStream_t st = c.stream;
  const int64_t N = r->n;
  // the set travels through a buffer that belongs to the bundle (device side) and a copy the bundle keeps (host side): no wait
  // is needed for either, so a chunked driver can queue refine + deposits without a host round trip (retraced == NULL)
  sr_rays *rw = const_cast<sr_rays *>(r);
  if (!rw->guard_set) SR_HIP(hipMalloc(&rw->guard_set, sizeof(GuardSet)));
  rw->guard_set_host.assign(reinterpret_cast<const char *>(&S), reinterpret_cast<const char *>(&S) + sizeof(GuardSet));
  SR_HIP(hipMemcpyAsync(rw->guard_set, rw->guard_set_host.data(), sizeof(GuardSet), hipMemcpyHostToDevice, st));
  int rc = SR_OK;
  SR_HIP(hipMemsetAsync(r->counters + 4, 0, 2 * sizeof(unsigned long long), st));
  SR_HIP(hipMemsetAsync(r->counters + 16 + 2 * (size_t)sr::kStripes * sr::kStripeStride, 0,
                        sizeof(unsigned long long) * sr::kStripes * sr::kStripeStride, st));
  Guard G{r->guard, r->guard_len, r->fb_list, r->counters + 4};
  hipLaunchKernelGGL(k_guard_flags, dim3(sr::grid_for(N, 256)), dim3(256), 0, st, (const GuardSet *)rw->guard_set, N, (const double *)r->rf, G);
  SR_HIP(hipGetLastError());
  rc = sr::retrace_f64(r, r->fb_list, r->counters + 4);
  if (rc) return rc;
  if (retraced) {
    unsigned long long n_again = 0;
    SR_HIP(hipMemcpyAsync(&n_again, r->counters + 4, sizeof n_again, hipMemcpyDeviceToHost, st));
    SR_HIP(hipStreamSynchronize(st));
    *retraced = (int64_t)n_again;
  }
  return SR_OK;
}

int sr_rays_deposit(const sr_rays *r, const sr_optic *chain, int n_ops, const sr_deposit_params *p, sr_image *img,
                    sr_deposit_stats *stats) {
  SR_CHECK(r && img, "sr_rays_deposit: NULL argument");
  if (!r->traced) return sr::fail(SR_ERR_STATE, "sr_rays_deposit: rays have not been traced");
  Chain C;
  int rc = make_chain(chain, n_ops, p ? p->kwave : 0.0, C);
  if (rc) return rc;
  RefSet R;
  if ((rc = make_refs(p, R))) return rc;
  sr::Context &c = sr::ctx();
  hipStream_t st = c.stream;
  const int64_t N = r->n;
  if (stats) *stats = sr_deposit_stats{0.0, 0, 0};
  if (N == 0) return SR_OK;
  unsigned long long *dep_stripes = r->counters + 16 + (size_t)sr::kStripes * sr::kStripeStride;
  SR_HIP(hipMemsetAsync(dep_stripes, 0, sizeof(unsigned long long) * sr::kStripes * sr::kStripeStride, st));
  SR_HIP(hipEventRecord(c.ev[0], st));
  const unsigned grid = sr::grid_for(N, 256);
  const bool tiled = p ? p->lds_tiles != 0 : true;
  const bool cplx = img->kind == SR_IMG_COMPLEX;
  const Edges ex = make_edges(img->x_lo, img->x_hi, cplx ? img->nx - 1 : img->nx);
  const Edges ey = make_edges(img->y_lo, img->y_hi, cplx ? img->ny - 1 : img->ny);
  const double *rf = r->rf, *Jf = r->Jf;
  unsigned long long *cnt = r->counters;
  // exact counts (the default): rays of a mixed-precision trace whose pixel or mask decision is not certain are traced
  // again in float64 and counted afterwards -- the image is the float64 build's, integer for integer
  const bool exact = !cplx && (p ? p->exact_counts != 0 : true) && r->guard_live;
  Guard G{r->guard, r->guard_len, r->fb_list, r->counters + 4};
#define SR_DEP(KIND, T, GD) hipLaunchKernelGGL((k_deposit<KIND, T, GD>), dim3(grid), dim3(256), 0, st, C, R, N, rf, Jf, ex, ey, img->d, cnt, G)
  if (cplx) {
    if (tiled)
      SR_DEP(SR_IMG_COMPLEX, true, false);
    else
      SR_DEP(SR_IMG_COMPLEX, false, false);
  } else if (exact) {
    SR_HIP(hipMemsetAsync(r->counters + 4, 0, 2 * sizeof(unsigned long long), st));
    SR_HIP(hipMemsetAsync(r->counters + 16 + 2 * (size_t)sr::kStripes * sr::kStripeStride, 0,
                          sizeof(unsigned long long) * sr::kStripes * sr::kStripeStride, st));
    if (tiled)
      SR_DEP(SR_IMG_COUNTS, true, true);
    else
      SR_DEP(SR_IMG_COUNTS, false, true);
    SR_HIP(hipGetLastError());
    rc = sr::retrace_f64(r, r->fb_list, r->counters + 4);
    if (rc) return rc;
    const unsigned lgrid = (unsigned)std::min<int64_t>(grid, (int64_t)c.n_cu * 8);
    hipLaunchKernelGGL(k_deposit_list, dim3(lgrid), dim3(256), 0, st, C, N, rf, ex, ey, (uint32_t *)img->d, cnt,
                       (const uint32_t *)r->fb_list, (const unsigned long long *)(r->counters + 4));
  } else {
    if (tiled)
      SR_DEP(SR_IMG_COUNTS, true, false);
    else
      SR_DEP(SR_IMG_COUNTS, false, false);
  }
#undef SR_DEP
  SR_HIP(hipGetLastError());
  SR_HIP(hipEventRecord(c.ev[1], st));
  if (stats) {
    std::vector<unsigned long long> hw(sr::kCounterWords, 0ull);
    SR_HIP(hipMemcpyAsync(hw.data(), r->counters, sizeof(unsigned long long) * sr::kCounterWords, hipMemcpyDeviceToHost, st));
    SR_HIP(hipStreamSynchronize(st));
    const unsigned long long h = sr::stripe_sum(hw.data(), 1);
    float ms = 0.f;
    SR_HIP(hipEventElapsedTime(&ms, c.ev[0], c.ev[1]));
    stats->kernel_ms = ms;
    stats->deposited = (int64_t)h;
    stats->retraced = exact ? (int64_t)hw[4] : 0;
  }
  return SR_OK;
}

}  // extern "C"
