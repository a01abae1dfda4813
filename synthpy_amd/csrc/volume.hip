// A1 + A5 on the device: calc_dndr (full_solver.py:211-234) and n_refrac (:271-274).
//
// HBM layout produced here (see common.hpp): one float4 per node
//   P[sr::node_index(ia, ib, ic)] = { dnd_b, dnd_c, dnd_a, hi(n-1) },  L[...] = lo(n-1)
// in octets of eight node planes along the probing axis `a`: a ray marching along `a` walks four
// line-sized streams and each 128-byte line serves eight consecutive node planes of one column.
// The file is compiled with -ffp-contract=off: the float32 gradient arithmetic must round
// exactly as numpy's separate multiply and add ufuncs do.
#include "common.hpp"

namespace {

// np.gradient coefficients of one axis (numpy/lib/_function_base_impl.py `gradient`)
struct AxisCoef {
  const float *ca, *cb, *cc;  // non-uniform interior a, b, c (index = node)
  float dx0, dxn, two_dx;     // edge spacings; 2*dx for the uniform interior
  int n;
  int uniform;
};

struct PackArgs {
  int nx, ny, nz;
  int axis;        // physical index of the fastest packed axis
  int na, nb, nc;  // packed dims
  int sny, snz;    // y, z extents of the SOURCE array (a slab's source carries halo planes on the probing axis)
  int a_src_off;   // source index of the first packed plane along `axis`
  int a_glob_off;  // whole-domain index of the first packed plane along `axis` (np.gradient's edge rule and coefficients)
  AxisCoef co[3];  // physical x, y, z
  float scale;     // float32(-0.5*c**2)
  double omega;
};

__device__ __forceinline__ float grad1(const float *f, int64_t idx, int64_t stride, int i, const AxisCoef &c) {
  if (i == 0) return (f[idx + stride] - f[idx]) / c.dx0;
  if (i == c.n - 1) return (f[idx] - f[idx - stride]) / c.dxn;
  if (c.uniform) return (f[idx + stride] - f[idx - stride]) / c.two_dx;
  const float t1 = c.ca[i] * f[idx - stride];
  const float t2 = c.cb[i] * f[idx];
  const float t3 = c.cc[i] * f[idx + stride];
  return (t1 + t2) + t3;
}

// ne_nc = float32(ne / n_c)   (full_solver.py:225)
__global__ void k_ne_nc_f64(const double *__restrict__ ne, int64_t n, double nc, float *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = (float)(ne[i] / nc);
}
__global__ void k_ne_nc_f32(const float *__restrict__ ne, int64_t n, float ncf, float *__restrict__ out) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    out[i] = ne[i] / ncf;
}

__device__ __forceinline__ void split_hi_lo(double v, float &hi, float &lo) {
  hi = (float)v;
  lo = (float)(v - (double)hi);
}

// One thread per packed voxel: gradients of ne_nc along x, y, z (numpy order of operations),
// scaled by float32(-c^2/2); n-1 from ne in float64 (full_solver.py:271-274) split hi/lo.
template <typename NE, bool PHASE>
__global__ void k_pack_from_ne(PackArgs A, const float *__restrict__ ne_nc, const NE *__restrict__ ne,
                               float4 *__restrict__ P, float *__restrict__ L) {
  const int64_t total = (int64_t)A.nx * A.ny * A.nz;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int i3[3];  // t counts the nodes with `a` fastest; q is where the node sits in the packed order
    i3[a] = (int)(t % A.na);
    i3[c] = (int)((t / A.na) % A.nc);
    i3[b] = (int)(t / ((int64_t)A.na * A.nc));
    const int64_t q = sr::node_index(i3[a], i3[b], i3[c], A.nb, A.nc);
    const int64_t sx = (int64_t)A.sny * A.snz, sy = A.snz;
    int j3[3] = {i3[0], i3[1], i3[2]}, ig[3] = {i3[0], i3[1], i3[2]};
    j3[a] += A.a_src_off;
    ig[a] += A.a_glob_off;
    const int64_t idx = j3[0] * sx + j3[1] * sy + j3[2];
    float gph[3];
    gph[0] = A.scale * grad1(ne_nc, idx, sx, ig[0], A.co[0]);
    gph[1] = A.scale * grad1(ne_nc, idx, sy, ig[1], A.co[1]);
    gph[2] = A.scale * grad1(ne_nc, idx, 1, ig[2], A.co[2]);
    float hi = 0.f, lo = 0.f;
    if (PHASE) {
      const double ne_cc = (double)ne[idx] * 1e-6;
      const double o_pe = 5.64e4 * sqrt(ne_cc);
      const double r = o_pe / A.omega;
      const double nref = sqrt(1.0 - r * r);
      split_hi_lo(nref - 1.0, hi, lo);
      L[q] = lo;
    }
    P[q] = make_float4(gph[b], gph[c], gph[a], hi);
  }
}

// same packing from fields the caller already computed (ScalarDomain.dndx/dndy/dndz, n_refrac())
template <bool PHASE>
__global__ void k_pack_from_fields(PackArgs A, const float *__restrict__ fx, const float *__restrict__ fy,
                                   const float *__restrict__ fz, const double *__restrict__ nref,
                                   float4 *__restrict__ P, float *__restrict__ L) {
  const int64_t total = (int64_t)A.nx * A.ny * A.nz;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int i3[3];  // t counts the nodes with `a` fastest; q is where the node sits in the packed order
    i3[a] = (int)(t % A.na);
    i3[c] = (int)((t / A.na) % A.nc);
    i3[b] = (int)(t / ((int64_t)A.na * A.nc));
    const int64_t q = sr::node_index(i3[a], i3[b], i3[c], A.nb, A.nc);
    const int64_t idx = ((int64_t)i3[0] * A.ny + i3[1]) * A.nz + i3[2];
    const float gph[3] = {fx[idx], fy[idx], fz[idx]};
    float hi = 0.f, lo = 0.f;
    if (PHASE) {
      split_hi_lo(nref[idx] - 1.0, hi, lo);
      L[q] = lo;
    }
    P[q] = make_float4(gph[b], gph[c], gph[a], hi);
  }
}

// packed -> native order, one component (0..2 physical x,y,z gradient) or n-1 (float64)
__global__ void k_unpack_f32(PackArgs A, const float4 *__restrict__ P, int comp_phys, float *__restrict__ out) {
  const int64_t total = (int64_t)A.nx * A.ny * A.nz;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  const int slot = comp_phys == b ? 0 : (comp_phys == c ? 1 : 2);
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int i3[3];  // t counts the nodes with `a` fastest; q is where the node sits in the packed order
    i3[a] = (int)(t % A.na);
    i3[c] = (int)((t / A.na) % A.nc);
    i3[b] = (int)(t / ((int64_t)A.na * A.nc));
    const int64_t q = sr::node_index(i3[a], i3[b], i3[c], A.nb, A.nc);
    const int64_t idx = ((int64_t)i3[0] * A.ny + i3[1]) * A.nz + i3[2];
    const float4 v = P[q];
    out[idx] = slot == 0 ? v.x : (slot == 1 ? v.y : v.z);
  }
}
__global__ void k_unpack_nm1(PackArgs A, const float4 *__restrict__ P, const float *__restrict__ L,
                             double *__restrict__ out) {
  const int64_t total = (int64_t)A.nx * A.ny * A.nz;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int i3[3];  // t counts the nodes with `a` fastest; q is where the node sits in the packed order
    i3[a] = (int)(t % A.na);
    i3[c] = (int)((t / A.na) % A.nc);
    i3[b] = (int)(t / ((int64_t)A.na * A.nc));
    const int64_t q = sr::node_index(i3[a], i3[b], i3[c], A.nb, A.nc);
    const int64_t idx = ((int64_t)i3[0] * A.ny + i3[1]) * A.nz + i3[2];
    out[idx] = (double)P[q].w + (double)L[q];
  }
}

// reference layout -> packed node order for the optional float64 fields
__global__ void k_pack_aux(PackArgs A, const double *__restrict__ kappa, const double *__restrict__ ne,
                           const double *__restrict__ B, double *__restrict__ K, double *__restrict__ Q) {
  const int64_t total = (int64_t)A.nx * A.ny * A.nz;
  const int a = A.axis, b = (a + 1) % 3, c = (a + 2) % 3;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    int i3[3];  // t counts the nodes with `a` fastest; q is where the node sits in the packed order
    i3[a] = (int)(t % A.na);
    i3[c] = (int)((t / A.na) % A.nc);
    i3[b] = (int)(t / ((int64_t)A.na * A.nc));
    const int64_t q = sr::node_index(i3[a], i3[b], i3[c], A.nb, A.nc);
    const int64_t idx = ((int64_t)i3[0] * A.ny + i3[1]) * A.nz + i3[2];
    if (K) K[q] = kappa[idx];
    if (Q) {
      Q[4 * q] = ne[idx];
      Q[4 * q + 1] = B[3 * idx];
      Q[4 * q + 2] = B[3 * idx + 1];
      Q[4 * q + 3] = B[3 * idx + 2];
    }
  }
}

struct HostCoef {
  std::vector<float> ca, cb, cc;
  float dx0 = 0, dxn = 0, two_dx = 0;
  int uniform = 1;
};

// numpy's coefficient arithmetic, in float32, evaluated once on the host
HostCoef make_coef(const float *x, int n) {
  HostCoef h;
  h.ca.assign(n, 0.f);
  h.cb.assign(n, 0.f);
  h.cc.assign(n, 0.f);
  std::vector<float> dx(n > 1 ? n - 1 : 1, 0.f);
  for (int i = 0; i + 1 < n; ++i) dx[i] = x[i + 1] - x[i];
  for (int i = 1; i + 1 < n; ++i)
    if (dx[i] != dx[0]) h.uniform = 0;
  for (int i = 1; i + 1 < n; ++i) {
    const float dx1 = dx[i - 1], dx2 = dx[i];
    h.ca[i] = -(dx2) / (dx1 * (dx1 + dx2));
    h.cb[i] = (dx2 - dx1) / (dx1 * dx2);
    h.cc[i] = dx1 / (dx2 * (dx1 + dx2));
  }
  h.dx0 = dx[0];
  h.dxn = dx[n > 1 ? n - 2 : 0];
  h.two_dx = 2.0f * dx[0];
  return h;
}

int check_axis(const char *name, const float *x, int n) {
  SR_CHECK(x != nullptr, "%s coordinates are NULL", name);
  SR_CHECK(n >= 2, "%s needs at least 2 nodes (np.gradient needs edge_order+1)", name);
  for (int i = 0; i + 1 < n; ++i)
    SR_CHECK(x[i + 1] > x[i], "%s coordinates must be strictly ascending (node %d)", name, i);
  return SR_OK;
}

int volume_common(sr_volume *v, int nx, int ny, int nz, const float *x, const float *y, const float *z,
                  int probing_axis, int flags, double omega) {
  const int dims[3] = {nx, ny, nz};
  const float *co[3] = {x, y, z};
  const int a = probing_axis, b = (a + 1) % 3, c = (a + 2) % 3;
  v->nx = nx;
  v->ny = ny;
  v->nz = nz;
  v->axis = a;
  v->na = dims[a];
  v->nb = dims[b];
  v->nc = dims[c];
  v->flags = flags;
  v->omega = omega;
  v->k_lo = 0;
  v->k_hi = dims[a] - 1;
  v->n_glob = dims[a];
  const int order[3] = {a, b, c};
  for (int k = 0; k < 3; ++k) {
    const int n = dims[order[k]];
    v->hg[k].resize(n);
    std::vector<double> rg(n, 0.0);
    for (int i = 0; i < n; ++i) v->hg[k][i] = (double)co[order[k]][i];  // float32 node -> float64, as scipy
    for (int i = 0; i + 1 < n; ++i) rg[i] = 1.0 / (v->hg[k][i + 1] - v->hg[k][i]);
    int rc = sr::dev_alloc(&v->g[k], (size_t)n);
    if (rc) return rc;
    rc = sr::dev_alloc(&v->rg[k], (size_t)n);
    if (rc) return rc;
    SR_HIP(hipMemcpy(v->g[k], v->hg[k].data(), sizeof(double) * n, hipMemcpyHostToDevice));
    SR_HIP(hipMemcpy(v->rg[k], rg.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  }
  const size_t packed = sr::packed_nodes(v->na, v->nb, v->nc);  // whole octets: the padding planes stay zero
  int rc = sr::dev_alloc(&v->P, packed);
  if (rc) return rc;
  SR_HIP(hipMemsetAsync(v->P, 0, packed * sizeof(float4), sr::ctx().stream));  // the library's stream: ordered before the pack kernel
  if (flags & SR_VOL_PHASE) {
    rc = sr::dev_alloc(&v->L, packed);
    if (rc) return rc;
    SR_HIP(hipMemsetAsync(v->L, 0, packed * sizeof(float), sr::ctx().stream));
  }
  return SR_OK;
}

PackArgs pack_args(const sr_volume *v) {
  PackArgs A{};
  A.nx = v->nx;
  A.ny = v->ny;
  A.nz = v->nz;
  A.axis = v->axis;
  A.na = v->na;
  A.nb = v->nb;
  A.nc = v->nc;
  A.sny = v->ny;
  A.snz = v->nz;
  A.a_src_off = 0;
  A.a_glob_off = 0;
  A.omega = v->omega;
  A.scale = (float)(-0.5 * (sr::kC * sr::kC));
  return A;
}

}  // namespace

extern "C" {

void sr_volume_destroy(sr_volume *v) {
  if (!v) return;
  sr::dev_free(v->P);
  sr::dev_free(v->L);
  sr::dev_free(v->K);
  sr::dev_free(v->Q);
  sr::dev_free(v->R);
  for (int k = 0; k < 3; ++k) {
    sr::dev_free(v->g[k]);
    sr::dev_free(v->rg[k]);
  }
  for (auto &t : v->step_tabs) sr::dev_free(t.second);
  delete v;
}

}  // extern "C"

namespace {

// nx, ny, nz, x, y, z: the whole domain; the volume built holds node planes k_lo..k_hi of the probing axis and `ne`
// holds planes max(k_lo-1, 0)..min(k_hi+1, n-1) (the whole array when k_lo = 0 and k_hi = n-1)
int create_impl(sr_volume **out, const void *ne, int ne_is_f64, int nx, int ny, int nz, const float *x, const float *y,
                const float *z, double lwl, int probing_axis, int flags, int k_lo, int k_hi, bool whole) {
  SR_CHECK(out != nullptr && ne != nullptr, "sr_volume_create: NULL argument");
  *out = nullptr;
  SR_CHECK(probing_axis >= 0 && probing_axis <= 2, "probing_axis must be 0 (x), 1 (y) or 2 (z), got %d", probing_axis);
  SR_CHECK(lwl > 0, "laser wavelength must be positive");
  int rc = check_axis("x", x, nx);
  if (rc) return rc;
  rc = check_axis("y", y, ny);
  if (rc) return rc;
  rc = check_axis("z", z, nz);
  if (rc) return rc;
  rc = sr::ensure_init();
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;

  const double omega = (2.0 * M_PI) * (sr::kC / lwl);  // full_solver.py:218
  const double ncrit = 3.14207787e-4 * (omega * omega);  // :219
  const int gdims[3] = {nx, ny, nz};
  const int a_ax = probing_axis, n_a = gdims[a_ax];
  if (whole) {
    k_lo = 0;
    k_hi = n_a - 1;
  }
  SR_CHECK(k_lo >= 0 && k_hi < n_a && k_hi > k_lo, "slab planes %d..%d out of range for %d node planes", k_lo, k_hi, n_a);
  const int h_lo = k_lo > 0 ? k_lo - 1 : 0, h_hi = k_hi < n_a - 1 ? k_hi + 1 : n_a - 1;
  int sdims[3] = {nx, ny, nz}, pdims[3] = {nx, ny, nz};  // source (with halo) and packed extents
  sdims[a_ax] = h_hi - h_lo + 1;
  pdims[a_ax] = k_hi - k_lo + 1;
  const float *pco[3] = {x, y, z};
  pco[a_ax] += k_lo;
  sr_volume *v = new sr_volume();
  rc = volume_common(v, pdims[0], pdims[1], pdims[2], pco[0], pco[1], pco[2], probing_axis, flags & SR_VOL_PHASE, omega);
  if (rc) {
    sr_volume_destroy(v);
    return rc;
  }
  v->is_slab = !whole;
  v->k_lo = k_lo;
  v->k_hi = k_hi;
  v->n_glob = n_a;
  const size_t src_total = (size_t)sdims[0] * sdims[1] * sdims[2];
  void *d_ne = nullptr;
  float *d_nenc = nullptr, *d_coef = nullptr;
  const size_t ne_bytes = src_total * (ne_is_f64 ? sizeof(double) : sizeof(float));
  auto cleanup = [&]() {
    sr::dev_free(d_ne);
    sr::dev_free(d_nenc);
    sr::dev_free(d_coef);
  };
#define SR_TRY(call)                                                                                     \
  do {                                                                                                   \
    hipError_t e_ = (call);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      cleanup();                                                                                         \
      sr_volume_destroy(v);                                                                              \
      return sr::fail(SR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                    \
  } while (0)
  SR_TRY(hipMalloc(&d_ne, ne_bytes));
  SR_TRY(hipMalloc(reinterpret_cast<void **>(&d_nenc), src_total * sizeof(float)));
  SR_TRY(hipMemcpyAsync(d_ne, ne, ne_bytes, hipMemcpyHostToDevice, st));

  // gradient coefficients (float32, numpy's arithmetic) for x, y, z
  const int dims[3] = {nx, ny, nz};
  const float *co[3] = {x, y, z};
  HostCoef hc[3] = {make_coef(x, nx), make_coef(y, ny), make_coef(z, nz)};
  std::vector<float> flat;
  size_t off[3][3];
  for (int k = 0; k < 3; ++k) {
    off[k][0] = flat.size();
    flat.insert(flat.end(), hc[k].ca.begin(), hc[k].ca.end());
    off[k][1] = flat.size();
    flat.insert(flat.end(), hc[k].cb.begin(), hc[k].cb.end());
    off[k][2] = flat.size();
    flat.insert(flat.end(), hc[k].cc.begin(), hc[k].cc.end());
  }
  (void)co;
  SR_TRY(hipMalloc(reinterpret_cast<void **>(&d_coef), flat.size() * sizeof(float)));
  SR_TRY(hipMemcpyAsync(d_coef, flat.data(), flat.size() * sizeof(float), hipMemcpyHostToDevice, st));
  PackArgs A = pack_args(v);
  A.sny = sdims[1];
  A.snz = sdims[2];
  A.a_src_off = k_lo - h_lo;
  A.a_glob_off = k_lo;
  for (int k = 0; k < 3; ++k) {
    A.co[k].ca = d_coef + off[k][0];
    A.co[k].cb = d_coef + off[k][1];
    A.co[k].cc = d_coef + off[k][2];
    A.co[k].dx0 = hc[k].dx0;
    A.co[k].dxn = hc[k].dxn;
    A.co[k].two_dx = hc[k].two_dx;
    A.co[k].n = dims[k];
    A.co[k].uniform = hc[k].uniform;
  }
  const int block = 256;
  const unsigned grid = (unsigned)std::min<int64_t>((int64_t)(src_total + block - 1) / block, (int64_t)sr::ctx().n_cu * 32);
  if (ne_is_f64)
    hipLaunchKernelGGL(k_ne_nc_f64, dim3(grid), dim3(block), 0, st, (const double *)d_ne, (int64_t)src_total, ncrit, d_nenc);
  else
    hipLaunchKernelGGL(k_ne_nc_f32, dim3(grid), dim3(block), 0, st, (const float *)d_ne, (int64_t)src_total, (float)ncrit, d_nenc);
  const bool phase = (flags & SR_VOL_PHASE) != 0;
  if (ne_is_f64) {
    if (phase)
      hipLaunchKernelGGL((k_pack_from_ne<double, true>), dim3(grid), dim3(block), 0, st, A, d_nenc, (const double *)d_ne, v->P, v->L);
    else
      hipLaunchKernelGGL((k_pack_from_ne<double, false>), dim3(grid), dim3(block), 0, st, A, d_nenc, (const double *)d_ne, v->P, v->L);
  } else {
    if (phase)
      hipLaunchKernelGGL((k_pack_from_ne<float, true>), dim3(grid), dim3(block), 0, st, A, d_nenc, (const float *)d_ne, v->P, v->L);
    else
      hipLaunchKernelGGL((k_pack_from_ne<float, false>), dim3(grid), dim3(block), 0, st, A, d_nenc, (const float *)d_ne, v->P, v->L);
  }
  SR_TRY(hipGetLastError());
  SR_TRY(hipStreamSynchronize(st));
#undef SR_TRY
  cleanup();
  *out = v;
  return SR_OK;
}

}  // namespace

extern "C" {

int sr_volume_create(sr_volume **out, const void *ne, int ne_is_f64, int nx, int ny, int nz, const float *x,
                     const float *y, const float *z, double lwl, int probing_axis, int flags) {
  return create_impl(out, ne, ne_is_f64, nx, ny, nz, x, y, z, lwl, probing_axis, flags, 0, 0, true);
}

int sr_volume_create_slab(sr_volume **out, const void *ne_slab, int ne_is_f64, int nx, int ny, int nz, const float *x,
                          const float *y, const float *z, double lwl, int probing_axis, int flags, int k_lo, int k_hi) {
  return create_impl(out, ne_slab, ne_is_f64, nx, ny, nz, x, y, z, lwl, probing_axis, flags, k_lo, k_hi, false);
}

int sr_volume_create_from_fields(sr_volume **out, const float *dndx, const float *dndy, const float *dndz,
                                 const double *nref, double omega, int nx, int ny, int nz, const float *x,
                                 const float *y, const float *z, int probing_axis) {
  SR_CHECK(out && dndx && dndy && dndz, "sr_volume_create_from_fields: NULL argument");
  *out = nullptr;
  SR_CHECK(probing_axis >= 0 && probing_axis <= 2, "probing_axis must be 0, 1 or 2, got %d", probing_axis);
  int rc = check_axis("x", x, nx);
  if (rc) return rc;
  rc = check_axis("y", y, ny);
  if (rc) return rc;
  rc = check_axis("z", z, nz);
  if (rc) return rc;
  rc = sr::ensure_init();
  if (rc) return rc;
  hipStream_t st = sr::ctx().stream;
  sr_volume *v = new sr_volume();
  rc = volume_common(v, nx, ny, nz, x, y, z, probing_axis, nref ? SR_VOL_PHASE : 0, omega);
  if (rc) {
    sr_volume_destroy(v);
    return rc;
  }
  const size_t total = (size_t)nx * ny * nz;
  float *d_f[3] = {nullptr, nullptr, nullptr};
  double *d_n = nullptr;
  auto cleanup = [&]() {
    for (auto p : d_f) sr::dev_free(p);
    sr::dev_free(d_n);
  };
  const float *src[3] = {dndx, dndy, dndz};
  hipError_t e = hipSuccess;
  for (int k = 0; k < 3 && e == hipSuccess; ++k) {
    e = hipMalloc(reinterpret_cast<void **>(&d_f[k]), total * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_f[k], src[k], total * sizeof(float), hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess && nref) {
    e = hipMalloc(reinterpret_cast<void **>(&d_n), total * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_n, nref, total * sizeof(double), hipMemcpyHostToDevice, st);
  }
  if (e == hipSuccess) {
    PackArgs A = pack_args(v);
    const int block = 256;
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)(total + block - 1) / block, (int64_t)sr::ctx().n_cu * 32);
    if (nref)
      hipLaunchKernelGGL((k_pack_from_fields<true>), dim3(grid), dim3(block), 0, st, A, d_f[0], d_f[1], d_f[2], d_n, v->P, v->L);
    else
      hipLaunchKernelGGL((k_pack_from_fields<false>), dim3(grid), dim3(block), 0, st, A, d_f[0], d_f[1], d_f[2], d_n, v->P, v->L);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  cleanup();
  if (e != hipSuccess) {
    sr_volume_destroy(v);
    return sr::fail(SR_ERR_HIP, "sr_volume_create_from_fields: %s", hipGetErrorString(e));
  }
  *out = v;
  return SR_OK;
}

int sr_volume_fields(const sr_volume *v, float *dndx, float *dndy, float *dndz, double *nref_minus_1) {
  SR_CHECK(v != nullptr, "sr_volume_fields: NULL volume");
  hipStream_t st = sr::ctx().stream;
  const size_t total = (size_t)v->nx * v->ny * v->nz;
  PackArgs A = pack_args(v);
  const int block = 256;
  const unsigned grid = (unsigned)std::min<int64_t>((int64_t)(total + block - 1) / block, (int64_t)sr::ctx().n_cu * 32);
  float *outs[3] = {dndx, dndy, dndz};
  for (int k = 0; k < 3; ++k) {
    if (!outs[k]) continue;
    float *tmp = nullptr;
    int rc = sr::dev_alloc(&tmp, total);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_f32, dim3(grid), dim3(block), 0, st, A, v->P, k, tmp);
    hipError_t e = hipMemcpyAsync(outs[k], tmp, total * sizeof(float), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    sr::dev_free(tmp);
    if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_volume_fields: %s", hipGetErrorString(e));
  }
  if (nref_minus_1) {
    SR_CHECK(v->L != nullptr, "sr_volume_fields: volume was created without SR_VOL_PHASE");
    double *tmp = nullptr;
    int rc = sr::dev_alloc(&tmp, total);
    if (rc) return rc;
    hipLaunchKernelGGL(k_unpack_nm1, dim3(grid), dim3(block), 0, st, A, v->P, v->L, tmp);
    hipError_t e = hipMemcpyAsync(nref_minus_1, tmp, total * sizeof(double), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    sr::dev_free(tmp);
    if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_volume_fields: %s", hipGetErrorString(e));
  }
  return SR_OK;
}

double sr_volume_omega(const sr_volume *v) { return v ? v->omega : 0.0; }

int64_t sr_volume_bytes(const sr_volume *v) {
  if (!v) return 0;
  const int64_t total = (int64_t)sr::packed_nodes(v->na, v->nb, v->nc);
  const int64_t recs = v->R ? (int64_t)v->na * (v->nb - 1) * (v->nc - 1) * 16 * (int64_t)sizeof(double) : 0;  // the tile path's ready-made records
  return recs + total * (int64_t)(sizeof(float4) + (v->L ? sizeof(float) : 0) + (v->K ? sizeof(double) : 0) + (v->Q ? 4 * sizeof(double) : 0));
}

int sr_volume_attach_aux(sr_volume *v, const double *kappa, const double *ne, const double *B, double verdet) {
  SR_CHECK(v != nullptr, "sr_volume_attach_aux: NULL volume");
  SR_CHECK((ne == nullptr) == (B == nullptr), "sr_volume_attach_aux: ne and B go together (both or neither)");
  SR_CHECK(kappa || ne, "sr_volume_attach_aux: nothing to attach");
  hipStream_t st = sr::ctx().stream;
  const size_t total = (size_t)v->nx * v->ny * v->nz;
  const size_t packed = sr::packed_nodes(v->na, v->nb, v->nc);
  double *d_k = nullptr, *d_ne = nullptr, *d_B = nullptr;
  auto cleanup = [&]() {
    sr::dev_free(d_k);
    sr::dev_free(d_ne);
    sr::dev_free(d_B);
  };
  sr::dev_free(v->K);
  sr::dev_free(v->Q);
  v->K = v->Q = nullptr;
  hipError_t e = hipSuccess;
  if (kappa) {
    e = hipMalloc(reinterpret_cast<void **>(&d_k), total * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_k, kappa, total * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&v->K), packed * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(v->K, 0, packed * sizeof(double), st);
  }
  if (e == hipSuccess && ne) {
    e = hipMalloc(reinterpret_cast<void **>(&d_ne), total * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_ne, ne, total * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_B), 3 * total * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(d_B, B, 3 * total * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&v->Q), 4 * packed * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(v->Q, 0, 4 * packed * sizeof(double), st);
  }
  if (e == hipSuccess) {
    const int block = 256;
    const unsigned grid = (unsigned)std::min<int64_t>((int64_t)(total + block - 1) / block, (int64_t)sr::ctx().n_cu * 32);
    hipLaunchKernelGGL(k_pack_aux, dim3(grid), dim3(block), 0, st, pack_args(v), (const double *)d_k, (const double *)d_ne,
                       (const double *)d_B, v->K, v->Q);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
  }
  cleanup();
  if (e != hipSuccess) {
    sr::dev_free(v->K);
    sr::dev_free(v->Q);
    v->K = v->Q = nullptr;
      return sr::fail(SR_ERR_HIP, "sr_volume_attach_aux: %s", hipGetErrorString(e));
  }
  v->verdet = verdet;
  return SR_OK;
}

}  // extern "C"
