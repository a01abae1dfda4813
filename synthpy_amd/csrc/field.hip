// The step before the path: volume synthesis (gaussian3D.domain_fft, src/field_generator/gaussian3D.py:215-271).
// The reference shapes complex Gaussian noise with sqrt(S(k)) and takes np.fft.ifftn(...).real / max|.| on the host
// (15 s of the 25 s a 512^3 volume takes there).  Here the noise and the float32 amplitude sqrt(S) arrive from the
// host (the noise must come from the caller's seeded np.random stream to reproduce the reference's field), the
// product, the 3-D inverse FFT (hipFFT Z2Z, bound at first use like RCCL) and the normalisation run on the GPU.
#include <dlfcn.h>
#include <hipfft/hipfft.h>

#include <map>
#include <tuple>

#include "common.hpp"

namespace {

struct Fft {
  void *h = nullptr;
  hipfftResult (*Plan3d)(hipfftHandle *, int, int, int, hipfftType) = nullptr;
  hipfftResult (*Plan2d)(hipfftHandle *, int, int, hipfftType) = nullptr;
  hipfftResult (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
  hipfftResult (*ExecZ2Z)(hipfftHandle, hipfftDoubleComplex *, hipfftDoubleComplex *, int) = nullptr;
  hipfftResult (*Destroy)(hipfftHandle) = nullptr;
};

int fft_lib(Fft **out) {
  static Fft F;
  if (!F.h) {
    const char *names[] = {"libhipfft.so", "libhipfft.so.0", "/opt/rocm/lib/libhipfft.so"};
    for (const char *n : names) {
      F.h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (F.h) break;
    }
    if (!F.h) return sr::fail(SR_ERR_HIP, "cannot load libhipfft.so: %s", dlerror());
#define SR_SYM(field, name)                                                   \
  F.field = reinterpret_cast<decltype(F.field)>(dlsym(F.h, name));            \
  if (!F.field) return sr::fail(SR_ERR_HIP, "libhipfft.so lacks symbol %s", name);
    SR_SYM(Plan3d, "hipfftPlan3d")
    SR_SYM(Plan2d, "hipfftPlan2d")
    SR_SYM(SetStream, "hipfftSetStream")
    SR_SYM(ExecZ2Z, "hipfftExecZ2Z")
    SR_SYM(Destroy, "hipfftDestroy")
#undef SR_SYM
  }
  *out = &F;
  return SR_OK;
}

// Plans are kept for the life of the process: creating one costs 1-2 s (rocFFT builds its kernels at run time), using
// it milliseconds.  Key: (n0, n1, n2) with n2 = 0 for a 2-D plan.
int fft_plan(Fft *F, int n0, int n1, int n2, hipfftHandle *out) {
  static std::map<std::tuple<int, int, int>, hipfftHandle> cache;
  const auto key = std::make_tuple(n0, n1, n2);
  auto it = cache.find(key);
  if (it == cache.end()) {
    hipfftHandle plan = nullptr;
    const hipfftResult r = n2 > 0 ? F->Plan3d(&plan, n0, n1, n2, HIPFFT_Z2Z) : F->Plan2d(&plan, n0, n1, HIPFFT_Z2Z);
    if (r != HIPFFT_SUCCESS) return sr::fail(SR_ERR_HIP, "hipfftPlan%dd(%d, %d, %d) failed: hipfftResult %d", n2 > 0 ? 3 : 2, n0, n1, n2, (int)r);
    it = cache.emplace(key, plan).first;
  }
  *out = it->second;
  return SR_OK;
}

// fft_field = noise * sqrt(S): complex128 times float32 promoted to float64, as numpy does
__global__ void k_shape_noise(double2 *__restrict__ w, const float *__restrict__ amp, int64_t n) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = (double)amp[i];
    double2 v = w[i];
    v.x *= a;
    v.y *= a;
    w[i] = v;
  }
}

// field = Re(ifft) = Re(unnormalised backward transform) * (1/N); max |field| by one atomic per wavefront
__global__ void k_real_scaled(const double2 *__restrict__ w, int64_t n, double inv_n, double *__restrict__ out,
                              unsigned long long *__restrict__ vmax_bits) {
  double m = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = w[i].x * inv_n;
    out[i] = v;
    const double a = fabs(v);
    m = a > m ? a : m;  // NaN never wins: an all-NaN field keeps 0
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(vmax_bits, (unsigned long long)__double_as_longlong(m));  // non-negative doubles order as integers
}

__global__ void k_divide(double *__restrict__ f, int64_t n, const unsigned long long *__restrict__ vmax_bits) {
  const double m = __longlong_as_double((long long)*vmax_bits);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) f[i] = f[i] / m;
}

__global__ void k_to_complex(const double *__restrict__ r, int64_t n, double2 *__restrict__ w) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) w[i] = make_double2(r[i], 0.0);
}

// |F|^2 / (n0*n1)^2 summed and counted per radial bin [edges[b], edges[b+1])  (power_spectrum.py:395-413)
__global__ void k_radial_bins(const double2 *__restrict__ F, int n0, int n1, const double *__restrict__ k0,
                              const double *__restrict__ k1, const double *__restrict__ edges, int n_edges, double norm,
                              double *__restrict__ sum, unsigned long long *__restrict__ cnt) {
  // the bins are few (99 in the reference's call): per-workgroup sums in LDS, one global atomic per bin and workgroup
  constexpr int kMaxBins = 512;
  __shared__ double lsum[kMaxBins];
  __shared__ unsigned long long lcnt[kMaxBins];
  const bool in_lds = n_edges - 1 <= kMaxBins;
  for (int t = threadIdx.x; t < kMaxBins; t += blockDim.x) {
    lsum[t] = 0.0;
    lcnt[t] = 0ull;
  }
  __syncthreads();
  const int64_t n = (int64_t)n0 * n1;
  for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < n; q += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(q / n1), j = (int)(q % n1);
    const double k = sqrt(k0[i] * k0[i] + k1[j] * k1[j]);
    if (!(k >= edges[0] && k < edges[n_edges - 1])) continue;
    int lo = 0, hi = n_edges - 1;  // edges[lo] <= k < edges[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (k < edges[mid])
        hi = mid;
      else
        lo = mid;
    }
    const double2 v = F[q];
    const double pw = (v.x * v.x + v.y * v.y) / norm;
    if (in_lds) {
      atomicAdd(&lsum[lo], pw);
      atomicAdd(&lcnt[lo], 1ull);
    } else {
      atomicAdd(&sum[lo], pw);
      atomicAdd(&cnt[lo], 1ull);
    }
  }
  __syncthreads();
  if (in_lds)
    for (int t = threadIdx.x; t < n_edges - 1; t += blockDim.x)
      if (lcnt[t]) {
        atomicAdd(&sum[t], lsum[t]);
        atomicAdd(&cnt[t], lcnt[t]);
      }
}

}  // namespace

// After the path: radially binned 2-D power spectrum of a detector image (radial_2Dspectrum,
// src/utils/power_spectrum.py:372-421): |fft2(img)|^2/(n0*n1)^2 averaged over the wavenumber bins [edges[b], edges[b+1]).
// k0 (n0) and k1 (n1) give the wavenumber of every index of the UNSHIFTED transform along each axis; sum and count
// (n_edges-1 each) come back, the caller divides (an empty bin is NaN there, as np.mean of nothing).
extern "C" int sr_radial_spectrum2d(const double *img, int n0, int n1, const double *k0, const double *k1,
                                    const double *edges, int n_edges, double *sum, uint64_t *count) {
  SR_CHECK(img && k0 && k1 && edges && sum && count, "sr_radial_spectrum2d: NULL argument");
  SR_CHECK(n0 > 0 && n1 > 0 && n_edges >= 2, "sr_radial_spectrum2d: bad sizes");
  int rc = sr::ensure_init();
  if (rc) return rc;
  Fft *F;
  if ((rc = fft_lib(&F))) return rc;
  hipStream_t st = sr::ctx().stream;
  const int64_t n = (int64_t)n0 * n1;
  const int nb = n_edges - 1;
  double *d_r = nullptr, *d_k = nullptr, *d_sum = nullptr;
  double2 *d_w = nullptr;
  unsigned long long *d_cnt = nullptr;
  hipfftHandle plan = nullptr;
  auto cleanup = [&]() {
    sr::dev_free(d_r);
    sr::dev_free(d_k);
    sr::dev_free(d_sum);
    sr::dev_free(d_w);
    sr::dev_free(d_cnt);
  };
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_r), sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_w), sizeof(double2) * n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_k), sizeof(double) * (size_t)(n0 + n1 + n_edges));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_sum), sizeof(double) * nb);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_cnt), sizeof(unsigned long long) * nb);
  if (e == hipSuccess) e = hipMemcpyAsync(d_r, img, sizeof(double) * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_k, k0, sizeof(double) * n0, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_k + n0, k1, sizeof(double) * n1, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemcpyAsync(d_k + n0 + n1, edges, sizeof(double) * n_edges, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipMemsetAsync(d_sum, 0, sizeof(double) * nb, st);
  if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long) * nb, st);
  if (e != hipSuccess) {
    cleanup();
    return sr::fail(SR_ERR_HIP, "sr_radial_spectrum2d: %s", hipGetErrorString(e));
  }
  const int block = 256;
  const unsigned grid = (unsigned)std::min<int64_t>((n + block - 1) / block, (int64_t)sr::ctx().n_cu * 32);
  hipLaunchKernelGGL(k_to_complex, dim3(grid), dim3(block), 0, st, (const double *)d_r, n, d_w);
  if ((rc = fft_plan(F, n0, n1, 0, &plan))) {
    cleanup();
    return rc;
  }
  hipfftResult fr = F->SetStream(plan, st);
  if (fr == HIPFFT_SUCCESS) fr = F->ExecZ2Z(plan, d_w, d_w, HIPFFT_FORWARD);
  if (fr != HIPFFT_SUCCESS) {
    cleanup();
    return sr::fail(SR_ERR_HIP, "sr_radial_spectrum2d: hipfftResult %d", (int)fr);
  }
  const double nn = (double)n0 * (double)n1;
  const unsigned bgrid = std::min<unsigned>(grid, (unsigned)sr::ctx().n_cu * 4);
  hipLaunchKernelGGL(k_radial_bins, dim3(bgrid), dim3(block), 0, st, (const double2 *)d_w, n0, n1, (const double *)d_k,
                     (const double *)(d_k + n0), (const double *)(d_k + n0 + n1), n_edges, nn * nn, d_sum, d_cnt);
  e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(sum, d_sum, sizeof(double) * nb, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(count, d_cnt, sizeof(uint64_t) * nb, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  if (e != hipSuccess) return sr::fail(SR_ERR_HIP, "sr_radial_spectrum2d: %s", hipGetErrorString(e));
  return SR_OK;
}

extern "C" int sr_field_ifft_real(const double *noise, const float *amp, int n0, int n1, int n2, int normalise, double *out) {
  SR_CHECK(noise && amp && out, "sr_field_ifft_real: NULL argument");
  SR_CHECK(n0 > 0 && n1 > 0 && n2 > 0, "sr_field_ifft_real: bad shape (%d, %d, %d)", n0, n1, n2);
  int rc = sr::ensure_init();
  if (rc) return rc;
  Fft *F;
  if ((rc = fft_lib(&F))) return rc;
  hipStream_t st = sr::ctx().stream;
  const int64_t n = (int64_t)n0 * n1 * n2;
  double2 *d_w = nullptr;
  float *d_a = nullptr;
  double *d_f = nullptr;
  unsigned long long *d_m = nullptr;
  hipfftHandle plan = nullptr;
  auto cleanup = [&]() {
    sr::dev_free(d_w);
    sr::dev_free(d_a);
    sr::dev_free(d_f);
    sr::dev_free(d_m);
  };
#define SR_TRY(call)                                                                                          \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      cleanup();                                                                                              \
      return sr::fail(SR_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    }                                                                                                         \
  } while (0)
#define SR_TRY_FFT(call)                                                                \
  do {                                                                                  \
    hipfftResult r_ = (call);                                                           \
    if (r_ != HIPFFT_SUCCESS) {                                                         \
      cleanup();                                                                        \
      return sr::fail(SR_ERR_HIP, "%s failed: hipfftResult %d", #call, (int)r_);        \
    }                                                                                   \
  } while (0)
  SR_TRY(hipMalloc(reinterpret_cast<void **>(&d_w), sizeof(double2) * n));
  SR_TRY(hipMalloc(reinterpret_cast<void **>(&d_a), sizeof(float) * n));
  SR_TRY(hipMalloc(reinterpret_cast<void **>(&d_m), sizeof(unsigned long long)));
  SR_TRY(hipMemcpyAsync(d_w, noise, sizeof(double2) * n, hipMemcpyHostToDevice, st));
  SR_TRY(hipMemcpyAsync(d_a, amp, sizeof(float) * n, hipMemcpyHostToDevice, st));
  SR_TRY(hipMemsetAsync(d_m, 0, sizeof(unsigned long long), st));
  const int block = 256;
  const unsigned grid = (unsigned)std::min<int64_t>((n + block - 1) / block, (int64_t)sr::ctx().n_cu * 32);
  hipLaunchKernelGGL(k_shape_noise, dim3(grid), dim3(block), 0, st, d_w, (const float *)d_a, n);
  SR_TRY(hipGetLastError());
  if ((rc = fft_plan(F, n0, n1, n2, &plan))) {  // C order: n2 fastest, as the NumPy array
    cleanup();
    return rc;
  }
  SR_TRY_FFT(F->SetStream(plan, st));
  SR_TRY_FFT(F->ExecZ2Z(plan, d_w, d_w, HIPFFT_BACKWARD));
  sr::dev_free(d_a);
  d_a = nullptr;
  SR_TRY(hipMalloc(reinterpret_cast<void **>(&d_f), sizeof(double) * n));
  hipLaunchKernelGGL(k_real_scaled, dim3(grid), dim3(block), 0, st, (const double2 *)d_w, n, 1.0 / (double)n, d_f, d_m);
  if (normalise) hipLaunchKernelGGL(k_divide, dim3(grid), dim3(block), 0, st, d_f, n, (const unsigned long long *)d_m);
  SR_TRY(hipGetLastError());
  SR_TRY(hipMemcpyAsync(out, d_f, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  SR_TRY(hipStreamSynchronize(st));
#undef SR_TRY
#undef SR_TRY_FFT
  cleanup();
  return SR_OK;
}
