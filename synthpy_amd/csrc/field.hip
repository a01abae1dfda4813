// The step before the path: volume synthesis (gaussian3D.domain_fft, src/field_generator/gaussian3D.py:215-271).
// The reference shapes complex Gaussian noise with sqrt(S(k)) and takes np.fft.ifftn(...).real / max|.| on the host
// (15 s of the 25 s a 512^3 volume takes there).  Here the noise and the float32 amplitude sqrt(S) arrive from the
// host (the noise must come from the caller's seeded np.random stream to reproduce the reference's field), the
// product, the 3-D inverse FFT (hipFFT Z2Z, bound at first use like RCCL) and the normalisation run on the GPU.
#include <dlfcn.h>
#include <hipfft/hipfft.h>

#include "common.hpp"

namespace {

struct Fft {
  void *h = nullptr;
  hipfftResult (*Plan3d)(hipfftHandle *, int, int, int, hipfftType) = nullptr;
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
    SR_SYM(SetStream, "hipfftSetStream")
    SR_SYM(ExecZ2Z, "hipfftExecZ2Z")
    SR_SYM(Destroy, "hipfftDestroy")
#undef SR_SYM
  }
  *out = &F;
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

}  // namespace

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
    if (plan) F->Destroy(plan);
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
  SR_TRY_FFT(F->Plan3d(&plan, n0, n1, n2, HIPFFT_Z2Z));  // C order: n2 fastest, as the NumPy array
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
