// Dependent-issue latency of the float64 VALU instructions on gfx950 (MI355X): cycles per wave64 instruction per SIMD when
// the instruction stream of a wave consists of C independent dependency chains (C = 1, 2, 4, 8), at 1 / 2 / 4 waves per SIMD.
// tools/valu_issue.hip measures the issue cost with 8 chains (nothing ever waits); the tracer's stage arithmetic is not
// that parallel, and k_trace_f64 runs at 2 waves per SIMD: this says how many independent instructions a wave needs
// between a result and its use for the SIMD to stay busy.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_latency.hip -o /tmp/valu_latency && /tmp/valu_latency > profiles/r02_valu_latency.json
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

struct Stamp {
  unsigned long long t0, t1;
};

#define X8(a, b, c, d, e, f, g, h) a b c d e f g h a b c d e f g h a b c d e f g h a b c d e f g h a b c d e f g h a b c d e f g h a b c d e f g h a b c d e f g h

// OP3: D = D*a + b (three sources); OP2: D = D op a; OP1: D = f(D)
#define I3(OP, D) OP " " D ", " D ", %8, %9\n"
#define I2(OP, D) OP " " D ", " D ", %8\n"
#define I1(OP, D) OP " " D ", " D "\n"
#define CH1(I, OP) X8(I(OP, "%0"), I(OP, "%0"), I(OP, "%0"), I(OP, "%0"), I(OP, "%0"), I(OP, "%0"), I(OP, "%0"), I(OP, "%0"))
#define CH2(I, OP) X8(I(OP, "%0"), I(OP, "%1"), I(OP, "%0"), I(OP, "%1"), I(OP, "%0"), I(OP, "%1"), I(OP, "%0"), I(OP, "%1"))
#define CH4(I, OP) X8(I(OP, "%0"), I(OP, "%1"), I(OP, "%2"), I(OP, "%3"), I(OP, "%0"), I(OP, "%1"), I(OP, "%2"), I(OP, "%3"))
#define CH8(I, OP) X8(I(OP, "%0"), I(OP, "%1"), I(OP, "%2"), I(OP, "%3"), I(OP, "%4"), I(OP, "%5"), I(OP, "%6"), I(OP, "%7"))

#define KERNEL(NAME, DT, ST, BODY)                                                                       \
  __global__ __launch_bounds__(1024) void NAME(Stamp *out, int iters, float seed) {                     \
    DT d0 = (DT)seed + 1, d1 = d0, d2 = d0, d3 = d0, d4 = d0, d5 = d0, d6 = d0, d7 = d0;                 \
    ST a = (ST)seed + 1, b = a;                                                                          \
    __syncthreads();                                                                                     \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
    for (int it = 0; it < iters; ++it)                                                                   \
      asm volatile(BODY : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) \
                   : "v"(a), "v"(b));                                                                    \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
    asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));           \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = Stamp{t0, t1};                                  \
  }

#define FOUR(N, DT, ST, I, OP)        \
  KERNEL(N##_c1, DT, ST, CH1(I, OP)) \
  KERNEL(N##_c2, DT, ST, CH2(I, OP)) \
  KERNEL(N##_c4, DT, ST, CH4(I, OP)) \
  KERNEL(N##_c8, DT, ST, CH8(I, OP))

FOUR(k_fma_f64, double, double, I3, "v_fma_f64")
FOUR(k_add_f64, double, double, I2, "v_add_f64")
FOUR(k_mul_f64, double, double, I2, "v_mul_f64")
FOUR(k_pk_fma_f32, double, double, I3, "v_pk_fma_f32")
FOUR(k_fma_f32, float, float, I3, "v_fma_f32")

typedef void (*kern_t)(Stamp *, int, float);
struct Entry {
  const char *name;
  kern_t fn[4];
};
#define E(N) {#N, {k_##N##_c1, k_##N##_c2, k_##N##_c4, k_##N##_c8}}

int main() {
  const Entry tab[] = {E(fma_f64), E(add_f64), E(mul_f64), E(pk_fma_f32), E(fma_f32)};
  Stamp *d_out = nullptr;
  CHECK(hipMalloc(&d_out, sizeof(Stamp) * 16));
  const int iters = 2000;
  printf("{\"unit\": \"shader cycles per wave64 instruction per SIMD; the stream of every wave is C independent dependency chains\",\n \"instructions\": {\n");
  const int n = sizeof(tab) / sizeof(tab[0]);
  for (int e = 0; e < n; ++e) {
    printf("  \"v_%s\": {", tab[e].name);
    for (int k = 1; k <= 4; k *= 2) {  // waves per SIMD
      printf("%s\"waves%d\": {", k > 1 ? ", " : "", k);
      for (int c = 0; c < 4; ++c) {
        std::vector<Stamp> h(16);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(tab[e].fn[c], dim3(1), dim3(256 * k), 0, 0, d_out, iters, 0.f);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * 4 * k, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int w = 0; w < 4 * k; ++w) {
          t0 = std::min(t0, h[w].t0);
          t1 = std::max(t1, h[w].t1);
        }
        printf("%s\"chains%d\": %.2f", c ? ", " : "", 1 << c, double(t1 - t0) / (double(k) * iters * 64));
      }
      printf("}");
    }
    printf("}%s\n", e + 1 < n ? "," : "");
  }
  printf(" }\n}\n");
  (void)hipFree(d_out);
  return 0;
}
