// VALU issue cost on gfx950 (MI355X): cycles one wave64 instruction holds its SIMD, for the instruction types the
// tracer kernels are made of, at 1 / 2 / 4 waves per SIMD.  The roofline of k_trace_f64 / k_trace_mixed is VALU issue
// (DESIGN.md "Measured"): bench.py prices the kernel's measured per-type instruction counts with THIS table
// (profiles/r02_valu_issue.json), so the bound fraction can be recomputed from the committed files.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue > profiles/r02_valu_issue.json
//
// Method: one workgroup of 256*k threads = k waves on each of the CU's 4 SIMDs; every wave runs `iters` times a block
// of 64 INDEPENDENT instructions (8 destination registers x 8 repeats, so no instruction waits for the one before it),
// stamps s_memtime before and after; cycles per instruction per SIMD = (last end - first start) / (k * iters * 64).
// s_memrealtime (100 MHz) beside it gives the shader clock during the loop.  A second launch with one such workgroup
// on every CU (grid 256) timed with HIP events gives the same figure chip-wide in nanoseconds.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 2;                                                                      \
    }                                                                                \
  } while (0)

// 8 independent destinations %0..%7, sources %8, %9; INS is the text of one instruction with D = destination
#define REP8(I0, I1, I2, I3, I4, I5, I6, I7) I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 \
  I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7 I0 I1 I2 I3 I4 I5 I6 I7

struct Stamp {
  unsigned long long t0, t1, r0, r1;
};

#define KERNEL(NAME, DT, ST, DINIT, SINIT, LIST) KERNEL_(NAME, DT, ST, DINIT, SINIT, LIST)
#define KERNEL_(NAME, DT, ST, DINIT, SINIT, I0, I1, I2, I3, I4, I5, I6, I7)                                       \
  __global__ __launch_bounds__(1024) void NAME(Stamp *out, int iters, float seed) {                                \
    DT d0 = DINIT, d1 = DINIT, d2 = DINIT, d3 = DINIT, d4 = DINIT, d5 = DINIT, d6 = DINIT, d7 = DINIT;             \
    ST a = SINIT, b = SINIT;                                                                                       \
    __syncthreads();                                                                                               \
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();             \
    for (int it = 0; it < iters; ++it)                                                                             \
      asm volatile(REP8(I0, I1, I2, I3, I4, I5, I6, I7)                                                            \
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)                \
                   : "v"(a), "v"(b)                                                                                \
                   : "vcc", "s20", "s21");                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();             \
    asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7));                     \
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1}; \
  }

#define FS (seed + 1.0f)
#define DS ((double)seed + 1.0)
#define PS (f2{seed + 1.0f, seed + 2.0f})
#define IS ((unsigned)seed + 3u)
#define LS ((unsigned long long)seed + 3ull)

// three-operand forms: D = D*a + b
#define T3(OP) OP " %0, %0, %8, %9\n", OP " %1, %1, %8, %9\n", OP " %2, %2, %8, %9\n", OP " %3, %3, %8, %9\n", \
               OP " %4, %4, %8, %9\n", OP " %5, %5, %8, %9\n", OP " %6, %6, %8, %9\n", OP " %7, %7, %8, %9\n"
// two-operand forms: D = D op a
#define T2(OP) OP " %0, %0, %8\n", OP " %1, %1, %8\n", OP " %2, %2, %8\n", OP " %3, %3, %8\n", OP " %4, %4, %8\n", \
               OP " %5, %5, %8\n", OP " %6, %6, %8\n", OP " %7, %7, %8\n"
// one-operand forms from a common source: D = f(a)
#define T1S(OP) OP " %0, %8\n", OP " %1, %8\n", OP " %2, %8\n", OP " %3, %8\n", OP " %4, %8\n", OP " %5, %8\n", \
                OP " %6, %8\n", OP " %7, %8\n"
// one-operand forms in place: D = f(D)
#define T1D(OP) OP " %0, %0\n", OP " %1, %1\n", OP " %2, %2\n", OP " %3, %3\n", OP " %4, %4\n", OP " %5, %5\n", \
                OP " %6, %6\n", OP " %7, %7\n"
#define TSEL(OP) OP " %0, %0, %8, vcc\n", OP " %1, %1, %8, vcc\n", OP " %2, %2, %8, vcc\n", OP " %3, %3, %8, vcc\n", \
                 OP " %4, %4, %8, vcc\n", OP " %5, %5, %8, vcc\n", OP " %6, %6, %8, vcc\n", OP " %7, %7, %8, vcc\n"
#define TCMP(OP) OP " vcc, %0, %8\n", OP " vcc, %1, %8\n", OP " vcc, %2, %8\n", OP " vcc, %3, %8\n", OP " vcc, %4, %8\n", \
                 OP " vcc, %5, %8\n", OP " vcc, %6, %8\n", OP " vcc, %7, %8\n"
#define TLSHL(OP) OP " %0, %0, 1, %8\n", OP " %1, %1, 1, %8\n", OP " %2, %2, 1, %8\n", OP " %3, %3, 1, %8\n", \
                  OP " %4, %4, 1, %8\n", OP " %5, %5, 1, %8\n", OP " %6, %6, 1, %8\n", OP " %7, %7, 1, %8\n"

// VOP2 multiply-accumulate: D = a*b + D
#define TMAC(OP) OP " %0, %8, %9\n", OP " %1, %8, %9\n", OP " %2, %8, %9\n", OP " %3, %8, %9\n", OP " %4, %8, %9\n", \
                 OP " %5, %8, %9\n", OP " %6, %8, %9\n", OP " %7, %8, %9\n"
// select by a mask held in an SGPR pair (VOP3 form) instead of vcc
#define TSELS(OP) OP " %0, %0, %8, s[20:21]\n", OP " %1, %1, %8, s[20:21]\n", OP " %2, %2, %8, s[20:21]\n", \
                  OP " %3, %3, %8, s[20:21]\n", OP " %4, %4, %8, s[20:21]\n", OP " %5, %5, %8, s[20:21]\n", \
                  OP " %6, %6, %8, s[20:21]\n", OP " %7, %7, %8, s[20:21]\n"

KERNEL(k_fma_f32, float, float, FS, FS, T3("v_fma_f32"))
KERNEL(k_fmac_f32, float, float, FS, FS, TMAC("v_fmac_f32"))
KERNEL(k_add_f32, float, float, FS, FS, T2("v_add_f32"))
KERNEL(k_fmac_f64, double, double, DS, DS, TMAC("v_fmac_f64"))
KERNEL(k_cndmask_b32_sgpr, unsigned, unsigned, IS, IS, TSELS("v_cndmask_b32"))
KERNEL(k_max_u32, unsigned, unsigned, IS, IS, T2("v_max_u32"))
KERNEL(k_cmp_lt_u32, unsigned, unsigned, IS, IS, TCMP("v_cmp_lt_u32"))
KERNEL(k_mul_f32, float, float, FS, FS, T2("v_mul_f32"))
KERNEL(k_min_f32, float, float, FS, FS, T2("v_min_f32"))
KERNEL(k_pk_fma_f32, f2, f2, PS, PS, T3("v_pk_fma_f32"))
KERNEL(k_pk_mul_f32, f2, f2, PS, PS, T2("v_pk_mul_f32"))
KERNEL(k_pk_add_f32, f2, f2, PS, PS, T2("v_pk_add_f32"))
KERNEL(k_fma_f64, double, double, DS, DS, T3("v_fma_f64"))
KERNEL(k_add_f64, double, double, DS, DS, T2("v_add_f64"))
KERNEL(k_mul_f64, double, double, DS, DS, T2("v_mul_f64"))
KERNEL(k_min_f64, double, double, DS, DS, T2("v_min_f64"))
KERNEL(k_cvt_f32_f64, float, double, FS, DS, T1S("v_cvt_f32_f64"))
KERNEL(k_cvt_f64_f32, double, float, DS, FS, T1S("v_cvt_f64_f32"))
KERNEL(k_rcp_f32, float, float, FS, FS, T1D("v_rcp_f32"))
KERNEL(k_rcp_f64, double, double, DS, DS, T1D("v_rcp_f64"))
KERNEL(k_cndmask_b32, unsigned, unsigned, IS, IS, TSEL("v_cndmask_b32"))
KERNEL(k_mov_b32, unsigned, unsigned, IS, IS, T1S("v_mov_b32"))
KERNEL(k_mov_b64, unsigned long long, unsigned long long, LS, LS, T1S("v_mov_b64"))
KERNEL(k_add_u32, unsigned, unsigned, IS, IS, T2("v_add_u32"))
KERNEL(k_lshl_add_u64, unsigned long long, unsigned long long, LS, LS, TLSHL("v_lshl_add_u64"))
KERNEL(k_cmp_lt_f64, double, double, DS, DS, TCMP("v_cmp_lt_f64"))
KERNEL(k_cmp_lt_f32, float, float, FS, FS, TCMP("v_cmp_lt_f32"))

// 4 selects interleaved with 4 float64 multiply-adds (8 instructions per block as everywhere else)
__global__ __launch_bounds__(1024) void k_selmix(Stamp *out, int iters, float seed) {
  unsigned d0 = IS, d1 = IS, d2 = IS, d3 = IS, m = IS;
  double e0 = DS, e1 = DS, e2 = DS, e3 = DS, a = DS, b = DS;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
    asm volatile(REP8("v_cndmask_b32 %0, %0, %8, vcc\n", "v_fma_f64 %4, %4, %9, %10\n", "v_cndmask_b32 %1, %1, %8, vcc\n",
                      "v_fma_f64 %5, %5, %9, %10\n", "v_cndmask_b32 %2, %2, %8, vcc\n", "v_fma_f64 %6, %6, %9, %10\n",
                      "v_cndmask_b32 %3, %3, %8, vcc\n", "v_fma_f64 %7, %7, %9, %10\n")
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3)
                 : "v"(m), "v"(a), "v"(b)
                 : "vcc");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  asm volatile("" ::"v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(e0), "v"(e1), "v"(e2), "v"(e3));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = Stamp{t0, t1, r0, r1};
}

typedef void (*kern_t)(Stamp *, int, float);
struct Entry {
  const char *name;
  kern_t fn;
};

int main() {
  const Entry tab[] = {
      {"v_fma_f32", k_fma_f32},         {"v_fmac_f32", k_fmac_f32},       {"v_add_f32", k_add_f32},
      {"v_fmac_f64", k_fmac_f64},       {"v_cndmask_b32 (mask in SGPRs)", k_cndmask_b32_sgpr},
      {"v_max_u32", k_max_u32},         {"v_cmp_lt_u32", k_cmp_lt_u32},   {"4 x (v_cndmask_b32, v_fma_f64) per 8", k_selmix},
      {"v_mul_f32", k_mul_f32},         {"v_min_f32", k_min_f32},
      {"v_pk_fma_f32", k_pk_fma_f32},   {"v_pk_mul_f32", k_pk_mul_f32},   {"v_pk_add_f32", k_pk_add_f32},
      {"v_fma_f64", k_fma_f64},         {"v_add_f64", k_add_f64},         {"v_mul_f64", k_mul_f64},
      {"v_min_f64", k_min_f64},         {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32},
      {"v_rcp_f32", k_rcp_f32},         {"v_rcp_f64", k_rcp_f64},         {"v_cndmask_b32", k_cndmask_b32},
      {"v_mov_b32", k_mov_b32},         {"v_mov_b64", k_mov_b64},         {"v_add_u32", k_add_u32},
      {"v_lshl_add_u64", k_lshl_add_u64}, {"v_cmp_lt_f64", k_cmp_lt_f64}, {"v_cmp_lt_f32", k_cmp_lt_f32},
  };
  const int iters = 2000, per_iter = 64, n_cu = 256;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  Stamp *d = nullptr;
  CHECK(hipMalloc(&d, sizeof(Stamp) * 16 * n_cu));
  std::vector<Stamp> h(16 * n_cu);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"iters\": %d, \"instructions_per_iteration\": %d,\n", prop.name, prop.gcnArchName,
         iters, per_iter);
  printf(" \"unit\": \"shader cycles one wave64 instruction holds its SIMD (one workgroup of 256*k threads on one CU, k waves per SIMD)\",\n");
  printf(" \"chip_unit\": \"ns per wave64 instruction per SIMD with one such workgroup on every CU (HIP events)\",\n \"instructions\": {\n");
  const size_t n_ent = sizeof(tab) / sizeof(tab[0]);
  for (size_t q = 0; q < n_ent; ++q) {
    printf("  \"%s\": {", tab[q].name);
    for (int k = 1; k <= 4; k *= 2) {
      const int threads = 256 * k, waves = 4 * k;
      // one CU
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(tab[q].fn, dim3(1), dim3(threads), 0, 0, d, iters, 0.5f);
      CHECK(hipDeviceSynchronize());
      CHECK(hipMemcpy(h.data(), d, sizeof(Stamp) * waves, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
      for (int w = 0; w < waves; ++w) {
        t0 = std::min(t0, h[w].t0);
        t1 = std::max(t1, h[w].t1);
        r0 = std::min(r0, h[w].r0);
        r1 = std::max(r1, h[w].r1);
      }
      const double cyc = double(t1 - t0) / (double(k) * iters * per_iter);
      const double ghz = (r1 > r0) ? double(t1 - t0) / (double(r1 - r0) * 10.0) : 0.0;  // 100 MHz reference: 10 ns per tick
      // every CU
      hipLaunchKernelGGL(tab[q].fn, dim3(n_cu), dim3(threads), 0, 0, d, iters, 0.5f);
      CHECK(hipEventRecord(e0, 0));
      for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(tab[q].fn, dim3(n_cu), dim3(threads), 0, 0, d, iters, 0.5f);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double ns = double(ms) * 1e6 / 5.0 / (double(k) * iters * per_iter);
      printf("%s\"waves%d\": {\"cycles\": %.3f, \"clock_ghz\": %.3f, \"chip_ns\": %.4f}", k == 1 ? "" : ", ", k, cyc, ghz, ns);
    }
    printf("}%s\n", q + 1 < n_ent ? "," : "");
  }
  printf(" }\n}\n");
  (void)hipFree(d);
  return 0;
}
