// Vector-memory pipe of one CU on gfx950 (MI355X): cycles one wave64 load instruction costs when every line it touches is
// in L1, by address pattern and width.  The tracer kernels gather: every lane loads the corner records of ITS cell, so a
// load instruction touches 1..64 lines; this measures what that costs against a coalesced read of the same bytes.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_rate.hip -o /tmp/gather_rate && /tmp/gather_rate > profiles/r02_gather_rate.json
//
// One workgroup of 512 threads (2 waves per SIMD) on ONE CU reads a 16 KB table (L1-resident after the first pass),
// 16 loads in flight per wave, `iters` passes (the 16 float adds that consume them cost 16 x 2 VALU cycles per wave against
// >= 16 x 16 cycles of loads); cycles per load instruction per CU = (last end - first start) / (8 * iters * 16).
// Patterns (byte address of lane l): same = every lane the same record; line = l * 128 (64 different lines);
// quad = (l / 4) * 128 + (l % 4) * W (four lanes per line); coalesced = l * W; cell16 = (l / 16) * 128 and cell4 = (l / 4) * 128
// (4 and 16 different records, each read by 16 / 4 neighbouring lanes: rays that share a cell read the SAME corner record).
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

template <typename T>
__global__ __launch_bounds__(512) void k_gather(const char *__restrict__ tab, int pattern, int iters, Stamp *out, T *sink) {
  const int l = threadIdx.x & 63;
  const int W = sizeof(T);
  size_t off = 0;
  if (pattern == 1) off = (size_t)l * 128;
  if (pattern == 2) off = (size_t)(l / 4) * 128 + (size_t)(l % 4) * W;
  if (pattern == 3) off = (size_t)l * W;
  if (pattern == 4) off = (size_t)(l / 16) * 128;
  if (pattern == 5) off = (size_t)(l / 4) * 128;
  const char *p = tab + off;
  float acc = 0.f;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const char *pi = p + (it & 1) * 16384;  // two images of the table alternate: the loads are not loop-invariant
    T v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)  // 16 independent loads, all L1 hits after the first passes
      v[q] = *reinterpret_cast<const T *>(pi + ((q & 1) ? 8192 : 0) + (q >> 1) * W * (pattern == 3 ? 64 : 0));
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += *reinterpret_cast<const float *>(&v[q]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (l == 0) out[threadIdx.x >> 6] = Stamp{t0, t1};
  if (acc == 12345.f) sink[threadIdx.x] = T{};
}

template <typename T>
int run(const char *name, const char *d_tab, Stamp *d_out, void *d_sink, bool last) {
  const char *pat[6] = {"same", "line", "quad", "coalesced", "cell16", "cell4"};
  const int iters = 4000;
  printf("  \"%s\": {", name);
  for (int p = 0; p < 6; ++p) {
    std::vector<Stamp> h(8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_gather<T>, dim3(1), dim3(512), 0, 0, d_tab, p, iters, d_out, (T *)d_sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h.data(), d_out, sizeof(Stamp) * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (auto &s : h) {
      t0 = std::min(t0, s.t0);
      t1 = std::max(t1, s.t1);
    }
    const double cyc = double(t1 - t0) / (8.0 * iters * 16);
    printf("%s\"%s\": {\"cycles_per_instruction\": %.2f, \"bytes_per_cycle\": %.1f}", p ? ", " : "", pat[p], cyc, 64.0 * sizeof(T) / cyc);
  }
  printf("}%s\n", last ? "" : ",");
  return 0;
}

int main() {
  char *d_tab = nullptr;
  Stamp *d_out = nullptr;
  void *d_sink = nullptr;
  CHECK(hipMalloc(&d_tab, 1 << 16));
  CHECK(hipMemset(d_tab, 0, 1 << 16));
  CHECK(hipMalloc(&d_out, sizeof(Stamp) * 8));
  CHECK(hipMalloc(&d_sink, 512 * 16));
  printf("{\"unit\": \"shader cycles per wave64 load instruction on ONE CU (8 waves, 16 loads in flight each, every line in L1)\",\n");
  printf(" \"patterns\": \"same: all lanes one record; line: 64 lines; quad: 16 lines, 4 lanes each; coalesced: contiguous; cell16 / cell4: 4 / 16 records, each read by 16 / 4 neighbouring lanes\",\n \"loads\": {\n");
  if (run<float>("global_load_dword", d_tab, d_out, d_sink, false)) return 2;
  if (run<float2>("global_load_dwordx2", d_tab, d_out, d_sink, false)) return 2;
  if (run<float4>("global_load_dwordx4", d_tab, d_out, d_sink, true)) return 2;
  printf(" }\n}\n");
  (void)hipFree(d_tab);
  (void)hipFree(d_out);
  (void)hipFree(d_sink);
  return 0;
}
