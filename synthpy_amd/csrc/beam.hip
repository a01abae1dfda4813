// The caller of the path: init_beam (src/solvers-legacy/full_solver.py:547-835) draws the ray bundle with NumPy's global
// stream on the host -- 45 ms per 5e5 rays, twenty times what the GPU then needs to trace them.  This is the same
// bundle drawn on the device: the SAME distributions ('circular': t = 2*pi*U, u = U + U folded at 1, phi = pi*U,
// chi = divergence*N(0,1), full_solver.py:567-580; 'square'/'rectangular': u, t = 2*U - 1, :600-640), a DIFFERENT
// stream (counter-based Philox4x32-10 keyed by the seed and the ray index), so a seeded run is reproducible across
// GPUs and chunk sizes but is not NumPy's sample.  Offered next to the host path (which reproduces the reference's
// seeded rays bit for bit), never instead of it.
#include <cmath>

#include "common.hpp"

namespace {

struct Philox {
  uint32_t c[4], k[2];
};
__device__ __forceinline__ void philox_round(Philox &p) {
  const uint64_t a = (uint64_t)0xD2511F53u * p.c[0], b = (uint64_t)0xCD9E8D57u * p.c[2];
  const uint32_t c0 = (uint32_t)(b >> 32) ^ p.c[1] ^ p.k[0], c2 = (uint32_t)(a >> 32) ^ p.c[3] ^ p.k[1];
  p.c[1] = (uint32_t)b;
  p.c[3] = (uint32_t)a;
  p.c[0] = c0;
  p.c[2] = c2;
  p.k[0] += 0x9E3779B9u;
  p.k[1] += 0xBB67AE85u;
}
// four 32-bit words for (ray, draw) under `seed`
__device__ __forceinline__ void philox4(uint64_t ray, uint32_t draw, uint64_t seed, uint32_t (&out)[4]) {
  Philox p{{(uint32_t)ray, (uint32_t)(ray >> 32), draw, 0x5eedu}, {(uint32_t)seed, (uint32_t)(seed >> 32)}};
#pragma unroll
  for (int r = 0; r < 10; ++r) philox_round(p);
  out[0] = p.c[0];
  out[1] = p.c[1];
  out[2] = p.c[2];
  out[3] = p.c[3];
}
// uniform in [0, 1) with 53 bits, as NumPy's random_sample
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {
  return (double)((((uint64_t)(hi >> 5)) << 26) | (uint64_t)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

// beam_type 0 circular (size a; radial law u = U + U folded at 1: the legacy generation), 1 square / rectangular (a x b),
// 2 linear (rays along a line in x, angles in the x-z plane, launched at z = -ne_extent whatever the probing direction:
// full_solver.py:707-720), 3 circular with the JAX generation's radial law u = np.random.power(2) = sqrt(U)
// (src/simulator/beam.py:66-77: positions uniform over the disc)
__global__ void k_beam(double *__restrict__ s0, int64_t N, int beam_type, double a, double b, double divergence, double ne_extent,
                       int axis, uint64_t seed, uint64_t first_ray) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= N) return;
  uint32_t w0[4], w1[4], w2[4];
  philox4(first_ray + (uint64_t)i, 0u, seed, w0);
  philox4(first_ray + (uint64_t)i, 1u, seed, w1);
  philox4(first_ray + (uint64_t)i, 2u, seed, w2);
  const double U0 = u01(w0[0], w0[1]), U1 = u01(w0[2], w0[3]), U2 = u01(w1[0], w1[1]), U3 = u01(w1[2], w1[3]);
  const double G0 = u01(w2[0], w2[1]), G1 = u01(w2[2], w2[3]);
  const double gauss = sqrt(-2.0 * log(1.0 - G0)) * cos((2.0 * M_PI) * G1);  // Box-Muller, 1 - G0 in (0, 1]
  double p1, p2, phi;
  if (beam_type == 2) {  // linear: as written in the reference, not rotated with the probing direction
    const double chi = divergence * gauss;
    double sc, cc;
    sincos(chi, &sc, &cc);
    s0[0 * N + i] = a * (2.0 * U0 - 1.0);
    s0[1 * N + i] = 0.0;
    s0[2 * N + i] = -ne_extent;
    s0[3 * N + i] = sr::kC * sc;
    s0[4 * N + i] = 0.0;
    s0[5 * N + i] = sr::kC * cc;
    s0[6 * N + i] = 1.0;
    s0[7 * N + i] = 0.0;
    s0[8 * N + i] = 0.0;
    return;
  }
  if (beam_type == 0 || beam_type == 3) {
    const double t = (2.0 * M_PI) * U0;
    double u;
    if (beam_type == 0) {
      u = U1 + U2;
      u = u > 1.0 ? 2.0 - u : u;
    } else {
      u = sqrt(U1);  // power(2): density 2u on [0, 1)
    }
    phi = M_PI * U3;
    p1 = a * u * cos(t);
    p2 = a * u * sin(t);
  } else {
    const double t = 2.0 * U0 - 1.0, u = 2.0 * U1 - 1.0;
    phi = M_PI * U2;
    p1 = a * u;
    p2 = b * t;
  }
  const double chi = divergence * gauss;
  // lateral axes of the cross-section for each probing direction: x -> (y, z), y -> (x, z), z -> (x, y)  (_beam.py)
  const int l1 = axis == 0 ? 1 : 0, l2 = axis == 2 ? 1 : 2;
  double sc, cc, sp, cp;
  sincos(chi, &sc, &cc);
  sincos(phi, &sp, &cp);
  s0[axis * N + i] = -ne_extent;
  s0[l1 * N + i] = p1;
  s0[l2 * N + i] = p2;
  s0[(3 + axis) * N + i] = sr::kC * cc;
  s0[(3 + l1) * N + i] = sr::kC * sc * cp;
  s0[(3 + l2) * N + i] = sr::kC * sc * sp;
  s0[6 * N + i] = 1.0;
  s0[7 * N + i] = 0.0;
  s0[8 * N + i] = 0.0;
}

}  // namespace

extern "C" int sr_rays_generate(sr_rays *r, int beam_type, double size_a, double size_b, double divergence, double ne_extent,
                                int probing_axis, uint64_t seed, uint64_t first_ray) {
  SR_CHECK(r != nullptr, "sr_rays_generate: NULL rays");
  SR_CHECK(beam_type >= 0 && beam_type <= 3,
           "sr_rays_generate: beam_type must be 0 (circular), 1 (square / rectangular), 2 (linear) or 3 (circular, power-law radius)");
  SR_CHECK(probing_axis >= 0 && probing_axis <= 2, "probing_axis must be 0, 1 or 2, got %d", probing_axis);
  if (r->n > 0) {
    hipLaunchKernelGGL(k_beam, dim3(sr::grid_for(r->n, 256)), dim3(256), 0, sr::ctx().stream, r->s0, r->n, beam_type, size_a, size_b,
                       divergence, ne_extent, probing_axis, seed, first_ray);
    SR_HIP(hipGetLastError());
  }
  {  // the launch positions' bounding box is known from the beam's parameters: no kernel, no wait
    const int l1 = probing_axis == 0 ? 1 : 0, l2 = probing_axis == 2 ? 1 : 2;
    double h1 = std::fabs(size_a), h2 = beam_type == 1 ? std::fabs(size_b) : std::fabs(size_a);
    for (int q = 0; q < 6; ++q) r->bbox[q] = 0.0;
    if (beam_type == 2) {  // a line in x launched at z = -ne_extent whatever the probing axis (full_solver.py:707-721)
      r->bbox[0] = -h1;
      r->bbox[3] = h1;
      r->bbox[2] = r->bbox[5] = -ne_extent;
    } else {
      r->bbox[l1] = -h1;
      r->bbox[3 + l1] = h1;
      r->bbox[l2] = -h2;
      r->bbox[3 + l2] = h2;
      r->bbox[probing_axis] = r->bbox[3 + probing_axis] = -ne_extent;
    }
    r->have_bbox = true;
    r->bbox_given = false;
  }
  r->have_s0 = true;
  r->traced = false;
  return SR_OK;
}
