/*
 * synthray.h — C ABI of libsynthray.so: the MI355X (gfx950) engine for synthPy's
 * ray-propagation → detector hot path.
 *
 * The reference (MAGPIE-ICL/synthPy) is pure Python and has no FFI of its own; the
 * boundary sits directly beneath its Python API and each entry point below replaces
 * the body of one reference function (cited as path:line in the reference tree).
 * Plain pointers and sizes only.  Conventions follow the reference's NumPy layouts:
 *   rays     (9, N) float64 row-major: x y z vx vy vz amplitude phase polarisation
 *   rf       (4, N) float64: x theta y phi         Jf (2, N) complex128 (re, im interleaved)
 *   volumes  C-order [ix][iy][iz] (z fastest), coordinates float32 as ScalarDomain keeps them
 *   images   [y_bin][x_bin]
 *
 * Ownership: the caller owns every host buffer; the library owns device memory behind
 * opaque handles and keeps no host pointer after a call returns.
 * Errors: 0 on success, a negative SR_ERR_* otherwise, text via sr_last_error()
 * (thread-local).  A ray that fails (rejected by an aperture, NaN input) is NaN in the
 * output, never an error — as in the reference.
 * Threading: one device per process/thread; calls are serialised on one HIP stream.
 */
#ifndef SYNTHRAY_H
#define SYNTHRAY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_ERR_INVALID (-1)  /* bad argument */
#define SR_ERR_HIP (-2)      /* a HIP runtime call failed (no device, out of memory, launch) */
#define SR_ERR_RCCL (-3)     /* RCCL missing or a collective failed */
#define SR_ERR_STATE (-4)    /* handle used before the data it needs exists */

typedef struct sr_volume sr_volume; /* device-resident fields of one ScalarDomain */
typedef struct sr_rays sr_rays;     /* device-resident ray bundle: s0, then sf / rf / Jf */
typedef struct sr_image sr_image;   /* device-resident detector image */
typedef struct sr_comm sr_comm;     /* RCCL communicator (one rank per GPU) */

/* ---- runtime ------------------------------------------------------------------ */
/* Threading and processes: ONE device per process, and the library is NOT thread-safe -- its context (device, the two
 * streams, the selected stream, timing events) is process-global, and sr_init(other_device) re-creates the streams, so
 * no handle made on the first device may be live across it.  Multi-GPU = one process per GPU (bench.py, run_trace,
 * distributed.py), as the reference runs one MPI rank per device.  Calls from one thread at a time. */
int sr_init(int device);            /* select the GPU and create the stream; idempotent per device */
/* >= 0, or SR_ERR_HIP.  0 only when the machine has no amdgpu driver node (/dev/kfd) this process may open.  Before the
 * process's first HIP call the device is opened in a bounded retry (open(/dev/kfd) + hsa_init, 6 tries, 50 ms doubling):
 * ranks of one job start in the same instant, and HIP's once-per-process initialisation cannot be repeated after it found
 * no agent.  A failure keeps the runtime's own words (errno / HSA status / hipError name) in sr_last_error(). */
int sr_device_count(void);
int sr_synchronize(void);            /* waits for every stream of the library */
/* HBM free / in all on the selected device (hipMemGetInfo): what ScalarDomain's auto_batching sizes its regions from
 * (the reference asks psutil / pynvml, src/simulator/domain.py:140-165) */
int sr_device_memory(int64_t *free_bytes, int64_t *total_bytes);
/* Every call queues its GPU work on the library's SELECTED stream (0 by default; 1 = a second one).  Work on different
 * streams may overlap: a job of many small ray bundles alternates them so that one bundle's tail runs beside the next
 * one's start-up (the reference's drivers trace 5e5-ray chunks one after the other, pvti_trace_mpi.py:144-163).  The
 * caller orders what the streams share: sr_synchronize() after creating volumes / zeroing images and before reading
 * images; a ray bundle is used with one stream at a time. */
int sr_stream_select(int index);
/* Orders the two streams without stopping the host: work queued on stream `waiting` AFTER this call starts once everything
 * queued on stream `on` BEFORE it is done.  The slab pipeline traces on stream 0 and moves the hand-off records (ncclSend /
 * ncclRecv) on stream 1, so that the send of chunk i runs beside the trace of chunk i+1 (distributed.SlabPipeline). */
int sr_stream_wait(int waiting, int on);
/* Page-locked host memory (hipHostMalloc): copies between it and the GPU run at the speed of the link and beside GPU work,
 * where pageable memory is staged by the driver at a third of that.  For the arrays sr_trace / sr_rays_download fill
 * (the reference returns fresh NumPy arrays from ScalarDomain.solve, full_solver.py:391-400; engine.trace hands out
 * arrays over such blocks and returns them when the arrays are collected). */
int sr_host_alloc(void **out, size_t bytes);
void sr_host_free(void *p);
/* sr_trace on a large host bundle keeps its device-side working set (three chunk-sized ray bundles and two staging blocks,
 * ~1.5 GB of HBM at the default chunk) between calls, so that a loop of solve() calls does not allocate and release it
 * every time.  This releases it (SYNTHRAY_TRACE_CACHE=0: never kept). */
int sr_release_caches(void);
const char *sr_last_error(void);
const char *sr_version(void);         /* "synthray <ver> (gfx950) src:<hash of the library's sources>" */

/* ---- A1 + A5: ScalarDomain.calc_dndr / n_refrac -------------------------------
 * replaces src/solvers-legacy/full_solver.py:211-234 (calc_dndr), :271-274 (n_refrac),
 *          src/simulator/propagator.py:63-91 (n_refrac, dndr's per-call np.gradient).
 * omega = 2*pi*c/lwl, n_c = 3.14207787e-4*omega^2, ne_nc = float32(ne/n_c),
 * dnd{x,y,z} = float32(-c^2/2) * np.gradient(ne_nc, coord, axis) in float32 exactly as
 * numpy evaluates it (non-uniform 2nd-order interior, 1st-order edges), and with
 * SR_VOL_PHASE the refractive index n = sqrt(1-(5.64e4*sqrt(ne*1e-6)/omega)^2) in float64.
 * ne: nx*ny*nz values, float64 (ne_is_f64 != 0) or float32.  probing_axis (0 x,1 y,2 z)
 * fixes the on-device layout (that axis is made the fastest one). */
#define SR_VOL_PHASE 1
int sr_volume_create(sr_volume **out, const void *ne, int ne_is_f64, int nx, int ny, int nz,
                     const float *x, const float *y, const float *z, double lwl,
                     int probing_axis, int flags);
/* same, from gradient volumes the caller already holds (float32, C-order) and an optional
 * float64 refractive index: the state calc_dndr leaves in ScalarDomain.dndx/dndy/dndz */
int sr_volume_create_from_fields(sr_volume **out, const float *dndx, const float *dndy,
                                 const float *dndz, const double *nref, double omega, int nx,
                                 int ny, int nz, const float *x, const float *y, const float *z,
                                 int probing_axis);
/* read the fields back in the reference's layout (any pointer may be NULL);
 * nref_minus_1 is n-1 in float64 */
int sr_volume_fields(const sr_volume *v, float *dndx, float *dndy, float *dndz, double *nref_minus_1);
/* A3/A4 at given points: the RegularGridInterpolator gathers of dsdt (full_solver.py:317-347, :538-541).
 * pts is (N, 3) [x, y, z]; out is (4, N): dndx, dndy, dndz, n-1 (0 where the volume has no phase field);
 * out-of-bounds points give the fill values (0, 0, 0, 0), NaN points give NaN. */
int sr_volume_sample(const sr_volume *v, const double *pts, int64_t n_pts, double *out);
/* The optional terms of dsdt: inverse-bremsstrahlung attenuation d(amp) = kappa(x)*amp (full_solver.py:334-339,
 * the field of ScalarDomain.kappa() :243-268) and Faraday rotation d(pol) = VerdetConst*ne(x)*(B(x).v)
 * (:356-374).  Replaces set_up_interps() (:276-289): float64 volumes in the reference's layout, kappa and ne
 * (nx, ny, nz), B (nx, ny, nz, 3); kappa may be NULL (inv_brems off), ne and B may both be NULL (B_on off).
 * Rays traced through a volume with these fields carry amp and pol (rows 6 and 8 of sf) through the same RK4
 * steps, in float64 throughout: the fields are held per ray as bilinear coefficient planes beside the gradient planes
 * (k_trace_f64<., AUX, .>); SR_PREC_MIXED asked for on such a volume runs the same float64 kernel. */
int sr_volume_attach_aux(sr_volume *v, const double *kappa, const double *ne, const double *B, double verdet);
/* the gathers of atten()/get_ne()/get_B() at given points: out is (5, N): kappa, ne, Bx, By, Bz (fill 0) */
int sr_volume_sample_aux(const sr_volume *v, const double *pts, int64_t n_pts, double *out);
/* A12 (the reference's region loop + back_propogate, propagator.py:300-349, 366-452; BASELINE config 5): the node
 * planes k_lo..k_hi of the probing axis as a volume of their own, for domains cut into slabs across GPUs.
 * nx, ny, nz, x, y, z describe the WHOLE domain; ne_slab holds the planes max(k_lo-1, 0)..min(k_hi+1, n-1) of the
 * probing axis (one halo plane each side for np.gradient's central difference; C order, that axis shortened), so the
 * slab's gradient values equal the whole domain's bit for bit.  Consecutive slabs share a node plane
 * (k_hi of one = k_lo of the next); rays go from slab to slab with SR_HANDOFF_* and the sr_rays_handoff_* calls. */
int sr_volume_create_slab(sr_volume **out, const void *ne_slab, int ne_is_f64, int nx, int ny, int nz,
                          const float *x, const float *y, const float *z, double lwl, int probing_axis,
                          int flags, int k_lo, int k_hi);
double sr_volume_omega(const sr_volume *v);
int64_t sr_volume_bytes(const sr_volume *v); /* HBM held by the handle */
void sr_volume_destroy(sr_volume *v);

/* ---- the step before the path: volume synthesis ------------------------------------
 * gaussian3D.domain_fft (src/field_generator/gaussian3D.py:215-271): out = Re(ifftn(noise * amp)) [/ max|.| when
 * normalise], noise complex128 (n0, n1, n2) interleaved (the caller's seeded np.random draws), amp = sqrt(S(k))
 * float32, out float64; C order.  3-D inverse FFT by hipFFT (bound at first use). */
int sr_field_ifft_real(const double *noise, const float *amp, int n0, int n1, int n2, int normalise, double *out);
/* ---- the step after the path: radially binned power spectrum of a detector image -----------
 * radial_2Dspectrum (src/utils/power_spectrum.py:372-421): |fft2(img)|^2/(n0*n1)^2 summed and counted over the
 * wavenumber bins [edges[b], edges[b+1]); k0 (n0), k1 (n1): wavenumber of each index of the unshifted transform. */
int sr_radial_spectrum2d(const double *img, int n0, int n1, const double *k0, const double *k1,
                         const double *edges, int n_edges, double *sum, uint64_t *count);

/* ---- A2 + A3 + A4 + A6: ScalarDomain.solve / propagator.solve -----------------
 * replaces full_solver.py:376-403 (solve), :516-544 (dsdt), :317-347 (dndr, phase: the
 * RegularGridInterpolator gathers), :838-894 (ray_to_Jonesvector);
 * src/simulator/propagator.py:351-702 (solve), :94-175, :178-298.
 * Integrates ds/dt = dsdt(s) over [0, t_end] per ray with RK4 stepping from node plane to
 * node plane of the probing axis (`substeps` per cell), keeps the final state sf (at t_end,
 * as the reference), back-projects to the plane `extent` on the probing axis (rf) and forms
 * the Jones vector (Jf). */
#define SR_ROWS_LEGACY 0 /* y-probing rf rows (x, z): full_solver.py:866-872 */
#define SR_ROWS_JAX 1    /* y-probing rf rows (z, x): propagator.py:223-243 */
#define SR_PREC_F64 0
#define SR_PREC_MIXED 1
typedef struct {
  double t_end;         /* s; the reference uses sqrt(8)*extent/c (full_solver.py:381) */
  double extent;        /* m; exit plane coordinate for the back-projection */
  double dt;            /* s; step of the time-stepping fallback (rays the plane form cannot take);
                           <= 0 selects one probing-axis cell / c */
  int32_t probing_axis; /* 0 x, 1 y, 2 z; must equal the volume's */
  int32_t row_order;    /* SR_ROWS_* */
  int32_t substeps;     /* RK4 steps per cell (>= 1) */
  int32_t sort_rays;    /* bin rays by entry cell before the launch (results do not depend on it) */
  int32_t precision;    /* SR_PREC_F64: every operation float64 (differs from the oracle by fused
                           multiply-adds only).  SR_PREC_MIXED: float64 state, stage positions and
                           accumulation; float32 interpolation weights, blend and RK4 slopes (one step per
                           cell, no optional terms; otherwise the float64 kernels run) */
  int32_t handoff;      /* 0, or SR_HANDOFF_* when the volume is a slab of node planes (sr_volume_create_slab) */
} sr_trace_params;
#define SR_HANDOFF_ENTER 1 /* the rays' state arrives on the slab's first node plane (sr_rays_handoff_upload / _recv) */
#define SR_HANDOFF_EXIT 2  /* leave the state on the slab's last node plane for the next slab instead of sf / rf / Jf */

typedef struct {
  int64_t ray_steps;      /* RK4 steps taken, summed over rays */
  int64_t fallback_rays;  /* rays the first kernel passed on: precision f64 -> the time-stepping form; mixed -> the
                             float64 plane kernel (and from there, if not plane-form rays, the time-stepping form) */
  double trace_kernel_ms; /* HIP-event time of the plane-stepping kernel alone */
  double total_ms;        /* HIP-event time of the whole call on the stream */
} sr_trace_stats;

/* host buffers in and out (any of sf, rf, Jf, stats may be NULL) */
int sr_trace(const sr_volume *v, const double *s0, int64_t n_rays, const sr_trace_params *p,
             double *sf, double *rf, double *Jf, sr_trace_stats *stats);

/* A6 alone on a host (9, N) state: ray_to_Jonesvector (full_solver.py:838-894; propagator.py:178-298).
 * Jf may be NULL. */
int sr_ray_to_jones(const double *sf, int64_t n_rays, double extent, int probing_axis, int row_order,
                    double *rf, double *Jf);

/* device-resident form: rays stay in HBM between trace and deposit */
int sr_rays_create(sr_rays **out, int64_t n_rays);
int sr_rays_upload(sr_rays *r, const double *s0);                 /* (9, N) */
/* A bundle put together from several host chunks: s0 is (9, n) and becomes the rays first .. first + n - 1 of the bundle (rows at
 * the bundle's pitch N).  The reference's drivers trace chunks of 5e5 rays one after the other (pvti_trace_mpi.py:27, 144-163); a
 * GPU traces a dense bundle faster per ray, the rays are independent and the detector images are sums over rays, so the driver
 * merges consecutive chunks -- each still its own seeded draw -- into one bundle before the trace (run_trace.chunked_trace).  The
 * parts may come in any order and must cover 0 .. N - 1 between them; `last` != 0 on the final call: the launch positions' bounding
 * box is found over the whole bundle and the bundle counts as uploaded. */
int sr_rays_upload_part(sr_rays *r, const double *s0, int64_t n, int64_t first, int last);
/* The bundle drawn ON the device instead of uploaded: init_beam's distributions (full_solver.py:547-835; beam_type 0
 * 'circular' radius size_a, 1 'square' / 'rectangular' half-sizes size_a x size_b, 2 'linear' (:707-721: a line of
 * half-length size_a in x, angles in the x-z plane, launched at z = -ne_extent as written there), 3 'circular' with the JAX
 * generation's radial law np.random.power(2) (src/simulator/beam.py:66-77); launch plane -ne_extent on the probing
 * axis) from a counter-based Philox stream keyed by (seed, first_ray + ray index): reproducible across GPUs and chunk
 * sizes, but NOT NumPy's sample -- the host path (init_beam + sr_rays_upload) is the one that reproduces the reference's
 * seeded rays. */
int sr_rays_generate(sr_rays *r, int beam_type, double size_a, double size_b, double divergence, double ne_extent,
                     int probing_axis, uint64_t seed, uint64_t first_ray);
/* stats != NULL waits for the stream and reads the counters.  stats == NULL returns as soon as the work is queued and
 * the counters of this call are carried into the next one: a chunked driver passes NULL in its loop (no host round trip
 * per chunk) and reads the totals once with sr_rays_trace_stats. */
int sr_rays_trace(sr_rays *r, const sr_volume *v, const sr_trace_params *p, sr_trace_stats *stats);
/* ray_steps and fallback_rays summed over every trace of this bundle since its counters were last read (by this call
 * or by a trace with stats != NULL); the two times are those of the last trace.  Waits for the stream. */
int sr_rays_trace_stats(sr_rays *r, sr_trace_stats *stats);

/* Which kernel carried the last trace of the bundle: the number of node-plane segments of the tile path (trace_tile.inc:
 * dense float64 bundles -- >= 8 rays per lateral cell of the beam's bounding box, whole volumes and slabs, with or without
 * the optional terms -- the coefficient records of a lateral cell built once per workgroup in LDS; then
 * sr_trace_stats.trace_kernel_ms is the sum of its launches), or 0 for the per-ray kernels.  Same results either way. */
int sr_rays_tile_segments(const sr_rays *r);
/* 1 when that tile path ran the RECORDS kernel (round 5: the coefficient records ready-made in HBM, 128 bytes per node plane and
 * lateral cell, built with the volume's own arithmetic at the first such trace and brought into the tile's ring by LDS-DMA; four
 * workgroups per CU), 0 for the producers' kernel (records that do not fit in HBM, SYNTHRAY_TILE_RECORDS=0, the optional terms) or
 * when the tile path did not run.  Same results either way, bit for bit. */
int sr_rays_tile_records(const sr_rays *r);
/* The density from which sr_rays_trace takes the tile path: rays per lateral cell of the beam's bounding box (sr_rays_get_bbox).
 * What a job that cuts its rays into chunks sizes them by (distributed.plan_chunks: the slab pipeline's chunk is the smallest
 * that every rank still traces with the tile kernel; the reference's drivers use a fixed 5e5, pvti_trace_mpi.py:27). */
double sr_tile_min_density(void);
int sr_rays_download(const sr_rays *r, double *sf, double *rf, double *Jf); /* original ray order */
int sr_rays_download_s0(const sr_rays *r, double *s0);            /* the bundle as uploaded / generated, (9, N) */
/* Per ray (original order), a bound [rad] on how far the exit angles of the last trace may be from the SR_PREC_F64
 * build's: 0 for rays a float64 kernel wrote (SR_PREC_F64, the mixed build's second level, rays an exact-counts deposit
 * has traced again, SR_PREC_MIXED with sub-steps or optional terms: the float64 kernels ran), the mixed kernel's own estimate
 * otherwise (8 * 2^-24 * sum of |lateral velocity changes| / v_a).  Positions: the bound times the volume's length along the
 * probing axis plus the distance from its last node plane to the plane `extent`.  This is what sr_deposit_params.exact_counts
 * works from. */
int sr_rays_error_bound(const sr_rays *r, float *bound);
int64_t sr_rays_count(const sr_rays *r);
/* A12 hand-off records, (10, N) float64 in launch order: p_b, p_c, v_a, v_b, v_c, phase, t, amp, pol, ray index
 * (the plane form's state on the shared node plane; v_a = NaN: ray lost).  Written by a trace with
 * SR_HANDOFF_EXIT, consumed by a trace with SR_HANDOFF_ENTER; between GPUs they travel host-side
 * (download / upload) or device to device over RCCL (send / recv, on the library stream). */
int sr_rays_handoff_download(const sr_rays *r, double *rec);
/* The bounding box of the BEAM a bundle's rays belong to, (min x, y, z, max x, y, z) of their launch positions [m]: what the
 * library judges the ray density by when it picks the kernel (rays per lateral cell of the beam, trace.hip: tile_plan).  It is
 * found at sr_rays_upload / known from the parameters at sr_rays_generate; rays that ARRIVE by hand-off (sr_rays_handoff_upload /
 * _recv: ranks > 0 of a slab pipeline, the reference's region loop propagator.py:366-452) have none and are judged by the whole
 * lateral grid -- unless the caller, who knows which beam the job traces, says so here: a box given with sr_rays_set_bbox stays
 * with the bundle across hand-offs until the next upload / generate (bbox = NULL takes it back).  sr_rays_get_bbox: *known = 0
 * when the bundle has no box. */
int sr_rays_set_bbox(sr_rays *r, const double *bbox);
int sr_rays_get_bbox(const sr_rays *r, double *bbox, int *known);
int sr_rays_handoff_upload(sr_rays *r, const double *rec);
int sr_rays_handoff_send(sr_rays *r, sr_comm *comm, int peer);
int sr_rays_handoff_recv(sr_rays *r, sr_comm *comm, int peer);
void sr_rays_destroy(sr_rays *r);

/* ---- A7 + A8: ray-transfer-matrix optics --------------------------------------
 * replaces src/solvers-legacy/rtm_solver.py:48-136 (m_to_mm, lens, distance, apertures) and
 * the chains :197-286, :376-422; src/simulator/diagnostics.py:122-245, :388-481, :614-638.
 * r is (4, N) in mm.  A rejected ray becomes a NaN column.  kwave > 0 also propagates the
 * field: after every SR_OP_DIST, E *= exp(1j*kwave*sqrt(dx^2+dy^2)) (rtm_solver.py:380-418). */
enum {
  SR_OP_DIST = 0,      /* a = d:  x += d*theta, y += d*phi; iarg = 1: without the field factor (see sr_optics) */
  SR_OP_LENS = 1,      /* a = f1, b = f2: theta -= x/f1, phi -= y/f2 */
  SR_OP_CIRC_AP = 2,   /* a = R: reject x^2+y^2 > R^2 */
  SR_OP_CIRC_STOP = 3, /* a = R: reject x^2+y^2 < R^2 */
  SR_OP_RECT_AP = 4,   /* a = Lx, b = Ly: reject x^2 > Lx^2 AND y^2 > Ly^2 (rtm_solver.py:114-117) */
  SR_OP_KNIFE = 5,     /* a = offset, b = direction (>0 rejects above, <0 below), iarg = row (0 x, 2 y) */
  SR_OP_SCALE = 6,     /* a = s: x *= s, y *= s  (m_to_mm: s = 1e3, rtm_solver.py:48-51; mm_to_m: 1e-3) */
  SR_OP_PHASE = 7      /* a = d: the field factor of SR_OP_DIST(d) without moving the ray (diagnostics.py:505-511) */
};
typedef struct {
  int32_t op;
  int32_t iarg;
  double a;
  double b;
} sr_optic;
#define SR_MAX_OPTICS 32
int sr_optics(const sr_optic *chain, int n_ops, double kwave, int64_t n_rays, const double *r_in,
              const double *E_in, double *r_out, double *E_out);

/* ---- A9: Rays.histogram (np.histogram2d) --------------------------------------
 * replaces rtm_solver.py:156-178, diagnostics.py:323-353.  Exact integer counts,
 * H[ny_bins][nx_bins]; numpy's edge rules (linspace edges, right-open bins, last edge closed,
 * NaN and outliers dropped). */
int sr_hist2d(const double *x, const double *y, int64_t n_rays, int nx_bins, int ny_bins,
              double x_lo, double x_hi, double y_lo, double y_hi, uint32_t *H);

/* ---- A10: Interferometry.interferogram ----------------------------------------
 * replaces rtm_solver.py:424-453, diagnostics.py:358-379.  n?_edges = pix // bin_scale edges,
 * n?_edges-1 bins, idx = digitize-1 (right edge open); amp is [2][ny_edges-1][nx_edges-1]
 * complex128: the per-pixel sums of E_x and E_y before H = sqrt(Re(Ax)^2 + Re(Ay)^2). */
int sr_interferogram(const double *x, const double *y, const double *E, int64_t n_rays,
                     int nx_edges, int ny_edges, double x_lo, double x_hi, double y_lo,
                     double y_hi, double *amp /* may be NULL */, double *H /* may be NULL: [ny-1][nx-1] */);

/* ---- A11: Interferometry.interfere_ref_beam (diagnostics.py:559-581) ----------- */
int sr_interfere_ref_beam(const double *x, const double *y, int64_t n_rays, double n_fringes,
                          double deg, double *E /* (2, N) complex128, in place */);

/* ---- fused, device-resident deposit -------------------------------------------
 * trace output (rf, Jf in HBM) -> m_to_mm -> [reference beam] -> optic chain -> image. */
#define SR_IMG_COUNTS 0  /* uint32 [ny][nx], A9 binning; nx, ny = number of bins */
#define SR_IMG_COMPLEX 1 /* float64 [2][ny-1][nx-1][2], A10 binning; nx, ny = number of EDGES */
int sr_image_create(sr_image **out, int kind, int nx, int ny, double x_lo, double x_hi,
                    double y_lo, double y_hi);
int sr_image_zero(sr_image *img);
int sr_image_download(const sr_image *img, void *host); /* uint32 or float64 buffer, see kind */
/* SR_IMG_COUNTS only: the counts as float64 [ny][nx] -- the dtype np.histogram2d hands back (rtm_solver.py:171-174);
 * converted on the device, so the host does not pay an astype over the 8.9e6 pixels of the default detector */
int sr_image_counts_f64(const sr_image *img, double *H);
/* SR_IMG_COMPLEX only: H = sqrt(Re(Ax)^2 + Re(Ay)^2), [ny-1][nx-1] float64 (rtm_solver.py:450) */
int sr_image_amplitude(const sr_image *img, double *H);
int64_t sr_image_bytes(const sr_image *img);
void sr_image_destroy(sr_image *img);

#define SR_MAX_REF_BEAMS 4
typedef struct {
  double kwave;         /* > 0: propagate E through the chain (interferometry) */
  /* reference beams (A11, diagnostics.py:559-581) added to E_y before the chain, in this order: the first ref_on entries.
   * The JAX generation's two_lens_solve adds (10, 20) by itself (diagnostics.py:616) after whatever the caller added. */
  double ref_n_fringes[SR_MAX_REF_BEAMS];
  double ref_deg[SR_MAX_REF_BEAMS];
  int32_t ref_on;       /* number of reference beams, 0..SR_MAX_REF_BEAMS */
  int32_t lds_tiles;    /* 1: LDS-privatised detector tiles; 0: global atomics only */
  int32_t exact_counts; /* SR_IMG_COUNTS, rays traced with SR_PREC_MIXED on a whole volume: 1 (the default with p == NULL) =
                           every ray whose bin, or the decision of a mask of the chain, could differ from the float64
                           build's inside the tracer's per-ray error bound is traced AGAIN in float64 (from s0, same
                           slots of sf / rf / Jf) and counted after that: the image equals the SR_PREC_F64 image integer
                           for integer (np.histogram2d of the reference's rays, rtm_solver.py:156-178).  Needs the
                           volume of that trace to be alive.  0 = count the mixed build's coordinates as they are. */
  int32_t reserved;
} sr_deposit_params;
typedef struct {
  double kernel_ms;     /* HIP-event time of the deposit (with exact_counts: the re-trace included) */
  int64_t deposited;    /* rays that landed inside the detector */
  int64_t retraced;     /* exact_counts: rays traced again in float64 by this deposit */
} sr_deposit_stats;
int sr_rays_deposit(const sr_rays *r, const sr_optic *chain, int n_ops, const sr_deposit_params *p,
                    sr_image *img, sr_deposit_stats *stats);
/* The same front end without the detector: exit-plane rays in HBM -> m_to_mm -> [reference beams] -> chain, written to HOST
 * arrays in the ORIGINAL ray order -- what Rays.rf (rtm_solver.py:197-286) / Diagnostic.rf, .Jf (diagnostics.py:388-481,
 * 614-638) hold after a *_solve().  The mirror classes keep the bundle ScalarDomain.solve left in HBM, deposit from it
 * (sr_rays_deposit) and call this only when a caller READS .rf / .rE.  n_ops == 0: r0 = m_to_mm(rf) itself.  p: kwave and
 * the reference beams (may be NULL); E_out (2, N) complex128 may be NULL. */
int sr_rays_optics(const sr_rays *r, const sr_optic *chain, int n_ops, const sr_deposit_params *p,
                   double *rf_out, double *E_out);
/* The edge guard of exact_counts for SEVERAL counts diagnostics at once: every ray of a mixed-precision trace whose bin or
 * mask decision is uncertain for ANY of the n_diag (chain, image) pairs is traced again in float64, ONE re-trace for all of
 * them (a re-trace costs the latency of a whole trace however few rays it holds).  Afterwards those rays carry bound 0, so
 * the deposits that follow find nothing left to refine, with exact_counts = 1 or 0.  Complex images are skipped.  A trace
 * in SR_PREC_F64 needs none: the call returns at once.  *retraced (may be NULL): rays traced again. */
#define SR_MAX_REFINE 4
int sr_rays_refine(const sr_rays *r, int n_diag, const sr_optic *const *chains, const int *n_ops,
                   sr_image *const *imgs, int64_t *retraced);

/* ---- ray-sharded multi-GPU: sum of the per-GPU images (RCCL over xGMI) ---------
 * replaces comm.reduce(sh.H, root=0, op=MPI.SUM): examples/jobs/run_scripts/pvti_trace_mpi.py:169-170,
 * interference_MPI.py:189.  The 128-byte id is made on rank 0 and handed to the other ranks by the
 * caller's launcher (synthpy_amd/_rendezvous.py: a TCP rendezvous over MASTER_ADDR:MASTER_PORT, no torch). */
#define SR_COMM_ID_BYTES 128
int sr_comm_unique_id(void *id128);
int sr_comm_create(sr_comm **out, const void *id128, int rank, int n_ranks);
int sr_image_reduce(sr_image *img, sr_comm *comm, int root); /* in place; root < 0: all-reduce */
/* what the communicator itself reports (ncclCommUserRank / ncclCommCount): the mpi4py analogue is comm.Get_rank() /
 * comm.Get_size(), pvti_trace_mpi.py:24-25 */
int sr_comm_ranks(const sr_comm *comm, int *rank, int *n_ranks);
void sr_comm_destroy(sr_comm *comm);

#ifdef __cplusplus
}
#endif
#endif /* SYNTHRAY_H */
