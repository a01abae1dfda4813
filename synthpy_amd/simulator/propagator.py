"""Host mirror of src/simulator/propagator.py: solve(s0, domain, probing_depth, ...) -> (rf, Jf | None, duration).

    rf, Jf, duration = solve(beam.s0, domain, probing_depth, return_E=True, lwl=1064e-9)

The reference vmaps a per-ray diffrax Tsit5 solve (PIDController rtol=1, atol=1e-5) over XLA devices
(propagator.py:491-599); here the rays go to the MI355X through sr_trace: RK4 from node plane to node
plane, float64 state.  Kept: the call signature, t_end = sqrt(8)*probing_depth/c (:454), the (rf, Jf,
duration) return (:702), JAX row order for y-probing (:223-243).  Accepted and ignored: parallelise,
jitted, save_steps, memory_debug, keep_domain (JAX execution knobs).  Not reproduced: the truncation
of Np to a multiple of the CPU core count (:504), the per-call recomputation of np.gradient inside
dndr (:79-87; the volume is built once and cached on the domain).  The inv_brems / phaseshift / B_on
branches call the interpolator with the wrong signature in the reference (:137-165) and cannot run there;
here they compute what they state: d(amp) = kappa*amp, d(phase) = omega*(n-1), d(pol) = VerdetConst*ne*(B.v)
with VerdetConst = 2.62e-13*lwl**2 (the legacy solver's, full_solver.py:223).
"""
from __future__ import annotations

from time import time

import numpy as np

from .. import engine, resident

c = engine.c


def omega_pe(ne):
    """Electron plasma frequency, rad/s, ne in cm^-3 (NRL p.28; propagator.py:23-26)."""
    return 5.64e4 * np.sqrt(ne)


def n_refrac(ne, omega):
    """Plasma refractive index (propagator.py:63-64)."""
    return np.sqrt(1.0 - (omega_pe(ne * 1e-6) / omega) ** 2)


def kappa(ne, Te, Z, omega):
    """Inverse-bremsstrahlung rate coefficient [1/s] (NRL formulary; propagator.py:30-60)."""
    ne_cc = np.asarray(ne) * 1e-6
    omega_max = np.maximum(omega_pe(ne_cc), omega)
    L_max = np.maximum(Z * 1.602176634e-19 / Te, 2.760428269727312e-10 / np.sqrt(Te))
    coulomb_log = np.maximum(2.0, np.log(4.19e5 * np.sqrt(Te) / (omega_max * L_max)))
    return 3.1e-5 * Z * c * np.power(ne_cc / omega, 2) * coulomb_log * np.power(Te, -1.5)


def _aux_fields(domain, lwl):
    """What set_up_interps gives dsdt besides the gradients (full_solver.py:276-289; propagator.py:30-60, 137-165): the
    whole-domain arrays (kappa | None, ne | None, B | None, VerdetConst) for engine.Volume.attach_aux."""
    if not (domain.inv_brems or domain.B_on):
        return None
    if domain.inv_brems and (domain.Te is None or domain.Z is None):
        raise ValueError("inv_brems=True needs external_Te() and external_Z()")
    if domain.B_on and domain.B is None:
        raise ValueError("B_on=True needs external_B()")
    omega = 2 * np.pi * c / lwl
    full = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), np.shape(domain.ne)))
    return (kappa(np.asarray(domain.ne, np.float64), full(domain.Te), full(domain.Z), omega) if domain.inv_brems else None,
            full(domain.ne) if domain.B_on else None,
            np.ascontiguousarray(domain.B, np.float64) if domain.B_on else None,
            2.62e-13 * lwl ** 2 if domain.B_on else 0.0)


def _volume_for(domain, lwl):
    key = (float(lwl), domain.probing_direction, bool(domain.phaseshift), id(domain.ne), bool(domain.inv_brems),
           bool(domain.B_on), id(domain.Te), id(domain.Z), id(domain.B))
    cache = getattr(domain, "_volume_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    if domain.ne is None:
        raise ValueError("the domain holds no electron density: pass ne_type= or call external_ne()")
    vol = engine.Volume.from_ne(domain.ne, domain.x, domain.y, domain.z, lwl,
                                probing_direction=domain.probing_direction, phaseshift=domain.phaseshift)
    aux = _aux_fields(domain, lwl)
    if aux is not None:
        vol.attach_aux(*aux)
    domain._volume_cache = (key, vol)
    return vol


def calc_dndr(domain, lwl=1064e-9, keep_domain=False):
    """Build (and cache on the domain) the device volume: gradients of ne/n_c and, with phaseshift, n."""
    return _volume_for(domain, lwl)


def dndr(r, ne, omega, x, y, z):
    """Gradient at the (N, 3) locations r -> (3, N) (propagator.py:66-91)."""
    lwl = 2 * np.pi * c / omega
    vol = engine.Volume.from_ne(ne, x, y, z, lwl)
    try:
        return vol.sample(r)[:3]
    finally:
        vol.close()


def dsdt(t, s, parallelise, inv_brems, phaseshift, B_on, ne, B, Te, Z, x, y, z, omega, VerdetConst, lengths=None, dims=None):
    """The RHS of the ray ODE for the flattened (9N,) state, with the reference's argument list (propagator.py:94-175):
    d(r) = v, d(v) = dndr(r), d(amp) = kappa*amp, d(phase) = omega*(n - 1), d(pol) = VerdetConst*ne*(B.v).  One call builds
    the device volume from `ne`, gathers, and lets it go: it is here so that code written against the reference's RHS (its
    own ODE loop, a check of one stage) runs; solve() never calls it -- the tracer kernels hold the same terms."""
    s = np.asarray(s, dtype=np.float64).reshape(9, -1)
    sprime = np.zeros_like(s)
    lwl = 2 * np.pi * c / omega
    vol = engine.Volume.from_ne(ne, x, y, z, lwl, phaseshift=bool(phaseshift))
    try:
        pts = np.ascontiguousarray(s[:3].T)
        F = vol.sample(pts)
        sprime[:3], sprime[3:6] = s[3:6], F[:3]
        if phaseshift:
            sprime[7] = omega * F[3]
        if inv_brems or B_on:
            full = lambda a: np.ascontiguousarray(np.broadcast_to(np.asarray(a, np.float64), np.shape(ne)))
            vol.attach_aux(kappa(np.asarray(ne, np.float64), full(Te), full(Z), omega) if inv_brems else None,
                           full(ne) if B_on else None, np.ascontiguousarray(B, np.float64) if B_on else None,
                           float(VerdetConst) if B_on else 0.0)
            X = vol.sample_aux(pts)
            if inv_brems:
                sprime[6] = X[0] * s[6]
            if B_on:
                sprime[8] = VerdetConst * X[1] * (X[2] * s[3] + X[3] * s[4] + X[4] * s[5])
    finally:
        vol.close()
    return sprime.flatten()


def ray_to_Jonesvector(rays, ne_extent, *, probing_direction="z", keep_current_plane=False, return_E=False):
    """(9, N) state -> (ray_p (4, N), ray_J (2, N) | None) (propagator.py:178-298)."""
    if keep_current_plane:
        raise NotImplementedError("keep_current_plane=True is only used by the reference's unfinished bkg()")
    return engine.ray_to_jones(rays, ne_extent, probing_direction, engine.ROWS_JAX, return_E=return_E)


def back_propogate(rays, ne_extent, probing_direction):
    """Project the ray states (9, N) back onto the plane <probing axis> = ne_extent (propagator.py:300-349), as written:
    for 'y' the reference stores z' in row 0 and x' in row 2.  Host NumPy: the region loop here hands the rays over ON
    the shared node plane (sr_trace_params.handoff), so nothing on the device path needs this projection."""
    rays = np.array(rays, dtype=np.float64, copy=True)
    x, y, z, vx, vy, vz = (rays[k].copy() for k in range(6))
    if probing_direction == "x":
        t_bp = (x - ne_extent) / vx
        rays[0], rays[1], rays[2] = ne_extent, y - vy * t_bp, z - vz * t_bp
    elif probing_direction == "y":
        t_bp = (y - ne_extent) / vy
        rays[0], rays[1], rays[2] = z - vz * t_bp, ne_extent, x - vx * t_bp
    elif probing_direction == "z":
        t_bp = (z - ne_extent) / vz
        rays[0], rays[1], rays[2] = x - vx * t_bp, y - vy * t_bp, ne_extent
    else:
        print("\nIncorrect probing direction. Use: x, y or z.")
    return rays


def _regions(domain, lwl):
    """How many slabs of node planes the trace goes through: the caller's region_count when it is > 1, else -- with
    auto_batching, the reference's memory-driven split (domain.py:166-199) -- decided ONCE per domain and field set, before
    the volume is built, and kept beside (not in) region_count: a second solve() finds the domain's own whole volume and
    the trace's working set already in HBM, and must not take them for somebody else's memory."""
    regions = max(1, int(getattr(domain, "region_count", 1)))
    if regions > 1 or not getattr(domain, "auto_batching", False):
        return regions
    key = (tuple(int(d) for d in domain.dims), bool(domain.phaseshift), bool(domain.inv_brems), bool(domain.B_on))
    cached = getattr(domain, "_auto_regions", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    whole = getattr(domain, "_volume_cache", None) is not None  # a whole volume of this domain is resident: it fits
    regions = 1 if whole else domain.regions_for_memory()
    domain._auto_regions = (key, regions)
    return regions


def _solve_by_regions(s0, domain, probing_depth, return_E, lwl, substeps, precision, regions):
    """The region loop of propagator.py:366-452: one slab of node planes of the probing axis in HBM at a time, the
    rays handed from slab to slab on the shared planes (engine.Volume.from_ne_slab, HANDOFF_*)."""
    if domain.ne is None:
        raise ValueError("the domain holds no electron density: pass ne_type= or call external_ne()")
    axis = "xyz".index(domain.probing_direction)
    ne = np.asarray(domain.ne)
    cuts = engine.slab_cuts(ne.shape[axis], regions)
    aux = _aux_fields(domain, lwl)  # the optional terms' fields: each slab gets its own node planes of them
    start = time()
    t_end = np.sqrt(8.0) * probing_depth / c
    rays = resident.acquire(s0.shape[1], getattr(domain, "_rays", None)).upload(s0)
    domain._rays = rays
    steps = 0
    for q, (lo, hi) in enumerate(cuts):
        vol = engine.Volume.from_ne_slab(engine.slab_source(ne, axis, lo, hi), domain.x, domain.y, domain.z, lwl,
                                         domain.probing_direction, lo, hi, phaseshift=domain.phaseshift)
        if aux is not None:
            sl = [slice(None)] * 3
            sl[axis] = slice(lo, hi + 1)
            part = lambda a: None if a is None else np.ascontiguousarray(a[tuple(sl)])
            vol.attach_aux(part(aux[0]), part(aux[1]), part(aux[2]), aux[3])
        flags = (engine.HANDOFF_ENTER if q else 0) | (engine.HANDOFF_EXIT if q + 1 < len(cuts) else 0)
        st = rays.trace(vol, t_end, probing_depth, row_order=engine.ROWS_JAX, substeps=substeps, precision=precision, handoff=flags)
        steps += st.ray_steps
        vol.close()
    _, rf, Jf = rays.download(sf=False, Jf=return_E)
    rf, Jf = resident.register(rays, rf, Jf)  # the same memory, write-tracked (resident.TrackedArray)
    duration = time() - start
    solve.last_stats = engine.TraceStats(steps, 0, 0.0, 0.0)
    return rf, Jf, duration


def solve(s0_import, ScalarDomain, probing_depth, *, return_E=False, parallelise=True, jitted=True, save_steps=2,
          memory_debug=False, lwl=1064e-9, keep_domain=False, substeps=1, precision=engine.DEFAULT_PRECISION):
    """Trace the rays s0 (9, N) through the domain and project them onto the exit plane.

    Returns (rf (4, N), Jf (2, N) | None, duration in s)  (propagator.py:351, :702)."""
    s0 = np.asarray(s0_import, dtype=np.float64)
    regions = _regions(ScalarDomain, lwl)
    if regions > 1:
        ScalarDomain._volume_cache = None  # a whole-volume copy from an earlier call must not sit beside the slabs
        return _solve_by_regions(s0, ScalarDomain, probing_depth, return_E, lwl, substeps, precision, regions)
    vol = _volume_for(ScalarDomain, lwl)
    start = time()
    t_end = np.sqrt(8.0) * probing_depth / c
    if s0.ndim != 2 or s0.shape[0] != 9:
        raise ValueError(f"s0 must have shape (9, N), got {s0.shape}")
    # the traced bundle stays in HBM and the Diagnostic classes find it again through the arrays returned here
    # (resident.attach): their histogram() / interferogram() then deposit from it instead of uploading rf again
    rays = resident.acquire(s0.shape[1], getattr(ScalarDomain, "_rays", None)).upload(s0)
    ScalarDomain._rays = rays
    stats = rays.trace(vol, t_end, probing_depth, row_order=engine.ROWS_JAX, substeps=substeps, precision=precision,
                       resident=False)  # rf goes back to the caller, who may bin it: "auto" = float64
    _, rf, Jf = rays.download(sf=False, Jf=return_E)
    rf, Jf = resident.register(rays, rf, Jf)  # the same memory, write-tracked (resident.TrackedArray)
    duration = time() - start
    solve.last_stats = stats
    return rf, Jf, duration
