"""Host mirror of src/simulator/propagator.py: solve(s0, domain, probing_depth, ...) -> (rf, Jf | None, duration).

    rf, Jf, duration = solve(beam.s0, domain, probing_depth, return_E=True, lwl=1064e-9)

The reference vmaps a per-ray diffrax Tsit5 solve (PIDController rtol=1, atol=1e-5) over XLA devices
(propagator.py:491-599); here the rays go to the MI355X through sr_trace: RK4 from node plane to node
plane, float64 state.  Kept: the call signature, t_end = sqrt(8)*probing_depth/c (:454), the (rf, Jf,
duration) return (:702), JAX row order for y-probing (:223-243).  Accepted and ignored: parallelise,
jitted, save_steps, memory_debug, keep_domain (JAX execution knobs).  Not reproduced: the truncation
of Np to a multiple of the CPU core count (:504), the per-call recomputation of np.gradient inside
dndr (:79-87; the volume is built once and cached on the domain), the dead inv_brems/phaseshift/B_on
branches with the wrong interpolator signature (:137-165): phaseshift works, the other two raise.
"""
from __future__ import annotations

from time import time

import numpy as np

from .. import engine

c = engine.c


def omega_pe(ne):
    """Electron plasma frequency, rad/s, ne in cm^-3 (NRL p.28; propagator.py:23-26)."""
    return 5.64e4 * np.sqrt(ne)


def n_refrac(ne, omega):
    """Plasma refractive index (propagator.py:63-64)."""
    return np.sqrt(1.0 - (omega_pe(ne * 1e-6) / omega) ** 2)


def _volume_for(domain, lwl):
    key = (float(lwl), domain.probing_direction, bool(domain.phaseshift), id(domain.ne))
    cache = getattr(domain, "_volume_cache", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    if domain.ne is None:
        raise ValueError("the domain holds no electron density: pass ne_type= or call external_ne()")
    vol = engine.Volume.from_ne(domain.ne, domain.x, domain.y, domain.z, lwl,
                                probing_direction=domain.probing_direction, phaseshift=domain.phaseshift)
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


def ray_to_Jonesvector(rays, ne_extent, *, probing_direction="z", keep_current_plane=False, return_E=False):
    """(9, N) state -> (ray_p (4, N), ray_J (2, N) | None) (propagator.py:178-298)."""
    if keep_current_plane:
        raise NotImplementedError("keep_current_plane=True is only used by the reference's unfinished bkg()")
    return engine.ray_to_jones(rays, ne_extent, probing_direction, engine.ROWS_JAX, return_E=return_E)


def solve(s0_import, ScalarDomain, probing_depth, *, return_E=False, parallelise=True, jitted=True, save_steps=2,
          memory_debug=False, lwl=1064e-9, keep_domain=False, substeps=1, precision=engine.DEFAULT_PRECISION):
    """Trace the rays s0 (9, N) through the domain and project them onto the exit plane.

    Returns (rf (4, N), Jf (2, N) | None, duration in s)  (propagator.py:351, :702)."""
    if ScalarDomain.inv_brems or ScalarDomain.B_on:
        raise NotImplementedError("inv_brems / B_on are not on the GPU path yet (DESIGN.md: next)")
    vol = _volume_for(ScalarDomain, lwl)
    s0 = np.asarray(s0_import, dtype=np.float64)
    start = time()
    t_end = np.sqrt(8.0) * probing_depth / c
    _, rf, Jf, stats = engine.trace(vol, s0, t_end, probing_depth, row_order=engine.ROWS_JAX, substeps=substeps,
                                    precision=precision, return_E=return_E, return_sf=False)
    duration = time() - start
    solve.last_stats = stats
    return rf, Jf, duration
