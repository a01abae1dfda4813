"""Thin object layer over the C ABI (include/synthray.h): device volumes, ray bundles,
detector images and the host-buffer entry points.  Every compute call goes to the HIP
library; nothing here does the path's arithmetic in NumPy.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from dataclasses import dataclass

import numpy as np

from . import _ffi
from ._ffi import check, f32, f64, lib, ptr

c = 299792458.0  # scipy.constants.c, as the reference uses (full_solver.py:93)

# optic op codes (include/synthray.h)
OP_DIST, OP_LENS, OP_CIRC_AP, OP_CIRC_STOP, OP_RECT_AP, OP_KNIFE, OP_SCALE, OP_PHASE = range(8)
ROWS_LEGACY, ROWS_JAX = 0, 1
IMG_COUNTS, IMG_COMPLEX = 0, 1
VOL_PHASE = 1

_AXES = {"x": 0, "y": 1, "z": 2}


def axis_index(probing_direction) -> int:
    """'x' | 'y' | 'z' -> 0 | 1 | 2.  The reference prints and carries on for anything else
    (propagator.py:260, beam.py:288); the engine raises."""
    if isinstance(probing_direction, str) and probing_direction in _AXES:
        return _AXES[probing_direction]
    raise ValueError(f"probing_direction must be 'x', 'y' or 'z', got {probing_direction!r}")


def device_count() -> int:
    _ffi.gpu_touched = True  # the library's first call opens the device: from here on this process must not fork
    n = lib.sr_device_count()
    if n < 0:
        check(n)
    return n


def init(device: int = 0) -> None:
    check(lib.sr_init(int(device)))


def init_rank(local_rank: int = 0, local_world: int = 1, *, device=None, shared=False) -> int:
    """Open this rank's GPU: the one way in for every multi-process driver (bench.py, run_trace, the tests' workers).
    device None: GPU `local_rank`; a job with more local ranks than visible GPUs fails HERE, loudly, instead of putting
    two ranks on device 0 (`shared=True` says that is wanted: the one-GPU rehearsal).  Ranks that start in the same
    instant are handled in the library (sr_device_count: bounded retry of the device open before the first HIP call)."""
    n = device_count()
    if n < 1:
        check(lib.sr_init(0))  # raises with the runtime's reason
    if device is None:
        if shared:
            device = 0
        elif local_world > n:
            raise _ffi.SynthrayError(f"{local_world} local ranks but {n} visible GPU(s): one process per GPU "
                                     "(HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES hide devices)")
        else:
            device = local_rank
    device = int(device)
    if not 0 <= device < n:
        raise ValueError(f"device {device} outside 0..{n - 1}")
    init(device)
    return device


def device_memory():
    """(free, total) bytes of HBM on the selected device."""
    f, t = C.c_int64(0), C.c_int64(0)
    check(lib.sr_device_memory(C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def volume_bytes_estimate(n_nodes, phaseshift=False, inv_brems=False, B_on=False, ne_itemsize=8):
    """HBM a Volume of n_nodes holds (16 B per node packed, +4 with the phase, +8 kappa, +32 {n_e, B})
    plus the staging its construction needs for a moment (the uploaded n_e, float32(n_e / n_c), and for the optional terms the
    uploaded arrays)."""
    held = 16 + (4 if phaseshift else 0) + (8 if inv_brems else 0) + (32 if B_on else 0)
    staging = ne_itemsize + 4 + (8 if inv_brems else 0) + (32 if B_on else 0)
    return int(n_nodes) * (held + staging)


def release_caches():
    """Free what sr_trace keeps on the device between calls (its chunk pipeline's working set, ~1.5 GB of HBM)."""
    check(lib.sr_release_caches())


def synchronize() -> None:
    """Waits for every stream of the library."""
    check(lib.sr_synchronize())


def select_stream(index: int) -> None:
    """Queue the following calls on stream 0 (the default) or 1; see sr_stream_select in include/synthray.h."""
    check(lib.sr_stream_select(int(index)))


def stream_wait(waiting: int, on: int) -> None:
    """Work queued on stream `waiting` after this call starts once what stream `on` holds now is done (sr_stream_wait)."""
    check(lib.sr_stream_wait(int(waiting), int(on)))


def default_t_end(extent: float) -> float:
    """t = sqrt(8)*extent/c: long enough for every ray to leave the volume (full_solver.py:381)."""
    return float(np.sqrt(8.0) * extent / c)


@dataclass
class TraceStats:
    ray_steps: int = 0
    fallback_rays: int = 0
    trace_kernel_ms: float = 0.0
    total_ms: float = 0.0


PRECISIONS = {"f64": 0, "mixed": 1}
DEFAULT_PRECISION = "auto"


def resolve_precision(precision, volume, *, resident=True, handoff=0, substeps=1) -> str:
    """"auto" (the default) picks the build by what the trace is for, so that EVERY default path gives the images of the
    float64 build (= the oracle's from the same s0: counts integer for integer, interferograms to 1e-5 of their maximum).

    * A volume built with the phase integral (phaseshift=True: the Jones vector feeds Interferometry) is traced in
      float64: the reference's field propagation multiplies E by exp(i*k*|dr|) with k = 2*pi/wavelength[m] against |dr|
      in mm (rtm_solver.py:380-384), 2.4e9 rad per radian of exit angle over a 400 mm leg, so only the float64 build
      (1e-14 rad from the oracle) reproduces an interferogram from the same rays
      (tests/test_gpu_parity.py::test_interferometry_end_to_end_from_s0).
    * Without the phase the detector images are counts (shadowgraphy, schlieren, refractometry).  Rays that stay in HBM
      (RayBundle.trace -> RayBundle.deposit) are traced by the mixed build (float32 stage arithmetic, 5e-11 m / 2e-8 rad
      from the oracle, twice as fast) and the deposit's EDGE GUARD traces again in float64 the ~1 % of them whose pixel
      or mask decision is not certain within the tracer's own per-ray error bound (sr_deposit_params.exact_counts).
    * Where no guard can follow -- host arrays out (`resident=False`: trace(), ScalarDomain.solve: the caller bins rf
      itself), slab hand-offs (a slab holds neither the rays' start nor the other planes) -- "auto" is float64; so it is with
      sub-steps and the optional terms, which the mixed build has no kernel for ("mixed" asked for by name runs the float64
      kernels there too)."""
    if precision in (None, "auto"):
        if getattr(volume, "phase", False) or not resident or handoff or substeps != 1 or getattr(volume, "aux", False):
            return "f64"
        return "mixed"
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be 'auto' or one of {sorted(PRECISIONS)}, got {precision!r}")
    return precision


HANDOFF_ENTER, HANDOFF_EXIT = 1, 2


def _trace_params(t_end, extent, axis, row_order, substeps, sort_rays, precision, dt=0.0, handoff=0):
    """precision "f64": every operation in float64 (the parity build: differs from the oracle by fused
    multiply-adds only, ~1e-17 m).  "mixed" (default): float64 state, stage positions and accumulation,
    float32 weights / blend / RK4 slopes: within 5e-11 m, 2e-8 rad and 4e-8 of the phase of the f64 build on
    the 512^3 benchmark, twice to three times as fast."""
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be 'auto' or one of {sorted(PRECISIONS)}, got {precision!r}")
    return _ffi.TraceParams(float(t_end), float(extent), float(dt), int(axis), int(row_order), int(substeps),
                            1 if sort_rays else 0, PRECISIONS[precision], int(handoff))


def deposit_params(kwave=0.0, ref_beam=None, lds_tiles=True, exact_counts=True):
    """sr_deposit_params.  ref_beam: None, one (n_fringes, deg) pair, or a list of up to MAX_REF_BEAMS pairs added in order."""
    refs = [] if ref_beam is None else ([tuple(ref_beam)] if np.isscalar(ref_beam[0]) else [tuple(b) for b in ref_beam])
    if len(refs) > _ffi.MAX_REF_BEAMS:
        raise ValueError(f"at most {_ffi.MAX_REF_BEAMS} reference beams per deposit, got {len(refs)}")
    p = _ffi.DepositParams()
    p.kwave = float(kwave)
    for q, (n_fringes, deg) in enumerate(refs):
        p.ref_n_fringes[q], p.ref_deg[q] = float(n_fringes), float(deg)
    p.ref_on, p.lds_tiles, p.exact_counts, p.reserved = len(refs), 1 if lds_tiles else 0, 1 if exact_counts else 0, 0
    return p


def make_chain(ops):
    """[(op, a[, b[, iarg]]), ...] -> ctypes array of sr_optic."""
    arr = (_ffi.Optic * max(1, len(ops)))()
    for k, o in enumerate(ops):
        arr[k].op = int(o[0])
        arr[k].a = float(o[1]) if len(o) > 1 else 0.0
        arr[k].b = float(o[2]) if len(o) > 2 else 0.0
        arr[k].iarg = int(o[3]) if len(o) > 3 else 0
    return arr


class Volume:
    """Device-resident fields of one ScalarDomain: the result of calc_dndr (+ n_refrac)."""

    def __init__(self, handle, shape, axis, phase=False):
        self._h = handle
        self.shape = tuple(int(s) for s in shape)
        self.axis = axis
        self.phase = bool(phase)  # the n-1 field is resident: the trace integrates the phase (A5)
        self.aux = False          # attach_aux: kappa / Faraday fields resident

    @classmethod
    def from_ne(cls, ne, x, y, z, lwl, probing_direction="z", phaseshift=False):
        ne = np.asarray(ne)
        if ne.dtype != np.float32:
            ne = f64(ne)
        ne = np.ascontiguousarray(ne)
        x, y, z = f32(x), f32(y), f32(z)
        if ne.shape != (len(x), len(y), len(z)):
            raise ValueError(f"ne has shape {ne.shape}, coordinates give {(len(x), len(y), len(z))}")
        axis = axis_index(probing_direction)
        h = C.c_void_p()
        check(lib.sr_volume_create(C.byref(h), ptr(ne), 0 if ne.dtype == np.float32 else 1, len(x), len(y), len(z),
                                   ptr(x), ptr(y), ptr(z), float(lwl), axis, VOL_PHASE if phaseshift else 0))
        return cls(h, ne.shape, axis, phaseshift)

    @classmethod
    def from_ne_slab(cls, ne_slab, x, y, z, lwl, probing_direction, k_lo, k_hi, phaseshift=False):
        """Node planes k_lo..k_hi of the probing axis as a volume of their own (A12; BASELINE config 5).  x, y, z are
        the WHOLE domain's coordinates; ne_slab holds planes max(k_lo-1, 0)..min(k_hi+1, n-1) of the probing axis (see
        slab_source) so that the gradients equal the whole domain's bit for bit."""
        x, y, z = f32(x), f32(y), f32(z)
        axis = axis_index(probing_direction)
        n = (len(x), len(y), len(z))
        h_lo, h_hi = max(k_lo - 1, 0), min(k_hi + 1, n[axis] - 1)
        want = tuple(h_hi - h_lo + 1 if k == axis else n[k] for k in range(3))
        ne_slab = np.asarray(ne_slab)
        if ne_slab.dtype != np.float32:
            ne_slab = f64(ne_slab)
        ne_slab = np.ascontiguousarray(ne_slab)
        if ne_slab.shape != want:
            raise ValueError(f"ne_slab has shape {ne_slab.shape}; planes {h_lo}..{h_hi} of the domain give {want}")
        h = C.c_void_p()
        check(lib.sr_volume_create_slab(C.byref(h), ptr(ne_slab), 0 if ne_slab.dtype == np.float32 else 1, *n, ptr(x), ptr(y),
                                        ptr(z), float(lwl), axis, VOL_PHASE if phaseshift else 0, int(k_lo), int(k_hi)))
        shape = tuple(k_hi - k_lo + 1 if k == axis else n[k] for k in range(3))
        return cls(h, shape, axis, phaseshift)

    @classmethod
    def from_fields(cls, dndx, dndy, dndz, x, y, z, omega, probing_direction="z", nref=None):
        dndx, dndy, dndz = f32(dndx), f32(dndy), f32(dndz)
        x, y, z = f32(x), f32(y), f32(z)
        shape = (len(x), len(y), len(z))
        for a in (dndx, dndy, dndz):
            if a.shape != shape:
                raise ValueError(f"gradient volume has shape {a.shape}, coordinates give {shape}")
        nref = None if nref is None else f64(nref)
        axis = axis_index(probing_direction)
        h = C.c_void_p()
        check(lib.sr_volume_create_from_fields(C.byref(h), ptr(dndx), ptr(dndy), ptr(dndz), ptr(nref), float(omega),
                                               *shape, ptr(x), ptr(y), ptr(z), axis))
        return cls(h, shape, axis, nref is not None)

    @property
    def omega(self) -> float:
        return float(lib.sr_volume_omega(self._h))

    @property
    def nbytes(self) -> int:
        return int(lib.sr_volume_bytes(self._h))

    def fields(self, phase=False):
        """(dndx, dndy, dndz[, n-1]) read back in the reference's layout."""
        out = [np.empty(self.shape, np.float32) for _ in range(3)]
        nm1 = np.empty(self.shape, np.float64) if phase else None
        check(lib.sr_volume_fields(self._h, ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(nm1)))
        return (*out, nm1) if phase else tuple(out)

    def sample(self, pts):
        """Interpolated (dndx, dndy, dndz, n-1) at pts (N,3): the gathers of dsdt, shape (4, N)."""
        pts = f64(pts).reshape(-1, 3)
        out = np.empty((4, len(pts)))
        check(lib.sr_volume_sample(self._h, ptr(pts), len(pts), ptr(out)))
        return out

    def attach_aux(self, kappa=None, ne=None, B=None, verdet=0.0):
        """The optional terms of dsdt (what set_up_interps builds, full_solver.py:276-289): kappa (nx,ny,nz) [1/s] for
        d(amp) = kappa*amp; ne (nx,ny,nz) and B (nx,ny,nz,3) with VerdetConst for d(pol) = VerdetConst*ne*(B.v)."""
        if kappa is not None:
            kappa = f64(kappa)
            if kappa.shape != self.shape:
                raise ValueError(f"kappa has shape {kappa.shape}, the volume {self.shape}")
        if (ne is None) != (B is None):
            raise ValueError("ne and B go together")
        if B is not None:
            ne, B = f64(ne), f64(B)
            if ne.shape != self.shape or B.shape != self.shape + (3,):
                raise ValueError(f"ne {ne.shape} / B {B.shape} do not match the volume {self.shape} (+ (3,))")
        check(lib.sr_volume_attach_aux(self._h, ptr(kappa), ptr(ne), ptr(B), float(verdet)))
        self.aux = kappa is not None or B is not None
        return self

    def sample_aux(self, pts):
        """Interpolated (kappa, ne, Bx, By, Bz) at pts (N,3): the gathers of atten / get_ne / get_B, shape (5, N)."""
        pts = f64(pts).reshape(-1, 3)
        out = np.empty((5, len(pts)))
        check(lib.sr_volume_sample_aux(self._h, ptr(pts), len(pts), ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib.sr_volume_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _PinnedBlock:
    """One page-locked host block (sr_host_alloc) under a NumPy array: the array's base; goes back to the pool when the
    last array over it is collected."""
    __slots__ = ("ptr", "nbytes", "__array_interface__", "__weakref__")

    def __init__(self, ptr_, nbytes, shape, dtype):
        self.ptr, self.nbytes = ptr_, nbytes
        self.__array_interface__ = {"shape": shape, "typestr": np.dtype(dtype).str, "data": (ptr_, False), "version": 3}

    def __del__(self):
        try:
            _pinned_release(self.ptr, self.nbytes)
        except Exception:  # interpreter shutting down
            pass


PINNED_MIN_BYTES = 8 << 20    # smaller result arrays are ordinary NumPy arrays
PINNED_POOL_BYTES = 4 << 30   # free blocks kept for the next call; beyond this they are given back (sr_host_free)
# "auto": a size is page-locked from the SECOND time it is asked for (locking 1.4 GB costs ~270 ms, four times what it
# saves per call: a script that traces once keeps NumPy's pageable arrays, a loop gets the fast ones); "1" always, "0" never
PINNED_RESULTS = os.environ.get("SYNTHRAY_PINNED_RESULTS", "auto")
_pinned_free, _pinned_held, _pinned_asked = {}, [0], {}
_pinned_live = [0]  # bytes of page-locked memory under arrays the caller still holds


def _mem_total():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal:"):
                return int(line.split()[1]) * 1024
    except (OSError, ValueError):
        pass
    return 64 << 30


# Page-locked memory is taken from what the host can swap or reclaim: a loop that KEEPS its results (1.4 GB per 1e7 rays)
# must not pin the machine down.  Live + pooled page-locked bytes stay under this cap (a quarter of the host's memory unless
# SYNTHRAY_PINNED_MAX_BYTES says otherwise); beyond it the arrays are ordinary NumPy arrays again.
PINNED_MAX_BYTES = int(float(os.environ.get("SYNTHRAY_PINNED_MAX_BYTES", 0)) or _mem_total() // 4)


def _pinned_release(p, nbytes):
    _pinned_live[0] -= nbytes
    if lib is None:  # interpreter shutting down after the pool was emptied
        return
    if _pinned_held[0] + nbytes <= PINNED_POOL_BYTES:
        _pinned_free.setdefault(nbytes, []).append(p)
        _pinned_held[0] += nbytes
    else:
        lib.sr_host_free(p)


def pinned_empty(shape, dtype=np.float64):
    """np.empty(shape, dtype) over page-locked memory when the array is large: the GPU writes it at link speed (pageable
    memory: a third of that, and the copy holds the host thread).  An ordinary, writeable NumPy array for the caller;
    blocks are recycled between calls (allocating page-locked memory costs more than the copy it speeds up)."""
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if nbytes < PINNED_MIN_BYTES or PINNED_RESULTS == "0":
        return np.empty(shape, dtype)
    _pinned_asked[nbytes] = _pinned_asked.get(nbytes, 0) + 1
    if PINNED_RESULTS == "auto" and _pinned_asked[nbytes] < 2 and not _pinned_free.get(nbytes):
        return np.empty(shape, dtype)
    blocks = _pinned_free.get(nbytes)
    if blocks:
        p = blocks.pop()
        _pinned_held[0] -= nbytes
    else:
        if _pinned_live[0] + _pinned_held[0] + nbytes > PINNED_MAX_BYTES:  # the caller keeps its results: no more pinning
            return np.empty(shape, dtype)
        h = C.c_void_p()
        if lib.sr_host_alloc(C.byref(h), nbytes) != 0 or not h.value:  # no page-locked memory left: an ordinary array does
            return np.empty(shape, dtype)
        p = h.value
    _pinned_live[0] += nbytes
    return np.asarray(_PinnedBlock(p, nbytes, tuple(shape), dtype))


def _pinned_drain():
    """At exit: the pooled blocks go back before the library is unloaded (blocks still under live arrays are the process's
    to lose)."""
    for blocks in _pinned_free.values():
        while blocks:
            lib.sr_host_free(blocks.pop())
    _pinned_held[0] = 0


import atexit  # noqa: E402

atexit.register(_pinned_drain)


def trace(volume: Volume, s0, t_end, extent, *, row_order=ROWS_LEGACY, substeps=1, sort_rays=True,
          precision=DEFAULT_PRECISION, return_E=True, return_sf=True, dt=0.0):
    """ScalarDomain.solve / propagator.solve on host arrays: s0 (9,N) -> (sf, rf, Jf, stats).  A large bundle goes through
    in chunks on two streams (sr_trace), its result arrays page-locked: transfers run beside the trace."""
    s0 = f64(s0)
    if s0.ndim != 2 or s0.shape[0] != 9:
        raise ValueError(f"s0 must have shape (9, N), got {s0.shape}")
    N = s0.shape[1]
    sf = pinned_empty((9, N)) if return_sf else None
    rf = pinned_empty((4, N))
    Jf = pinned_empty((2, N), np.complex128) if return_E else None
    p = _trace_params(t_end, extent, volume.axis, row_order, substeps, sort_rays,
                      resolve_precision(precision, volume, resident=False, substeps=substeps), dt)
    st = _ffi.TraceStats()
    check(lib.sr_trace(volume._h, ptr(s0), N, C.byref(p), ptr(sf), ptr(rf), ptr(Jf), C.byref(st)))
    return sf, rf, Jf, TraceStats(st.ray_steps, st.fallback_rays, st.trace_kernel_ms, st.total_ms)


def ray_to_jones(sf, extent, probing_direction="z", row_order=ROWS_LEGACY, return_E=True):
    sf = f64(sf)
    N = sf.shape[1]
    rf = np.empty((4, N))
    Jf = np.empty((2, N), np.complex128) if return_E else None
    check(lib.sr_ray_to_jones(ptr(sf), N, float(extent), axis_index(probing_direction), row_order, ptr(rf), ptr(Jf)))
    return rf, Jf


class RayBundle:
    """Rays resident in HBM: upload once, trace, deposit on any number of detectors."""

    def __init__(self, n_rays: int):
        self.n = int(n_rays)
        self._h = C.c_void_p()
        self._volume, self.retraced = None, 0
        self.generation = 0              # counts every call that changes the rays: what a resident handle is checked against
        self.holders = weakref.WeakSet()  # diagnostics objects that deposit from this bundle (resident.attach)
        check(lib.sr_rays_create(C.byref(self._h), self.n))

    @property
    def alive(self) -> bool:
        return bool(getattr(self, "_h", None))

    def upload(self, s0):
        s0 = f64(s0)
        if s0.shape != (9, self.n):
            raise ValueError(f"s0 must have shape (9, {self.n}), got {s0.shape}")
        self.generation += 1
        check(lib.sr_rays_upload(self._h, ptr(s0)))
        return self

    def upload_part(self, s0, first, last=False):
        """(9, n) host rays become rays first .. first + n - 1 of the bundle; last=True on the final part (sr_rays_upload_part:
        a bundle merged from several host chunks, each its own seeded draw)."""
        s0 = f64(s0)
        if s0.ndim != 2 or s0.shape[0] != 9 or first < 0 or first + s0.shape[1] > self.n:
            raise ValueError(f"part of shape {s0.shape} at ray {first} does not fit a bundle of {self.n} rays")
        self.generation += 1
        check(lib.sr_rays_upload_part(self._h, ptr(s0), int(s0.shape[1]), int(first), 1 if last else 0))
        return self

    def generate(self, beam_size, divergence, ne_extent, beam_type="circular", probing_direction="z", seed=0, first_ray=0,
                 radial_law="legacy"):
        """Draw the bundle on the device: init_beam's distributions from a Philox stream keyed by (seed, first_ray + ray
        index).  Not NumPy's sample (use init_beam + upload to reproduce the reference's seeded rays).  radial_law
        ('circular' only): "legacy" = u = U + U folded at 1 (full_solver.py:567-572), "power" = np.random.power(2) of the
        JAX generation's Beam (src/simulator/beam.py:66-77)."""
        if beam_type == "circular":
            if radial_law not in ("legacy", "power"):
                raise ValueError("radial_law must be 'legacy' or 'power'")
            kind, a, b = (0 if radial_law == "legacy" else 3), float(beam_size), 0.0
        elif beam_type == "linear":
            kind, a, b = 2, float(beam_size), 0.0
        elif beam_type == "square":
            kind, a, b = 1, float(beam_size), float(beam_size)
        elif beam_type == "rectangular":
            kind, a, b = 1, float(beam_size[0]), float(beam_size[1])
        else:
            raise ValueError(f"beam_type {beam_type!r}: 'circular', 'square', 'rectangular' or 'linear' on the device")
        self.generation += 1
        check(lib.sr_rays_generate(self._h, kind, a, b, float(divergence), float(ne_extent), axis_index(probing_direction),
                                   int(seed), int(first_ray)))
        return self

    def download_s0(self):
        """The bundle as generated / uploaded, (9, N)."""
        s0 = np.empty((9, self.n))
        check(lib.sr_rays_download_s0(self._h, ptr(s0)))
        return s0

    def trace(self, volume: Volume, t_end, extent, *, row_order=ROWS_LEGACY, substeps=1, sort_rays=True,
              precision=DEFAULT_PRECISION, dt=0.0, want_stats=True, handoff=0, resident=True) -> TraceStats:
        """handoff (slab volumes, A12): HANDOFF_ENTER takes the state from the hand-off records instead of s0,
        HANDOFF_EXIT leaves it in the records instead of writing sf / rf / Jf.  resident=False: the caller is going to
        download rf and bin it itself, so precision "auto" may not count on the deposit's edge guard (resolve_precision)."""
        p = _trace_params(t_end, extent, volume.axis, row_order, substeps, sort_rays,
                          resolve_precision(precision, volume, resident=resident, handoff=handoff, substeps=substeps), dt, handoff)
        st = _ffi.TraceStats()
        self._volume = volume  # an exact-counts deposit may trace some rays again: the volume lives as long as the bundle needs it
        self.generation += 1
        check(lib.sr_rays_trace(self._h, volume._h, C.byref(p), C.byref(st) if want_stats else None))
        return TraceStats(st.ray_steps, st.fallback_rays, st.trace_kernel_ms, st.total_ms)

    @property
    def tile_segments(self) -> int:
        """Node-plane segments of the tile path (trace_tile.inc) in the last trace of this bundle; 0: the per-ray kernels."""
        return int(lib.sr_rays_tile_segments(self._h))

    @property
    def tile_records(self) -> bool:
        """True when the last trace's tile path ran the records kernel (sr_rays_tile_records)."""
        return bool(lib.sr_rays_tile_records(self._h))

    @property
    def bbox(self):
        """(min x, y, z, max x, y, z) of the launch positions of the beam these rays belong to [m], or None: what the library
        judges the ray density by when it picks the kernel (sr_rays_get_bbox)."""
        box, known = np.zeros(6), C.c_int(0)
        check(lib.sr_rays_get_bbox(self._h, ptr(box), C.byref(known)))
        return box if known.value else None

    @bbox.setter
    def bbox(self, box):
        """Name the beam of rays that arrive by hand-off (ranks > 0 of a slab pipeline): stays with the bundle across
        hand-offs until the next upload / generate; None takes it back (sr_rays_set_bbox)."""
        check(lib.sr_rays_set_bbox(self._h, None if box is None else ptr(f64(np.asarray(box, dtype=np.float64).reshape(6)))))

    def error_bound(self):
        """(N,) float32: per ray, the bound [rad] on the exit-angle difference to the float64 build (sr_rays_error_bound)."""
        out = np.empty(self.n, np.float32)
        check(lib.sr_rays_error_bound(self._h, ptr(out)))
        return out

    def trace_stats(self) -> TraceStats:
        """Totals of every trace of this bundle since its counters were last read (traces run with want_stats=False
        queue their work and return; this waits for them).  Times are those of the last trace."""
        st = _ffi.TraceStats()
        check(lib.sr_rays_trace_stats(self._h, C.byref(st)))
        return TraceStats(st.ray_steps, st.fallback_rays, st.trace_kernel_ms, st.total_ms)

    def handoff_download(self):
        """(10, N) records in launch order: p_b, p_c, v_a, v_b, v_c, phase, t, amp, pol, ray index."""
        rec = np.empty((10, self.n))
        check(lib.sr_rays_handoff_download(self._h, ptr(rec)))
        return rec

    def handoff_upload(self, rec):
        rec = f64(rec)
        if rec.shape != (10, self.n):
            raise ValueError(f"records must have shape (10, {self.n}), got {rec.shape}")
        self.generation += 1
        check(lib.sr_rays_handoff_upload(self._h, ptr(rec)))
        return self

    def handoff_send(self, comm, peer):
        check(lib.sr_rays_handoff_send(self._h, comm, int(peer)))

    def handoff_recv(self, comm, peer):
        self.generation += 1
        check(lib.sr_rays_handoff_recv(self._h, comm, int(peer)))

    def download(self, sf=True, rf=True, Jf=True):
        a = pinned_empty((9, self.n)) if sf else None
        b = pinned_empty((4, self.n)) if rf else None
        e = pinned_empty((2, self.n), np.complex128) if Jf else None
        check(lib.sr_rays_download(self._h, ptr(a), ptr(b), ptr(e)))
        return a, b, e

    def deposit(self, image: "DetectorImage", ops, *, kwave=0.0, ref_beam=None, lds_tiles=True, want_stats=True,
                exact_counts=True):
        """m_to_mm -> [reference beam] -> chain -> image.  exact_counts (counts images of a mixed-precision trace): the
        edge guard of sr_deposit_params; `self.retraced` then holds how many rays it traced again in float64."""
        chain = make_chain(ops)
        p = deposit_params(kwave, ref_beam, lds_tiles, exact_counts)
        st = _ffi.DepositStats()
        check(lib.sr_rays_deposit(self._h, chain, len(ops), C.byref(p), image._h, C.byref(st) if want_stats else None))
        self.retraced = int(st.retraced)
        return st.kernel_ms, int(st.deposited)

    def optics(self, ops=(), *, kwave=0.0, ref_beam=None, with_E=False):
        """The deposit's front end without the detector (sr_rays_optics): exit-plane rays -> m_to_mm -> [reference beams] ->
        chain, as HOST arrays in the original ray order: (r (4, N) mm, E (2, N) | None).  ops == (): r0 = m_to_mm(rf)."""
        ops = list(ops)
        r = pinned_empty((4, self.n))
        E = pinned_empty((2, self.n), np.complex128) if with_E else None
        p = deposit_params(kwave, ref_beam)
        check(lib.sr_rays_optics(self._h, make_chain(ops), len(ops), C.byref(p), ptr(r), ptr(E)))
        return r, E

    def refine(self, diagnostics, want_stats=True):
        """The exact-counts edge guard for several diagnostics at once: [(DetectorImage, ops), ...] (at most 4; complex images
        are ignored).  ONE float64 re-trace of every ray whose pixel or mask decision is uncertain for any of them; the
        deposits that follow then have nothing left to refine.  Returns the number of rays traced again (want_stats=False:
        queued without waiting, returns 0)."""
        diagnostics = list(diagnostics)
        n = len(diagnostics)
        if n == 0:
            return 0
        keep = [make_chain(ops) for _, ops in diagnostics]
        chains = (C.c_void_p * n)(*[C.cast(k, C.c_void_p) for k in keep])
        n_ops = (C.c_int * n)(*[len(ops) for _, ops in diagnostics])
        imgs = (C.c_void_p * n)(*[img._h for img, _ in diagnostics])
        again = C.c_int64(0)
        check(lib.sr_rays_refine(self._h, n, chains, n_ops, imgs, C.byref(again) if want_stats else None))
        self.retraced = int(again.value)
        return self.retraced

    def close(self):
        for img in self.__dict__.pop("_images", {}).values():  # the detectors resident.DeviceRays kept with the bundle
            img.close()
        if getattr(self, "_h", None):
            lib.sr_rays_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DetectorImage:
    """A detector image in HBM.  kind IMG_COUNTS: nx, ny are bins (A9); IMG_COMPLEX: edges (A10)."""

    def __init__(self, kind, nx, ny, x_lo, x_hi, y_lo, y_hi):
        self.kind, self.nx, self.ny = kind, int(nx), int(ny)
        self.range = (float(x_lo), float(x_hi), float(y_lo), float(y_hi))
        self._h = C.c_void_p()
        check(lib.sr_image_create(C.byref(self._h), kind, self.nx, self.ny, *self.range))

    @classmethod
    def counts(cls, bin_scale=1, pix_x=3448, pix_y=2574, Lx=18.0, Ly=13.5):
        """Detector of Rays.histogram (rtm_solver.py:156-178; KAF-8300 defaults)."""
        return cls(IMG_COUNTS, pix_x // bin_scale, pix_y // bin_scale, -Lx / 2, Lx / 2, -Ly / 2, Ly / 2)

    @classmethod
    def complex_field(cls, bin_scale=1, pix_x=3448, pix_y=2574, Lx=18.0, Ly=13.5):
        """Detector of Interferometry.interferogram: edges linspace(-L//2, L//2, pix//bin_scale) — floor
        division as written in the reference (rtm_solver.py:436-437): x in [-9, 9], y in [-7, 6] for the defaults."""
        return cls(IMG_COMPLEX, pix_x // bin_scale, pix_y // bin_scale, -Lx // 2, Lx // 2, -Ly // 2, Ly // 2)

    def zero(self):
        check(lib.sr_image_zero(self._h))

    def download(self):
        if self.kind == IMG_COUNTS:
            H = np.empty((self.ny, self.nx), np.uint32)
        else:
            H = np.empty((2, self.ny - 1, self.nx - 1), np.complex128)
        check(lib.sr_image_download(self._h, ptr(H)))
        return H

    def counts_f64(self):
        """IMG_COUNTS as float64 [ny][nx], np.histogram2d's dtype (rtm_solver.py:171-174): converted on the device."""
        H = pinned_empty((self.ny, self.nx))
        check(lib.sr_image_counts_f64(self._h, ptr(H)))
        return H

    def amplitude(self):
        """IMG_COMPLEX: H = sqrt(Re(Ax)^2 + Re(Ay)^2) (rtm_solver.py:450)."""
        H = pinned_empty((self.ny - 1, self.nx - 1))
        check(lib.sr_image_amplitude(self._h, ptr(H)))
        return H

    @property
    def nbytes(self) -> int:
        return int(lib.sr_image_bytes(self._h))

    def close(self):
        if getattr(self, "_h", None):
            lib.sr_image_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- host-buffer entry points (A7-A11) --------------------------------------------------
def optics(r_mm, ops, E=None, kwave=0.0):
    """Apply an optic chain to r (4,N) in mm; returns (r_out, E_out)."""
    r = f64(r_mm)
    if r.ndim != 2 or r.shape[0] != 4:
        raise ValueError(f"rays must have shape (4, N), got {r.shape}")
    N = r.shape[1]
    ro = pinned_empty(r.shape)  # page-locked when large and asked for before (pinned_empty): the next step's upload is fast too
    Ei = None if E is None else np.ascontiguousarray(E, dtype=np.complex128)
    Eo = None if E is None else pinned_empty(Ei.shape, np.complex128)
    check(lib.sr_optics(make_chain(ops), len(ops), float(kwave), N, ptr(r), ptr(Ei), ptr(ro), ptr(Eo)))
    return ro, Eo


def hist2d(x, y, nxb, nyb, xlo, xhi, ylo, yhi):
    """np.histogram2d(x, y, bins=[nxb, nyb], range=...)[0].T as exact integer counts."""
    x, y = f64(x), f64(y)
    if x.shape != y.shape or x.ndim != 1:
        raise ValueError("x and y must be 1-D arrays of equal length")
    H = np.empty((int(nyb), int(nxb)), np.uint32)
    check(lib.sr_hist2d(ptr(x), ptr(y), len(x), int(nxb), int(nyb), float(xlo), float(xhi), float(ylo), float(yhi), ptr(H)))
    return H


def interferogram(x, y, E, nxe, nye, xlo, xhi, ylo, yhi, sums=False):
    """Interferometry.interferogram (rtm_solver.py:424-453): H = sqrt(Re(sum E_x)^2 + Re(sum E_y)^2),
    shape (nye-1, nxe-1); with sums=True also the per-pixel complex sums (2, nye-1, nxe-1)."""
    x, y = f64(x), f64(y)
    E = np.ascontiguousarray(E, dtype=np.complex128)
    if E.shape != (2, len(x)):
        raise ValueError(f"E must have shape (2, {len(x)}), got {E.shape}")
    amp = np.empty((2, int(nye) - 1, int(nxe) - 1), np.complex128) if sums else None
    H = np.empty((int(nye) - 1, int(nxe) - 1))
    check(lib.sr_interferogram(ptr(x), ptr(y), ptr(E), len(x), int(nxe), int(nye), float(xlo), float(xhi), float(ylo),
                               float(yhi), ptr(amp), ptr(H)))
    return (H, amp) if sums else H


def interfere_ref_beam(x, y, E, n_fringes, deg):
    x, y = f64(x), f64(y)
    Eo = np.array(E, dtype=np.complex128, order="C", copy=True)
    check(lib.sr_interfere_ref_beam(ptr(x), ptr(y), len(x), float(n_fringes), float(deg), ptr(Eo)))
    return Eo


# ---- the reference's fixed optic chains (A8) --------------------------------------------
def chain_shadow_single(L=400.0, R=25.0, focal_plane=0.0):
    """Shadowgraphy.single_lens_solve (rtm_solver.py:197-203; diagnostics.py:388-394)."""
    return [(OP_DIST, 3 * L / 4 - focal_plane), (OP_CIRC_AP, R), (OP_LENS, L / 2, L / 2), (OP_DIST, 3 * L / 2)]


def chain_shadow_two(L=400.0, R=25.0, focal_plane=0.0):
    """Shadowgraphy.two_lens_solve (rtm_solver.py:205-214; diagnostics.py:396-405) and the optics of
    Interferometry.two_lens_solve (rtm_solver.py:376-422)."""
    return [(OP_DIST, L - focal_plane), (OP_CIRC_AP, R), (OP_LENS, L / 2, L / 2), (OP_DIST, L * 2), (OP_CIRC_AP, R),
            (OP_LENS, L / 2, L / 2), (OP_DIST, L)]


def chain_shadow_exp(L=400.0, R=25.0, detL=400.0):
    """Shadowgraphy.single_exp_solve (rtm_solver.py:216-222)."""
    return [(OP_DIST, L), (OP_CIRC_AP, R), (OP_LENS, L / 2, L / 2), (OP_DIST, detL)]


def chain_schlieren(L=400.0, R=25.0, focal_plane=0.0, stop_R=1.0, dark_field=True):
    """Schlieren.DF_solve / LF_solve (rtm_solver.py:231-267; diagnostics.py:415-458)."""
    return [(OP_DIST, L - focal_plane), (OP_CIRC_AP, R), (OP_LENS, L, L), (OP_DIST, L),
            (OP_CIRC_STOP if dark_field else OP_CIRC_AP, stop_R), (OP_DIST, L), (OP_CIRC_AP, R), (OP_LENS, L, L),
            (OP_DIST, L)]


def chain_refractometry(L=400.0, R=25.0, focal_plane=0.0):
    """Refractometry.incoherent_solve (rtm_solver.py:276-286; diagnostics.py:467-481)."""
    return [(OP_DIST, 3 * L / 4 - focal_plane), (OP_CIRC_AP, R), (OP_LENS, L / 2, L / 2), (OP_DIST, 3 * L / 2),
            (OP_RECT_AP, 15, 30), (OP_CIRC_AP, R), (OP_LENS, L / 3, L / 2), (OP_DIST, L)]


def chain_refractometry_coherent(L=400.0, R=25.0, focal_plane=0.0, as_written_jax=False):
    """Refractometry.coherent_solve: use with E and kwave = 2*pi/wavelength.  Both generations are reproduced as
    written.  Legacy (rtm_solver.py:288-331): travel, aperture, lens, travel, aperture, hybrid lens, travel, where the
    field factor of the middle leg is computed between the aperture's output and input (:311-313) and is therefore 1.
    JAX file (diagnostics.py:505-524): the first aperture is applied to r0 and the chain carries on from there, so the
    first travel only contributes its field factor; the middle leg's factor is applied."""
    first, mid_flag = (OP_PHASE, 0) if as_written_jax else (OP_DIST, 1)
    return [(first, 3 * L / 4 - focal_plane), (OP_CIRC_AP, R), (OP_LENS, L / 2, L / 2), (OP_DIST, 3 * L / 2, 0.0, mid_flag),
            (OP_CIRC_AP, R), (OP_LENS, L / 3, L / 2), (OP_DIST, L)]


def slab_cuts(n_planes, n_slabs):
    """Node-plane ranges [(k_lo, k_hi), ...] of n_slabs slabs that share their boundary planes and cover 0..n_planes-1."""
    n_slabs = max(1, min(int(n_slabs), n_planes - 1))
    edges = [round(q * (n_planes - 1) / n_slabs) for q in range(n_slabs + 1)]
    return list(zip(edges[:-1], edges[1:]))


def slab_source(ne, axis, k_lo, k_hi):
    """The part of a whole-domain array a slab is built from: planes max(k_lo-1, 0)..min(k_hi+1, n-1) along `axis`."""
    n = ne.shape[axis]
    sl = [slice(None)] * 3
    sl[axis] = slice(max(k_lo - 1, 0), min(k_hi + 1, n - 1) + 1)
    return np.ascontiguousarray(ne[tuple(sl)])
