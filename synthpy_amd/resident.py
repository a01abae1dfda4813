"""The traced bundle a solve() left in HBM, found again by the diagnostics classes.

A synthPy caller writes (examples/jobs/run_scripts/pvti_trace_mpi.py:111-131, src/solvers-legacy/rtm_solver.py:142-178):

    rf = field.solve(ss); sh = rtm.Shadowgraphy(rf); sh.two_lens_solve(); sh.histogram()

The API hands `rf` over as a host array, but the rays themselves are still in HBM.  `register()` (called by the two
solve() mirrors) remembers which bundle an `rf` / `Jf` pair came from; `attach()` (called by Rays.__init__ /
Diagnostic.__init__) gives the bundle back when the arrays are THE arrays solve() returned and still hold what it wrote.
The diagnostic then records its optic chain in *_solve() and histogram() / interferogram() run the fused deposit
(sr_rays_deposit: m_to_mm -> reference beams -> chain -> LDS-tiled detector atomics) on the resident rays; `.r0`, `.rf`,
`.rE` / `.Jf` are formed on the device and copied to the host only when somebody reads them (sr_rays_optics).

The guard, in this order:

1. object identity and the bundle's generation counter (any upload / trace since then invalidates);
2. WRITE TRACKING (round 5): solve() hands rf / Jf out as `TrackedArray`s -- plain ndarrays to every reader -- whose
   writes go through Python: `rf[:, bad] = nan`, `rf[0, mask] = nan`, in-place operators (`rf[0:4:2, :] *= 1e3`,
   pvti_trace_mpi.py:120), ufuncs with `out=`, `np.copyto / put / place / putmask`, `.fill()`, `.sort()`, writes through any
   view (`rf[0][k] = ...`, `.T`, `.reshape`, `.real`): each marks the array's root dirty, and a dirty root is refused --
   the diagnostic then works on the host arrays as they are now, as the reference would (rtm_solver.py:142-178).  The arrays
   a device-backed diagnostic hands out itself (`.r0`, `.rf`, `.rE` / `.Jf`) are tracked the same way and looked at again by
   every *_solve() / histogram() / interferogram(): the reference reads `self.r0`, `self.E`, `self.rf` when those run, so a
   write between construction and the call counts;
3. for writers Python cannot see (`np.asarray(rf)` base-class views, `.data`, ctypes, C extensions): a sample of PROBE
   values per array compared bit for bit, every row sampled.  Single elements changed THAT way between the probes are
   not seen; SYNTHRAY_RESIDENT=full compares whole arrays against the device copy (one download: slower than the host
   path's upload), SYNTHRAY_RESIDENT=0 switches the whole mechanism off.
"""
from __future__ import annotations

import os
import weakref

import numpy as np

from . import engine

PROBE = 8192
MODE = os.environ.get("SYNTHRAY_RESIDENT", "1")  # "1" sampled guard, "full" whole-array guard, "0" off



class _Flag:
    """Shared by a TrackedArray and every view of it."""
    __slots__ = ("dirty",)

    def __init__(self):
        self.dirty = False


def _mark(a):
    f = getattr(a, "_sr_flag", None)
    if f is not None:
        f.dirty = True


def _plain(a):
    return a.view(np.ndarray) if isinstance(a, TrackedArray) else a


class TrackedArray(np.ndarray):
    """An ndarray that remembers whether anybody has written to it (or to a view of it) since track() made it.

    Readers see an ordinary array: arithmetic and ufuncs return plain ndarrays, copies and fancy-indexed selections
    start untracked.  Every write numpy routes through Python marks the shared flag (module docstring, point 2)."""

    _sr_flag = None

    def __array_finalize__(self, obj):
        # a VIEW of a tracked array shares its flag; a new buffer (copy, astype, fancy index, a ufunc's result) does not
        f = getattr(obj, "_sr_flag", None)
        if f is not None and self.base is not None and (self.base is obj or self.base is obj.base):
            self._sr_flag = f

    # ---- writes -------------------------------------------------------------------------------------
    def __setitem__(self, key, value):
        _mark(self)
        super().__setitem__(key, value)

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        if out is not None:
            for o in out:
                _mark(o)
            kwargs["out"] = tuple(_plain(o) for o in out)
        if method == "at" and inputs:
            _mark(inputs[0])
        res = getattr(ufunc, method)(*(_plain(x) for x in inputs), **kwargs)
        if out is not None and method != "at":  # `a += b` must leave `a` what it was
            return out[0] if len(out) == 1 else tuple(out)
        return res

    def __array_function__(self, func, types, args, kwargs):
        name = getattr(func, "__name__", "")
        if name in _WRITERS:
            if args:
                _mark(args[0])
            for k in ("dst", "a", "arr"):
                _mark(kwargs.get(k))
        elif name == "nan_to_num" and kwargs.get("copy") is False and args:
            _mark(args[0])
        o = kwargs.get("out")
        for t in (o if isinstance(o, tuple) else (o,)):
            _mark(t)
        return super().__array_function__(func, types, args, kwargs)

    @property
    def flat(self):  # a flatiter writes straight to memory: handing one out counts as a write
        _mark(self)
        return np.ndarray.flat.__get__(self)

    @flat.setter
    def flat(self, value):
        _mark(self)
        np.ndarray.flat.__set__(self, value)

    @property
    def real(self):
        return np.ndarray.real.__get__(self)

    @real.setter
    def real(self, value):
        _mark(self)
        np.ndarray.real.__set__(self, value)

    @property
    def imag(self):
        return np.ndarray.imag.__get__(self)

    @imag.setter
    def imag(self, value):
        _mark(self)
        np.ndarray.imag.__set__(self, value)

    def __reduce__(self):  # pickles as the plain array it is to every reader
        return np.asarray(self).__reduce__()


_WRITERS = frozenset(("copyto", "put", "place", "putmask", "put_along_axis", "fill_diagonal"))


def _marking(name):
    """ndarray.<name>, a method that writes in place, with the mark in front."""
    base = getattr(np.ndarray, name)

    def method(self, *a, **k):
        _mark(self)
        return base(self, *a, **k)

    method.__name__ = name
    method.__doc__ = base.__doc__
    return method


for _n in ("fill", "sort", "partition", "put", "itemset", "setfield", "resize", "byteswap", "setflags"):
    if hasattr(np.ndarray, _n):  # (itemset went with NumPy 2)
        setattr(TrackedArray, _n, _marking(_n))
del _n


def track(a):
    """`a` as a TrackedArray with a clean flag of its own (a view: no copy); None stays None."""
    if a is None or MODE == "0":
        return a
    t = np.asarray(a).view(TrackedArray)
    t._sr_flag = _Flag()
    return t


def dirty(a) -> bool:
    """True when a tracked array (or a view of it) has been written to since track()."""
    f = getattr(a, "_sr_flag", None)
    return f is not None and f.dirty


_entries = {}   # id(rf) -> _Entry
_orphans = []   # weakrefs of bundles that only diagnostics still hold (their domain has moved on to another bundle)


_index_cache = {}


def _probe_index(a):
    n = a.size
    idx = _index_cache.get(n)
    if idx is None:
        idx = np.linspace(0, n - 1, min(PROBE, n)).astype(np.int64) if n else np.zeros(0, np.int64)
        if len(_index_cache) > 64:
            _index_cache.clear()
        _index_cache[n] = idx
    return idx


def _probe(a):
    flat = a.reshape(-1)
    return flat[_probe_index(a)].copy()


def _same(a, probe):
    got = _probe(a)
    return got.shape == probe.shape and got.tobytes() == probe.tobytes()  # bit for bit: NaN columns compare equal to themselves


class _Entry:
    # the bundle by weak reference: its domain and the diagnostics that deposit from it keep it alive, an old rf array
    # somebody still holds does not pin its HBM
    __slots__ = ("bundle_ref", "generation", "rf_ref", "rf_probe", "Jf_ref", "Jf_probe")


def register(bundle, rf, Jf=None):
    """solve() returns `rf` (and `Jf`) from `bundle`: remember it until the arrays are collected.  Gives back the arrays to
    hand to the caller: the same memory as TrackedArrays (module docstring)."""
    if MODE == "0" or rf is None:
        return rf, Jf
    rf, Jf = track(rf), track(Jf)
    e = _Entry()
    e.bundle_ref, e.generation = weakref.ref(bundle), bundle.generation
    key = id(rf)
    e.rf_ref = weakref.ref(rf, lambda _r, key=key: _entries.pop(key, None))
    e.rf_probe = _probe(rf)
    e.Jf_ref = None if Jf is None else weakref.ref(Jf)
    e.Jf_probe = None if Jf is None else _probe(Jf)
    _entries[key] = e
    return rf, Jf


def attach(owner, rf, E=None):
    """The bundle `rf` (and `E`, when the diagnostic carries the field) came from, or None when the caller's arrays are not
    what a solve() returned any more (or never were): then the host path runs, on the arrays as they are."""
    if MODE == "0" or not isinstance(rf, np.ndarray):
        return None
    e = _entries.get(id(rf))
    if e is None or e.rf_ref() is not rf:
        return None
    b = e.bundle_ref()
    if b is None or not b.alive or b.generation != e.generation:
        _entries.pop(id(rf), None)
        return None
    if E is not None and (e.Jf_ref is None or e.Jf_ref() is not E):
        return None
    if dirty(rf) or dirty(E):  # written to since solve() returned them: the host path, on the arrays as they are now
        return None
    if MODE == "full":
        _, drf, dJf = b.download(sf=False, Jf=E is not None)
        if drf.tobytes() != rf.tobytes() or (E is not None and dJf.tobytes() != np.ascontiguousarray(E).tobytes()):
            return None
    elif not _same(rf, e.rf_probe) or (E is not None and not _same(E, e.Jf_probe)):
        return None
    b.holders.add(owner)
    return b


def bundle_bytes(n):
    """HBM of one RayBundle of n rays with its hand-off records (sr_rays: s0, sf 9 rows; rf, Jf 4; rec, rec2 and the tile path's
    straggler records 10 each; 32-bit index arrays)."""
    return int(n) * (8 * (9 + 9 + 4 + 4 + 10 + 10 + 10) + 4 * 8)


def acquire(n, current=None):
    """The bundle the next solve() of a domain traces in.  `current` is reused when no diagnostic deposits from it any more;
    otherwise it is left to its diagnostics (they keep it alive) and a new one is made -- after the oldest such left-behind
    bundles have been written out to their diagnostics' host arrays if HBM is short."""
    n = int(n)
    if current is not None and current.alive and current.n == n and not len(current.holders):
        return current
    if current is not None and current.alive and len(current.holders):
        _orphans.append(weakref.ref(current))
    need = bundle_bytes(n) + (2 << 30)
    while _orphans:
        free, _ = engine.device_memory()
        if free >= need:
            break
        b = _orphans.pop(0)()
        if b is not None and b.alive:
            release(b)
    return engine.RayBundle(n)


def release(bundle):
    """Every diagnostic that still deposits from `bundle` takes its rays to the host (what the reference's constructor did
    in the first place: r0 = m_to_mm(rf), a host copy); then the bundle's HBM is given back."""
    for owner in list(bundle.holders):
        owner._to_host()
    bundle.holders.clear()
    bundle.close()


class DeviceRays:
    """What a diagnostic object keeps instead of r0 / rf host arrays while its rays are resident: the bundle, the chain its
    *_solve() recorded, the field's wavenumber and the reference beams added before the chain."""

    def __init__(self, bundle, has_E):
        self.bundle, self.has_E = bundle, bool(has_E)
        self.ops, self.kwave, self.refs = None, 0.0, []

    @property
    def live(self):
        return self.bundle is not None and self.bundle.alive

    def add_ref(self, n_fringes, deg):
        """False when the deposit cannot hold another reference beam (the caller then goes to the host path)."""
        if len(self.refs) >= engine._ffi.MAX_REF_BEAMS:
            return False
        self.refs.append((float(n_fringes), float(deg)))
        return True

    def record(self, ops, kwave=0.0):
        self.ops, self.kwave = list(ops), float(kwave)

    def host(self, ops=None, with_E=None):
        """(r (4, N) mm, E | None) on the host in the original ray order: r0 for ops == (), else the chain's output."""
        ops = self.ops if ops is None else ops
        with_E = self.has_E if with_E is None else with_E
        return self.bundle.optics(ops, kwave=self.kwave if ops else 0.0, ref_beam=self.refs or None, with_E=with_E)

    def _image(self, kind, nx, ny, rng):
        """The detector image of this geometry, kept with the bundle (a chunk loop asks for the same detector every time:
        no hipMalloc / hipFree per histogram) and zeroed for this deposit."""
        cache = self.bundle.__dict__.setdefault("_images", {})
        key = (kind, int(nx), int(ny), tuple(float(v) for v in rng))
        img = cache.get(key)
        if img is None:
            if len(cache) >= 4:  # a caller sweeping bin_scale: do not pile detectors up
                for old in cache.values():
                    old.close()
                cache.clear()
            img = cache[key] = engine.DetectorImage(kind, nx, ny, *rng)
        else:
            img.zero()
        return img

    def counts(self, nx, ny, x_lo, x_hi, y_lo, y_hi):
        """Rays.histogram on the resident rays: float64 [ny][nx] holding exact integer counts."""
        img = self._image(engine.IMG_COUNTS, nx, ny, (x_lo, x_hi, y_lo, y_hi))
        # exact_counts off: the rays in HBM are bit for bit the rf solve() returned, so this IS np.histogram2d of them
        self.bundle.deposit(img, self.ops, want_stats=False, exact_counts=False)
        return img.counts_f64()

    def amplitude(self, nxe, nye, x_lo, x_hi, y_lo, y_hi):
        """Interferometry.interferogram on the resident rays: H = sqrt(Re(sum E_x)^2 + Re(sum E_y)^2), (nye-1, nxe-1)."""
        img = self._image(engine.IMG_COMPLEX, nxe, nye, (x_lo, x_hi, y_lo, y_hi))
        self.bundle.deposit(img, self.ops, kwave=self.kwave, ref_beam=self.refs or None, want_stats=False)
        return img.amplitude()

    def drop(self, owner):
        if self.bundle is not None:
            self.bundle.holders.discard(owner)
        self.bundle = None
