"""ctypes front-end of the parity oracle (oracle/synthray_oracle.c).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; the product package (synthpy_amd) never imports it.

Every function mirrors one row of SURVEY.md §8(a); the C file carries the
reference file:line citations.  Arrays follow the reference's layouts:
rays `(9, N)` float64, `rf (4, N)`, `Jf (2, N)` complex128, volumes C-order
`[ix, iy, iz]`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SYNTHRAY_ORACLE_LIB: another build of the same source (oracle/Makefile `asan`: the restatement under ASan + UBSan)
_LIB_PATH = os.path.abspath(os.environ["SYNTHRAY_ORACLE_LIB"]) if os.environ.get("SYNTHRAY_ORACLE_LIB") else os.path.join(_HERE, "liboracle.so")

c = 299792458.0  # scipy.constants.c

# optic op codes (same numbering as include/synthray.h SR_OP_*)
DIST, LENS, CIRC_AP, CIRC_STOP, RECT_AP, KNIFE, SCALE, PHASE = range(8)


class Optic(C.Structure):
    _fields_ = [("op", C.c_int32), ("iarg", C.c_int32), ("a", C.c_double), ("b", C.c_double)]


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "synthray_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_omega.restype = C.c_double
        _lib.orc_omega.argtypes = [C.c_double]
        _lib.orc_ncrit.restype = C.c_double
        _lib.orc_ncrit.argtypes = [C.c_double]
        _lib.orc_grad_scale.restype = C.c_float
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_num_threads(n: int) -> None:
    lib().orc_set_num_threads(C.c_int(int(n)))


# ---------------------------------------------------------------- A1 / A5
def omega(lwl: float) -> float:
    return float(lib().orc_omega(C.c_double(lwl)))


def calc_dndr(ne, x, y, z, lwl):
    """full_solver.ScalarDomain.calc_dndr → (omega, dndx, dndy, dndz) float32 volumes."""
    L = lib()
    x, y, z = _f32(x), _f32(y), _f32(z)
    nx, ny, nz = len(x), len(y), len(z)
    om = omega(lwl)
    nc = float(L.orc_ncrit(C.c_double(om)))
    ne = np.ascontiguousarray(ne)
    assert ne.shape == (nx, ny, nz), (ne.shape, (nx, ny, nz))
    ne_nc = np.empty((nx, ny, nz), np.float32)
    if ne.dtype == np.float32:
        L.orc_ne_nc_f32(_p(ne), C.c_int64(ne.size), C.c_double(nc), _p(ne_nc))
    else:
        ne = _f64(ne)
        L.orc_ne_nc_f64(_p(ne), C.c_int64(ne.size), C.c_double(nc), _p(ne_nc))
    scale = C.c_float(L.orc_grad_scale())
    out = []
    for axis, co in enumerate((x, y, z)):
        g = np.empty((nx, ny, nz), np.float32)
        L.orc_gradient_f32(_p(ne_nc), C.c_int(nx), C.c_int(ny), C.c_int(nz), C.c_int(axis), _p(co), scale, _p(g))
        out.append(g)
    return om, out[0], out[1], out[2]


def n_refrac(ne, om):
    ne = _f64(ne)
    out = np.empty_like(ne)
    lib().orc_n_refrac(_p(ne), C.c_int64(ne.size), C.c_double(om), _p(out))
    return out


# ---------------------------------------------------------------- A4
def interp(x, y, z, values, pts, fill):
    """scipy RegularGridInterpolator((x,y,z), values, bounds_error=False, fill_value=fill)(pts)."""
    gx, gy, gz = _f64(x), _f64(y), _f64(z)
    pts = _f64(pts).reshape(-1, 3)
    out = np.empty(len(pts))
    values = np.ascontiguousarray(values)
    fn = lib().orc_interp_f32 if values.dtype == np.float32 else lib().orc_interp_f64
    if values.dtype != np.float32:
        values = _f64(values)
    fn(C.c_int(len(gx)), C.c_int(len(gy)), C.c_int(len(gz)), _p(gx), _p(gy), _p(gz), _p(values), _p(pts),
       C.c_int64(len(pts)), C.c_double(fill), _p(out))
    return out


def kappa(ne, Te, Z, om):
    """ScalarDomain.kappa() (full_solver.py:243-268): NRL inverse-bremsstrahlung rate [1/s], float64 volume."""
    ne_cc = np.asarray(ne) * 1e-6
    o_max = 5.64e4 * np.sqrt(ne_cc)
    o_max = np.where(o_max < om, om, o_max)
    L_max = np.maximum(Z * 1.602176634e-19 / Te, 2.760428269727312e-10 / np.sqrt(Te))
    CL = np.maximum(2.0, np.log(4.19e5 * np.sqrt(Te) / (o_max * L_max)))
    return 3.1e-5 * Z * c * np.power(ne_cc / om, 2) * CL * np.power(Te, -1.5)


def verdet(lwl: float) -> float:
    """VerdetConst (full_solver.py:223)."""
    return 2.62e-13 * lwl ** 2


class Domain:
    """The fields the RHS reads: the result of calc_dndr (+ n_refrac when phaseshift, + kappa() when
    inv_brems, + ne and B with VerdetConst when B_on; full_solver.py:276-289)."""

    def __init__(self, x, y, z, dndx, dndy, dndz, om, nref=None, kappa=None, ne=None, B=None, verdet=0.0):
        self.x32, self.y32, self.z32 = _f32(x), _f32(y), _f32(z)
        self.gx, self.gy, self.gz = _f64(self.x32), _f64(self.y32), _f64(self.z32)
        self.dndx, self.dndy, self.dndz = _f32(dndx), _f32(dndy), _f32(dndz)
        self.nref = None if nref is None else _f64(nref)
        self.omega = float(om)
        self.kappa = None if kappa is None else _f64(kappa)
        self.ne = self.B = None
        self.verdet = float(verdet)
        if B is not None:
            self.ne = _f64(ne)
            self.B = [_f64(np.asarray(B)[..., k]) for k in range(3)]

    @classmethod
    def from_ne(cls, ne, x, y, z, lwl, phaseshift=False, Te=None, Z=None, B=None):
        om, gx, gy, gz = calc_dndr(ne, x, y, z, lwl)
        return cls(x, y, z, gx, gy, gz, om, n_refrac(ne, om) if phaseshift else None,
                   kappa(ne, Te, Z, om) if Te is not None else None, ne if B is not None else None, B, verdet(lwl))

    def _args(self):
        opt = lambda a: _p(a) if a is not None else None
        Bs = self.B or [None, None, None]
        return (C.c_int(len(self.gx)), C.c_int(len(self.gy)), C.c_int(len(self.gz)), _p(self.gx), _p(self.gy),
                _p(self.gz), _p(self.dndx), _p(self.dndy), _p(self.dndz), opt(self.nref), C.c_double(self.omega),
                opt(self.kappa), opt(self.ne), opt(Bs[0]), opt(Bs[1]), opt(Bs[2]), C.c_double(self.verdet))


# ---------------------------------------------------------------- A3
def dsdt(dom: Domain, s):
    s = _f64(s)
    N = s.shape[1]
    out = np.empty_like(s)
    lib().orc_dsdt(*dom._args(), _p(s), C.c_int64(N), _p(out))
    return out


# ---------------------------------------------------------------- A2
def default_t_end(extent: float) -> float:
    """t = sqrt(8)*extent/c  (full_solver.py:381)."""
    return float(np.sqrt(8.0) * extent / c)


def trace_rk4(dom: Domain, s0, dt, t_end, probing_direction="z", mode="planes", sub=1):
    """RK4 over dsdt on [0, t_end]; returns (sf (9,N) at t_end, ray_steps).

    mode "planes": steps from node plane to node plane of the probing axis (`sub` per cell), the
    engine's production form; mode "time": fixed time step dt with located entry/exit faces."""
    s0 = _f64(s0)
    N = s0.shape[1]
    sf = np.empty_like(s0)
    steps = C.c_int64(0)
    lib().orc_trace_rk4(*dom._args(), _p(s0), C.c_int64(N), C.c_int("xyz".index(probing_direction)), C.c_double(dt),
                        C.c_double(t_end), C.c_int(1 if mode == "planes" else 0), C.c_int(sub), _p(sf), C.byref(steps))
    return sf, int(steps.value)


def trace_slab(dom: Domain, t_end, probing_direction, k_lo, k_hi, *, s0=None, rec=None, last=False, sub=1):
    """A12: the plane form over the node planes [k_lo, k_hi] only.  First slab: s0 (9,N) in; later slabs: rec (10,N)
    in.  Returns (rec_out (10,N) | sf (9,N) when last, ray_steps)."""
    first = rec is None
    src = _f64(s0 if first else rec)
    N = src.shape[1]
    out = np.empty((9 if last else 10, N))
    steps = C.c_int64(0)
    lib().orc_trace_slab(*dom._args(), C.c_int("xyz".index(probing_direction)), C.c_int(sub), C.c_double(t_end),
                         C.c_int(k_lo), C.c_int(k_hi), C.c_int(first), C.c_int(last), _p(src) if first else None,
                         None if first else _p(src), C.c_int64(N), None if last else _p(out), _p(out) if last else None,
                         C.byref(steps))
    return out, int(steps.value)


# ---------------------------------------------------------------- A6
def ray_to_jones(sf, extent, probing_direction="z", order="legacy", return_E=True):
    sf = _f64(sf)
    N = sf.shape[1]
    rf = np.empty((4, N))
    Jf = np.empty((2, N), np.complex128) if return_E else None
    lib().orc_ray_to_jones(_p(sf), C.c_int64(N), C.c_double(extent), C.c_int("xyz".index(probing_direction)),
                           C.c_int(0 if order == "legacy" else 1), _p(rf), _p(Jf) if return_E else None)
    return rf, Jf


# ---------------------------------------------------------------- A7 / A8
def make_chain(ops):
    """ops: list of tuples (op, a[, b[, iarg]])."""
    arr = (Optic * len(ops))()
    for k, o in enumerate(ops):
        arr[k].op = int(o[0])
        arr[k].a = float(o[1]) if len(o) > 1 else 0.0
        arr[k].b = float(o[2]) if len(o) > 2 else 0.0
        arr[k].iarg = int(o[3]) if len(o) > 3 else 0
    return arr


def m_to_mm(r):
    rr = np.array(r, dtype=np.float64, copy=True)
    rr[0::2, :] *= 1e3
    return rr


def optics(r_mm, ops, E=None, kwave=0.0):
    """Apply an optic chain to r (4,N, mm); returns (r_out, E_out)."""
    r = np.array(r_mm, dtype=np.float64, copy=True, order="C")
    N = r.shape[1]
    Eo = None if E is None else np.array(E, dtype=np.complex128, copy=True, order="C")
    chain = make_chain(ops)
    lib().orc_optics(chain, C.c_int(len(ops)), C.c_double(kwave), C.c_int64(N), _p(r), _p(Eo) if Eo is not None else None)
    return r, Eo


# the reference's fixed chains (rtm_solver.py:197-286; diagnostics.py:388-481)
def chain_shadow_single(L=400.0, R=25.0, focal_plane=0.0):
    return [(DIST, 3 * L / 4 - focal_plane), (CIRC_AP, R), (LENS, L / 2, L / 2), (DIST, 3 * L / 2)]


def chain_shadow_two(L=400.0, R=25.0, focal_plane=0.0):
    return [(DIST, L - focal_plane), (CIRC_AP, R), (LENS, L / 2, L / 2), (DIST, L * 2), (CIRC_AP, R),
            (LENS, L / 2, L / 2), (DIST, L)]


def chain_schlieren(L=400.0, R=25.0, focal_plane=0.0, stop_R=1.0, dark_field=True):
    return [(DIST, L - focal_plane), (CIRC_AP, R), (LENS, L, L), (DIST, L),
            (CIRC_STOP if dark_field else CIRC_AP, stop_R), (DIST, L), (CIRC_AP, R), (LENS, L, L), (DIST, L)]


def chain_refractometry(L=400.0, R=25.0, focal_plane=0.0):
    return [(DIST, 3 * L / 4 - focal_plane), (CIRC_AP, R), (LENS, L / 2, L / 2), (DIST, 3 * L / 2), (RECT_AP, 15, 30),
            (CIRC_AP, R), (LENS, L / 3, L / 2), (DIST, L)]


def chain_refractometry_coherent(L=400.0, R=25.0, focal_plane=0.0, as_written_jax=False):
    """Refractometry.coherent_solve.  Legacy (rtm_solver.py:288-331): the field factor of the middle leg is taken
    between r5 and r4 (the aperture's output and input, :311-313), i.e. left out -> DIST with iarg = 1.  JAX file
    (diagnostics.py:505-524): the first aperture is applied to r0, so the first travel only contributes its field
    factor (PHASE); the middle leg's factor is there."""
    return [(PHASE if as_written_jax else DIST, 3 * L / 4 - focal_plane), (CIRC_AP, R), (LENS, L / 2, L / 2),
            (DIST, 3 * L / 2, 0.0, 0 if as_written_jax else 1), (CIRC_AP, R), (LENS, L / 3, L / 2), (DIST, L)]


def speckle(rf, E, sigma=0.8, Lx=18.0, Ly=13.5):
    """Refractometry.refractogram's per-ray random phase (rtm_solver.py:358-363): one np.random.randn() per ray that
    lands inside the digitize range, in ray order, from the global stream."""
    x, y = rf[0], rf[2]
    hit = (x >= -Lx // 2) & (x < Lx // 2) & (y >= -Ly // 2) & (y < Ly // 2)
    E = np.array(E, dtype=np.complex128)
    ph = np.array([sigma * np.random.randn() for _ in range(int(hit.sum()))])
    E[:, hit] = E[:, hit] * np.exp(1.0j * ph)
    return E


# ---------------------------------------------------------------- A9 / A10 / A11
def hist2d(x, y, nxb, nyb, xlo, xhi, ylo, yhi):
    x, y = _f64(x), _f64(y)
    H = np.zeros((nyb, nxb), np.uint32)
    lib().orc_hist2d(_p(x), _p(y), C.c_int64(len(x)), C.c_int(nxb), C.c_int(nyb), C.c_double(xlo), C.c_double(xhi),
                     C.c_double(ylo), C.c_double(yhi), _p(H))
    return H


def histogram(rf, bin_scale=10, pix_x=3448, pix_y=2574, Lx=18.0, Ly=13.5):
    """Rays.histogram (rtm_solver.py:156-178): H [y_bin, x_bin] of exact integer counts."""
    return hist2d(rf[0], rf[2], pix_x // bin_scale, pix_y // bin_scale, -Lx / 2, Lx / 2, -Ly / 2, Ly / 2)


def interferogram_sums(rf, E, bin_scale=1, pix_x=3448, pix_y=2574, Lx=18.0, Ly=13.5):
    """Complex per-pixel sums (2, ny, nx) before the final sqrt (rtm_solver.py:436-448)."""
    x, y = _f64(rf[0]), _f64(rf[2])
    E = np.ascontiguousarray(E, dtype=np.complex128)
    nxe, nye = pix_x // bin_scale, pix_y // bin_scale
    amp = np.zeros((2, nye - 1, nxe - 1), np.complex128)
    lib().orc_interferogram(_p(x), _p(y), _p(E), C.c_int64(len(x)), C.c_int(nxe), C.c_int(nye),
                            C.c_double(-Lx // 2), C.c_double(Lx // 2), C.c_double(-Ly // 2), C.c_double(Ly // 2), _p(amp))
    return amp


def interferogram(rf, E, **kw):
    a = interferogram_sums(rf, E, **kw)
    return np.sqrt(np.real(a[0]) ** 2 + np.real(a[1]) ** 2)


def interfere_ref_beam(rf, E, n_fringes, deg):
    E = np.array(E, dtype=np.complex128, copy=True, order="C")
    x, y = _f64(rf[0]), _f64(rf[2])
    lib().orc_interfere_ref_beam(_p(x), _p(y), C.c_int64(len(x)), C.c_double(n_fringes), C.c_double(deg), _p(E))
    return E
