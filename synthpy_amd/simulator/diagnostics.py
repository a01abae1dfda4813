"""Host mirror of src/simulator/diagnostics.py (the JAX-generation diagnostics).

    sh = Shadowgraphy(lwl, rf); sh.single_lens_solve(); sh.histogram(bin_scale=1); sh.H
    sc = Schlieren(lwl, rf); sc.DF_solve(); sc.histogram()
    rr = Refractometry(lwl, rf); rr.incoherent_solve(); rr.histogram()
    it = Interferometry(lwl, rf, Jf); it.two_lens_solve(); it.interferogram(); it.H

Free functions are functional (return new arrays), rejected rays are NaN columns, metres in,
millimetres inside.  Optics, binning and the complex sums run on the GPU.
As in the reference, Interferometry.two_lens_solve first adds the reference beam
interfere_ref_beam(10, 20) (diagnostics.py:616) and propagates the field with k = 2*pi/wavelength.
Refractometry.coherent_solve is reproduced as written (the first aperture is applied to r0, so the first travel
only contributes its field factor, diagnostics.py:505-511).  Not carried over, and saying so when called:
Refractometry.fresnel_solve (needs the fresnel_integral module and attributes no caller sets) and Interferometry.bkg (as
shipped it stops at its first line: `probing_direction` is not defined there, diagnostics.py:583-584).
"""
from __future__ import annotations

import numpy as np

from .. import engine, resident
from ..engine import OP_CIRC_AP, OP_CIRC_STOP, OP_DIST, OP_KNIFE, OP_LENS, OP_RECT_AP, OP_SCALE


def _apply(r, ops, E=None, kwave=0.0):
    return engine.optics(np.asarray(r, dtype=np.float64), ops, E=E, kwave=kwave)


def m_to_mm(r):
    return _apply(r, [(OP_SCALE, 1e3)])[0]


def mm_to_m(r):
    return _apply(r, [(OP_SCALE, 1e-3)])[0]


def lens(r, f1, f2):
    return _apply(r, [(OP_LENS, f1, f2)])[0]


def sym_lens(r, f):
    return lens(r, f, f)


def travel(r, d):
    return _apply(r, [(OP_DIST, d)])[0]


def circular_aperture(r, R, E=None):
    """Reject rays outside radius R; with E also blanks the field of the rejected rays (diagnostics.py:173-189)."""
    if E is None:
        return _apply(r, [(OP_CIRC_AP, R)])[0]
    return _apply(r, [(OP_CIRC_AP, R)], E=E)


def circular_stop(r, R):
    return _apply(r, [(OP_CIRC_STOP, R)])[0]


def annular_stop(r, R1, R2):
    """The mask of the rays between R1 and R2, r untouched, as written (diagnostics.py:201-210)."""
    rr = np.asarray(r)[0, :] ** 2 + np.asarray(r)[2, :] ** 2
    return (rr > R1 ** 2) & (rr < R2 ** 2)


def rect_aperture(r, Lx, Ly):
    return _apply(r, [(OP_RECT_AP, Lx, Ly)])[0]


def knife_edge(r, offset, axis, direction):
    if axis not in ("x", "y"):
        raise ValueError("axis must be 'x' or 'y'")
    if direction == 0:
        raise ValueError("Direction must be < 0 or > 0")
    return _apply(r, [(OP_KNIFE, offset, direction, 0 if axis == "x" else 2)])[0]


def d2r(d):
    return d * np.pi / 180


def ray(x, θ, y, ϕ):
    """A 4 x 1 symbolic ray [x, theta, y, phi] (diagnostics.py:258-263; sympy, as there)."""
    import sympy as sym

    return sym.Matrix([x, θ, y, ϕ])


class Diagnostic:
    """Inheritable class for ray diagnostics (diagnostics.py:269-379).

    Built from the (rf, Jf) propagator.solve() has just returned, the object works on the bundle that call left in HBM
    (resident.attach): *_solve() and interfere_ref_beam() record what they would do, histogram() / interferogram() run the
    fused deposit (sr_rays_deposit), and .r0 / .rf / .Jf are copied to the host only when they are read.  Arrays from
    anywhere else, or changed since solve() wrote them, take the host path."""

    def __init__(self, wavelength, rf, Jf=None, *, focal_plane=0, L=400, R=25, Lx=18, Ly=13.5, x=None, y=None,
                 x_l=None, y_l=None, amp=None, phase=None):
        self.wavelength, self.focal_plane, self.L, self.R, self.Lx, self.Ly = wavelength, focal_plane, L, R, Lx, Ly
        self.x, self.y, self.x_l, self.y_l = x, y, x_l, y_l
        self.amp, self.phase = amp, phase
        if rf is None:
            raise ValueError("rf should not be None")
        self._r0 = self._rf = self._Jf = None
        self._has_Jf = Jf is not None
        self._assigned = False  # rf / Jf set by the caller or by a host-path step: work on THOSE, not on the resident rays
        bundle = resident.attach(self, rf, Jf)
        self._dev = None if bundle is None else resident.DeviceRays(bundle, Jf is not None)
        if self._dev is None:
            self._rf = np.array(rf, dtype=np.float64)
            self._Jf = None if Jf is None else np.array(Jf, dtype=np.complex128)
            self._r0 = m_to_mm(self._rf)

    # ---- the rays, wherever they are ----------------------------------------------------------------
    @property
    def on_device(self) -> bool:
        """True while histogram() / interferogram() deposit from the bundle solve() left in HBM."""
        return self._dev is not None and self._dev.live

    def _guard(self):
        """The arrays this object hands out while its rays are resident (.r0, .rf, .Jf) are write-tracked
        (resident.TrackedArray): once one of them has been written to, the object works on its host arrays, as they are
        now, from here on -- what the reference's methods would read (diagnostics.py:323-379)."""
        if self._dev is not None and any(resident.dirty(a) for a in (self._r0, self._rf, self._Jf)):
            self._leave_device(keep=True)

    @property
    def r0(self):
        if self._r0 is None and self.on_device:
            self._r0 = resident.track(self._dev.host(ops=[], with_E=False)[0])
        return self._r0

    @r0.setter
    def r0(self, value):
        self._leave_device(keep=False)
        self._r0 = value

    def _fetch(self):
        """rf and Jf as the reference would hold them now: the chain's output after a *_solve(), else the rays as given
        (metres) with whatever reference beams have been added to Jf."""
        if self._assigned or not self.on_device:
            return
        dev = self._dev
        field = dev.ops is not None and dev.kwave > 0  # the recorded chain carries Jf (two_lens_solve, coherent_solve)
        if self._rf is None:
            if dev.ops is None:
                self._rf = resident.track(dev.bundle.download(sf=False, rf=True, Jf=False)[1])
            else:
                rf, E = dev.host(with_E=field)
                self._rf = resident.track(rf)
                if field:
                    self._Jf = resident.track(E)
        if self._Jf is None and self._has_Jf:
            # a recorded chain that carries the field: its output (the caller may have set rf itself since); else as given,
            # plus the reference beams added so far
            self._Jf = resident.track(dev.host(with_E=True)[1] if field else dev.host(ops=[], with_E=True)[1])

    @property
    def rf(self):
        self._fetch()
        return self._rf

    @rf.setter
    def rf(self, value):
        if value is not None:
            self._fetch()  # BEFORE the new rf is stored: Jf stays what the recorded chain made of it
        self._rf = value
        self._assigned = value is not None

    @property
    def Jf(self):
        self._fetch()
        return self._Jf

    @Jf.setter
    def Jf(self, value):
        self._fetch()
        self._Jf, self._assigned = value, True
        self._has_Jf = value is not None

    def _to_host(self):
        self._leave_device(keep=True)

    def _leave_device(self, keep):
        if self._dev is None:
            return
        if keep and self._dev.live:
            _ = self.r0
            self._fetch()
        self._dev.drop(self)
        self._dev = None

    def __getstate__(self):
        self._to_host()
        return self.__dict__.copy()

    def _deposits_from_device(self):
        return self.on_device and self._dev.ops is not None and not self._assigned

    def _run(self, ops):
        self._guard()
        if self.on_device and not self._assigned:
            self._dev.record(ops)
            self._rf = None
        else:
            self.rf = _apply(self.r0, ops)[0]

    def _run_field(self, ops):
        """A chain that carries the field: Jf *= exp(1j*k*|dr|) over its legs, k = 2*pi/wavelength (diagnostics.py:311-318)."""
        k = 2 * np.pi / self.wavelength
        self._guard()
        if self.on_device and not self._assigned:
            self._dev.record(ops, kwave=k)
            self._rf = self._Jf = None
        else:
            r, E = _apply(self.r0, ops, E=self.Jf, kwave=k)
            self._rf, self._Jf, self._assigned = r, E, True

    def propagate_E(self, r1, r0):
        """Jf *= exp(1j * k * sqrt(dx^2 + dy^2)) for the leg r0 -> r1, k = 2*pi/wavelength (diagnostics.py:315-321).  The
        caller's own positions decide the factor, so this is host arithmetic on the host copy of the field."""
        r1, r0 = np.asarray(r1, dtype=np.float64), np.asarray(r0, dtype=np.float64)
        dx, dy = r1[0] - r0[0], r1[2] - r0[2]
        k = 2 * np.pi / self.wavelength
        self._to_host()
        self.Jf = self.Jf * np.exp(1.0j * k * np.sqrt(dx ** 2 + dy ** 2))

    def histogram(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        """histogram2d of the detector-plane positions, H [y_bin, x_bin] (diagnostics.py:323-353)."""
        nx, ny = pix_x // bin_scale, pix_y // bin_scale
        rng = (-self.Lx / 2, self.Lx / 2, -self.Ly / 2, self.Ly / 2)
        self._guard()
        if self._deposits_from_device():
            self.H = self._dev.counts(nx, ny, *rng)
        else:
            self.H = engine.hist2d(self.rf[0], self.rf[2], nx, ny, *rng).astype(np.float64)
        self.xedges = np.linspace(-self.Lx / 2, self.Lx / 2, nx + 1)
        self.yedges = np.linspace(-self.Ly / 2, self.Ly / 2, ny + 1)
        if clear_mem:
            clear_rays(self)

    def histogram_legacy(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        """Per-pixel complex sums of Jf, H = sqrt(Re^2 + Re^2); edges linspace(-L // 2, L // 2, pix // bin_scale)
        with the floor divisions as written (diagnostics.py:358-379)."""
        if not self._has_Jf:
            raise ValueError("This diagnostic requires a calculated Jf matrix.")
        rng = (-self.Lx // 2, self.Lx // 2, -self.Ly // 2, self.Ly // 2)
        self._guard()
        if self._deposits_from_device():
            self.H = self._dev.amplitude(pix_x // bin_scale, pix_y // bin_scale, *rng)
        else:
            self.H = engine.interferogram(self.rf[0], self.rf[2], self.Jf, pix_x // bin_scale, pix_y // bin_scale, *rng)
        if clear_mem:
            clear_rays(self)

    def plot(self, ax, clim=None, cmap=None):
        ax.imshow(self.H, interpolation="nearest", origin="lower", clim=clim, cmap=cmap,
                  extent=[self.xedges[0], self.xedges[-1], self.yedges[0], self.yedges[-1]])


def clear_rays(self):
    self._leave_device(keep=False)
    self._r0 = self._rf = self._Jf = None
    self._assigned = False


class Shadowgraphy(Diagnostic):
    def single_lens_solve(self):
        self._run(engine.chain_shadow_single(self.L, self.R, self.focal_plane))

    def two_lens_solve(self):
        self._run(engine.chain_shadow_two(self.L, self.R, self.focal_plane))


class Schlieren(Diagnostic):
    def DF_solve(self, R=1):
        self._run(engine.chain_schlieren(self.L, self.R, self.focal_plane, R, dark_field=True))

    def LF_solve(self, R=1):
        self._run(engine.chain_schlieren(self.L, self.R, self.focal_plane, R, dark_field=False))


class Refractometry(Diagnostic):
    def incoherent_solve(self):
        self._run(engine.chain_refractometry(self.L, self.R, self.focal_plane))

    def coherent_solve(self):
        if not self._has_Jf:
            raise ValueError("coherent_solve needs the field Jf (solve(..., return_E=True))")
        self._run_field(engine.chain_refractometry_coherent(self.L, self.R, self.focal_plane, as_written_jax=True))

    def refractogram(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        self.histogram_legacy(bin_scale=bin_scale, pix_x=pix_x, pix_y=pix_y, clear_mem=clear_mem)

    def fresnel_solve(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        raise NotImplementedError("fresnel_solve is outside this engine's path: the reference's version (diagnostics.py:529-552) "
                                  "replaces Jf by fresnel_integral.propagate(...) of x, y, x_l, y_l, amp, phase grids that no "
                                  "caller in the reference sets; trace the rays and use coherent_solve() + refractogram() instead")


class Interferometry(Diagnostic):
    def interfere_ref_beam(self, n_fringes, deg):
        """Add a tilted plane-wave reference to E_y: exp(2*n_fringes/3 * 1j*(x_w*x + y_w*y)) with
        y_w = arctan(deg*pi/180), x_w = sqrt(1 - y_w^2), deg >= 45 -> -|deg - 90| (diagnostics.py:559-581).
        x, y are self.rf as held (metres before a *_solve)."""
        if not self._has_Jf:
            print("This diagnostic requires a calculated Jf matrix.")
            return None
        self._guard()
        # resident rays that no *_solve() has moved yet: the beam is added by the deposit itself, before the chain
        if self.on_device and not self._assigned and self._dev.ops is None and self._dev.add_ref(n_fringes, deg):
            self._Jf = None
            return None
        self._to_host()
        self.Jf = engine.interfere_ref_beam(self.rf[0], self.rf[2], self.Jf, n_fringes, deg)

    def bkg(self, domain_length, n_fringes, deg, ne_extent):
        raise NotImplementedError("bkg cannot be reproduced: as shipped it stops at its first line (`probing_direction` is not "
                                  "defined in it and ray_to_Jonesvector has no keep_current_plane argument, diagnostics.py:583-584). "
                                  "A fringe background is the same chain on an empty domain: solve() rays through test_null(), "
                                  "then interfere_ref_beam(n_fringes, deg); two_lens_solve(); interferogram()")

    def two_lens_solve(self):
        self.interfere_ref_beam(10, 20)
        self._run_field(engine.chain_shadow_two(self.L, self.R, self.focal_plane))

    def interferogram(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        self.histogram_legacy(bin_scale=bin_scale, pix_x=pix_x, pix_y=pix_y, clear_mem=clear_mem)
