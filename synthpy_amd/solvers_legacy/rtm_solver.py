"""Host mirror of the reference's ray-transfer-matrix diagnostics (src/solvers-legacy/rtm_solver.py):
same functions, classes, defaults and NaN-column convention; optics, binning and the complex
detector sums run on the GPU.  A diagnostic built from the arrays ScalarDomain.solve() has just returned works on the
bundle that call left in HBM: *_solve() records the chain, histogram() / interferogram() run the fused deposit
(sr_rays_deposit), .rf / .rE come to the host only when read (synthpy_amd/resident.py).  Any other arrays go through
the host-buffer entry points (sr_optics, sr_hist2d, sr_interferogram).

    sh = Shadowgraphy(rf, L=400, R=25); sh.two_lens_solve(); sh.histogram(bin_scale=10); sh.H
    sc = Schlieren(rf); sc.DF_solve(R=1); sc.histogram()
    rr = Refractometry(rf); rr.incoherent_solve(); rr.histogram()
    it = Interferometry(rf, E=Jf); it.two_lens_solve(wl=532e-9); it.interferogram(bin_scale=1)

Units: metres into the classes, millimetres inside (m_to_mm), as the reference.
The free functions return the modified rays; like the reference's, the aperture functions also
write the NaN columns into their argument.
Refractometry.refractogram's speckle phases (rtm_solver.py:361-363) are drawn on the host from the global
np.random stream in the reference's order, so a seeded call reproduces the reference's image.
"""
from __future__ import annotations

import numpy as np

from .. import engine, resident
from ..engine import OP_CIRC_AP, OP_CIRC_STOP, OP_DIST, OP_KNIFE, OP_LENS, OP_RECT_AP, OP_SCALE


def _apply(r, ops):
    return engine.optics(np.asarray(r, dtype=np.float64), ops)[0]


def m_to_mm(r):
    """Positions (rows 0, 2) metres -> millimetres (rtm_solver.py:48-51)."""
    return _apply(r, [(OP_SCALE, 1e3)])


def lens(r, f1, f2):
    """Thin lens, focal lengths f1, f2 in the two axes (rtm_solver.py:53-65)."""
    return _apply(r, [(OP_LENS, f1, f2)])


def sym_lens(r, f):
    return lens(r, f, f)


def distance(r, d):
    """Free propagation over d (rtm_solver.py:73-82)."""
    return _apply(r, [(OP_DIST, d)])


def _mask_inplace(r, ops):
    out = _apply(r, ops)
    r[...] = out
    return r


def circular_aperture(r, R):
    """Reject rays outside radius R (rtm_solver.py:84-90)."""
    return _mask_inplace(r, [(OP_CIRC_AP, R)])


def circular_stop(r, R):
    """Reject rays inside radius R (rtm_solver.py:92-98)."""
    return _mask_inplace(r, [(OP_CIRC_STOP, R)])


def annular_stop(r, R1, R2):
    """The MASK of the rays between radii R1 and R2 -- the reference returns the filter and leaves r alone
    (rtm_solver.py:100-108); host NumPy, nothing to accelerate."""
    rr = np.asarray(r)[0, :] ** 2 + np.asarray(r)[2, :] ** 2
    return (rr > R1 ** 2) & (rr < R2 ** 2)


def rect_aperture(r, Lx, Ly):
    """Reject rays with x^2 > Lx^2 AND y^2 > Ly^2 — the product of the two tests, as written (rtm_solver.py:110-118)."""
    return _mask_inplace(r, [(OP_RECT_AP, Lx, Ly)])


def knife_edge(r, offset, axis, direction):
    """Knife edge in 'x' or 'y'; direction > 0 rejects above the offset, < 0 below (rtm_solver.py:120-136)."""
    if axis not in ("x", "y"):
        raise ValueError("axis must be 'x' or 'y'")
    if direction == 0:
        raise ValueError("Direction must be <0 or >0")
    return _mask_inplace(r, [(OP_KNIFE, offset, direction, 0 if axis == "x" else 2)])


class Rays:
    """Inheritable class for ray diagnostics (rtm_solver.py:138-189).

    When r0 (and E) are the arrays ScalarDomain.solve() has just returned, the rays are still in HBM (resident.attach): the
    object then deposits from there -- *_solve() records its optic chain, histogram() / interferogram() run the fused
    deposit kernel (sr_rays_deposit) -- and .r0, .rf, .rE are copied to the host only when they are read.  Arrays from
    anywhere else, or changed since solve() wrote them, take the host path (sr_optics, sr_hist2d, sr_interferogram)."""

    def __init__(self, r0, E=None, focal_plane=0, L=400, R=25, Lx=18, Ly=13.5):
        self._dev = None
        self._E, self.focal_plane, self.L, self.R, self.Lx, self.Ly = E, focal_plane, L, R, Lx, Ly
        self._r0 = self._rf = self._rE = None
        self._assigned = False  # rf was set by the caller (or by a host-path solve): bin THAT, not the resident rays
        bundle = resident.attach(self, r0, E)
        self._dev = None if bundle is None else resident.DeviceRays(bundle, E is not None)
        if self._dev is None:
            self._r0 = m_to_mm(r0)

    # ---- the rays, wherever they are ----------------------------------------------------------------
    @property
    def on_device(self) -> bool:
        """True while histogram() / interferogram() deposit from the bundle solve() left in HBM."""
        return self._dev is not None and self._dev.live

    @property
    def E(self):
        return self._E

    @E.setter
    def E(self, value):  # another field than the resident one (the reference reads self.E when a *_solve() runs)
        self._leave_device(keep=True)
        self._E = value

    def _guard(self):
        """The reference reads self.E, self.r0 and self.rf when *_solve() / histogram() / interferogram() run, so a write to
        any of them since this object took the device path counts: the arrays are write-tracked (resident.TrackedArray), and
        once one has been written to the object works on its host arrays, as they are now, from here on."""
        if self._dev is not None and any(resident.dirty(a) for a in (self._E, self._r0, self._rf, self._rE)):
            self._leave_device(keep=True)

    @property
    def r0(self):
        if self._r0 is None and self.on_device:
            self._r0 = resident.track(self._dev.host(ops=[], with_E=False)[0])
        return self._r0

    @r0.setter
    def r0(self, value):
        self._leave_device(keep=False)  # other rays than the resident ones from here on
        self._r0 = value

    def _chain_output(self):
        if self._rf is None and not self._assigned and self.on_device and self._dev.ops is not None:
            rf, rE = self._dev.host(with_E=self._dev.has_E and self._dev.kwave > 0)
            self._rf, self._rE = resident.track(rf), resident.track(rE)

    @property
    def rf(self):
        self._chain_output()
        return self._rf

    @rf.setter
    def rf(self, value):
        self._rf, self._assigned = value, value is not None

    @property
    def rE(self):
        self._chain_output()
        return self._rE

    @rE.setter
    def rE(self, value):
        self._rE = value

    def _to_host(self):
        """Bring r0 (and the last *_solve()'s output) to the host and let go of the bundle (resident.release, pickling)."""
        self._leave_device(keep=True)

    def _leave_device(self, keep):
        if self._dev is None:
            return
        if keep and self._dev.live:
            _ = self.r0
            self._chain_output()
        self._dev.drop(self)
        self._dev = None

    def __getstate__(self):
        self._to_host()
        return self.__dict__.copy()

    def _run(self, ops):
        self._guard()
        if self.on_device:
            self._dev.record(ops)
            self._rf = self._rE = None
            self._assigned = False
        else:
            self.rf = _apply(self.r0, ops)

    def _deposits_from_device(self):
        return self.on_device and self._dev.ops is not None and not self._assigned

    def histogram(self, bin_scale=10, pix_x=3448, pix_y=2574, clear_mem=False):
        """np.histogram2d of the detector-plane positions; H [y_bin, x_bin] float64 holding exact counts,
        xedges / yedges as numpy returns them (rtm_solver.py:156-178)."""
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
            self.clear_rays()

    def plot(self, ax, clim=None, cmap=None):
        ax.imshow(self.H, interpolation="nearest", origin="lower", clim=clim, cmap=cmap,
                  extent=[self.xedges[0], self.xedges[-1], self.yedges[0], self.yedges[-1]])

    def clear_rays(self):
        self._leave_device(keep=False)
        self._r0 = self._rf = None
        self._assigned = False


class Shadowgraphy(Rays):
    """Two-lens telescope (M = 1) or single lens (M ~ 2); lenses f = L/2, radius R (rtm_solver.py:191-222)."""

    def single_lens_solve(self):
        self._run(engine.chain_shadow_single(self.L, self.R, self.focal_plane))

    def two_lens_solve(self):
        self._run(engine.chain_shadow_two(self.L, self.R, self.focal_plane))

    def single_exp_solve(self, detL=400):
        self._run(engine.chain_shadow_exp(self.L, self.R, detL))


class Schlieren(Rays):
    """Dark/light-field schlieren: telescope with f = L and a stop / pinhole of radius R at the focus
    (rtm_solver.py:224-267)."""

    def DF_solve(self, R=1):
        self._run(engine.chain_schlieren(self.L, self.R, self.focal_plane, R, dark_field=True))

    def LF_solve(self, R=1):
        self._run(engine.chain_schlieren(self.L, self.R, self.focal_plane, R, dark_field=False))


class Refractometry(Rays):
    """Imaging refractometer: spherical lens f = L/2 then a hybrid lens (L/3, L/2) (rtm_solver.py:269-286)."""

    def incoherent_solve(self):
        self._run(engine.chain_refractometry(self.L, self.R, self.focal_plane))

    def coherent_solve(self, wl=1064e-9):
        """The same imaging system carrying the field (rtm_solver.py:288-331): E *= exp(1j*k*|dr|) over every leg."""
        if self.E is None:
            raise ValueError("coherent_solve needs the field E (the Jf returned by solve(..., return_E=True))")
        ops, k = engine.chain_refractometry_coherent(self.L, self.R, self.focal_plane), 2 * np.pi / wl
        self._guard()
        if self.on_device:  # the speckle phases of refractogram() are drawn per ray on the host: it reads rf / rE from there
            self._dev.record(ops, kwave=k)
            self._rf = self._rE = None
            self._assigned = False
        else:
            self.rf, self.rE = engine.optics(self.r0, ops, E=self.E, kwave=k)

    def refractogram(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        """Complex sums per pixel with a random speckle phase 0.8*randn() per ray that lands on the detector
        (rtm_solver.py:333-369).  The phases are drawn on the host from the global np.random stream in ray order, one
        per in-range ray as the reference's loop does, so np.random.seed(k) reproduces the reference's image."""
        x, y = self.rf[0], self.rf[2]
        xlo, xhi, ylo, yhi = -self.Lx // 2, self.Lx // 2, -self.Ly // 2, self.Ly // 2
        hit = (x >= xlo) & (x < xhi) & (y >= ylo) & (y < yhi)
        E = np.array(self.rE, dtype=np.complex128)
        E[:, hit] *= np.exp(1.0j * (0.8 * np.random.randn(int(hit.sum()))))
        self.H = engine.interferogram(x, y, E, pix_x // bin_scale, pix_y // bin_scale, xlo, xhi, ylo, yhi)
        if clear_mem:
            self.clear_rays()


class Interferometry(Rays):
    """Two-lens telescope that also carries the field: E *= exp(1j*k*|dr|) over every free-space leg
    (rtm_solver.py:372-453)."""

    def two_lens_solve(self, wl=532e-9):
        if self.E is None:
            raise ValueError("Interferometry needs the field E (the Jf returned by solve(..., return_E=True))")
        ops, k = engine.chain_shadow_two(self.L, self.R, self.focal_plane), 2 * np.pi / wl
        self._guard()
        if self.on_device:
            self._dev.record(ops, kwave=k)
            self._rf = self._rE = None
            self._assigned = False
        else:
            self.rf, self.rE = engine.optics(self.r0, ops, E=self.E, kwave=k)

    def interferogram(self, bin_scale=1, pix_x=3448, pix_y=2574, clear_mem=False):
        """Per-pixel complex sums of E_x, E_y, H = sqrt(Re^2 + Re^2).  Edges are
        linspace(-L//2, L//2, pix//bin_scale): floor division as written, so y spans [-7, 6] for Ly = 13.5
        (rtm_solver.py:436-437)."""
        rng = (-self.Lx // 2, self.Lx // 2, -self.Ly // 2, self.Ly // 2)
        self._guard()
        if self._deposits_from_device() and self._dev.has_E:
            self.H = self._dev.amplitude(pix_x // bin_scale, pix_y // bin_scale, *rng)
        else:
            self.H = engine.interferogram(self.rf[0], self.rf[2], self.rE, pix_x // bin_scale, pix_y // bin_scale, *rng)
        if clear_mem:
            self.clear_rays()
