"""Host mirror of the reference's legacy solver module (src/solvers-legacy/full_solver.py):
same names, argument meaning and shapes; the work is done by libsynthray.so on the GPU.

    domain = ScalarDomain(x, y, z, extent, phaseshift=...)      full_solver.py:102
    domain.external_ne(ne) | domain.test_slab() ...             :130-175   (inputs: host NumPy)
    domain.calc_dndr(lwl)                                       :211-234   -> sr_volume_create
    s0 = init_beam(Np, beam_size, divergence, ne_extent, beam_type, probing_direction)   :547-835 (input)
    rf = domain.solve(s0)   |   rf, Jf = domain.solve(s0, return_E=True)   :376-403 -> sr_trace

Differences from the reference, decided rather than copied (SURVEY.md §7 "quirks"):
  * the integrator is a fixed-step RK4 from node plane to node plane instead of solve_ivp's RK45 at
    rtol 1e-3 with one step size shared by all rays; results agree with the reference's RHS integrated
    at rtol 1e-10 to 1e-8 m / 1e-6 rad (tests/), i.e. far inside the reference's own default error;
  * phaseshift=True works (the reference raises NameError: omega_pe, full_solver.py:236,273);
  * inv_brems=True works (the reference's kappa() raises the same NameError); as in the reference the rate is
    applied with a plus sign, d(amp) = +kappa*amp (full_solver.py:268,540): amplitudes grow;
  * set_up_interps() (the upload of the kappa / ne / B volumes) is called by solve() if the caller has not;
  * an unknown probing_direction or beam_type raises ValueError instead of printing and continuing;
  * beam_type 'even' and 'rect_trackers' are broken in the reference (float range(), undefined
    N_trackers: full_solver.py:781-795,825) and are rejected.
"""
from __future__ import annotations

from time import time

import numpy as np

from .. import _beam, engine, resident

c = engine.c


class ScalarDomain:
    """Holds the scalar fields of the plasma volume and traces rays through them (full_solver.py:96-128)."""

    def __init__(self, x, y, z, extent, B_on=False, inv_brems=False, phaseshift=False, probing_direction="z"):
        self.x, self.y, self.z = np.float32(x), np.float32(y), np.float32(z)  # full_solver.py:119
        # the reference keeps full meshgrids (:120); sparse ones broadcast to the same values
        self.XX, self.YY, self.ZZ = np.meshgrid(x, y, z, indexing="ij", sparse=True, copy=False)
        self.extent = extent
        self.probing_direction = probing_direction
        self.B_on = B_on
        self.inv_brems = inv_brems
        self.phaseshift = phaseshift
        self._volume = None
        self._fields = None
        self._rays = self._sf = self._Jf = None
        self.precision = engine.DEFAULT_PRECISION  # "auto" | "mixed" | "f64", see engine.resolve_precision
        self.substeps = 1                          # RK4 steps per cell

    # ---- analytic test profiles (inputs; full_solver.py:130-167) -------------------------
    def _full(self, a):
        return np.ascontiguousarray(np.broadcast_to(a, (len(self.x), len(self.y), len(self.z))))

    def test_null(self):
        """Null test, an empty cube."""
        self.ne = np.zeros((len(self.x), len(self.y), len(self.z)))

    def test_slab(self, s=1, n_e0=2e23):
        """n_e = n_e0*(1 + s*x/extent): deflects rays in x."""
        self.ne = self._full(n_e0 * (1.0 + s * self.XX / self.extent))

    def test_linear_cos(self, s1=0.1, s2=0.1, n_e0=2e23, Ly=1):
        """Linearly growing sinusoidal perturbation."""
        self.ne = self._full(n_e0 * (1.0 + s1 * self.XX / self.extent) * (1 + s2 * np.cos(2 * np.pi * self.YY / Ly)))

    def test_exponential_cos(self, n_e0=1e24, Ly=1e-3, s=2e-3):
        """Exponentially growing sinusoidal perturbation."""
        self.ne = self._full(n_e0 * 10 ** (self.XX / s) * (1 + np.cos(2 * np.pi * self.YY / Ly)))

    def external_ne(self, ne):
        """Load an externally generated (nx, ny, nz) grid of n_e in m^-3."""
        self.ne = ne
        self._volume = None

    def external_B(self, B):
        """(nx, ny, nz, 3) grid of B in T."""
        self.B = B
        self._aux_ready = False

    def external_Te(self, Te, Te_min=1.0):
        """(nx, ny, nz) grid of T_e in eV, floored at Te_min."""
        self.Te = np.maximum(Te_min, Te)
        self._aux_ready = False

    def external_Z(self, Z):
        """(nx, ny, nz) grid of the ionisation."""
        self.Z = Z
        self._aux_ready = False

    def test_B(self, Bmax=1.0):
        self.B = np.zeros((len(self.x), len(self.y), len(self.z), 3))
        self.B[:, :, :, 2] = self._full(Bmax * self.XX / self.extent)
        self._aux_ready = False

    def kappa(self):
        """Inverse-bremsstrahlung rate coefficient [1/s] on the grid (NRL formulary; full_solver.py:243-268):
        3.1e-5 * Z * c * (n_e[cm^-3]/omega)^2 * ln(Lambda) * Te^-1.5, ln(Lambda) = max(2, ln(v_the / (omega_max * L_max)))."""
        ne_cc = np.asarray(self.ne) * 1e-6
        omega_max = np.maximum(5.64e4 * np.sqrt(ne_cc), self.omega)          # max(omega_pe, omega)
        L_max = np.maximum(self.Z * 1.602176634e-19 / self.Te,                # classical distance of closest approach
                           2.760428269727312e-10 / np.sqrt(self.Te))          # de Broglie length
        coulomb_log = np.maximum(2.0, np.log(4.19e5 * np.sqrt(self.Te) / (omega_max * L_max)))
        return 3.1e-5 * self.Z * c * np.power(ne_cc / self.omega, 2) * coulomb_log * np.power(self.Te, -1.5)

    def n_refrac(self):
        """Plasma refractive index on the grid (full_solver.py:271-274), read back from the GPU volume."""
        return self._volume.fields(phase=True)[3] + 1.0

    def set_up_interps(self):
        """Upload the volumes of the optional RHS terms (full_solver.py:276-289): kappa() when inv_brems, n_e and B
        when B_on.  (The n_e gradient and refractive-index volumes are built by calc_dndr.)"""
        if self._volume is None:
            raise RuntimeError("call calc_dndr(lwl) first")
        if self.inv_brems or self.B_on:
            self._volume.attach_aux(self.kappa() if self.inv_brems else None,
                                    self._full(np.asarray(self.ne, np.float64)) if self.B_on else None,
                                    np.ascontiguousarray(self.B, np.float64) if self.B_on else None,
                                    self.VerdetConst if self.B_on else 0.0)
        self._aux_ready = True

    # ---- A1 / A5 ---------------------------------------------------------------------
    def calc_dndr(self, lwl=1053e-9):
        """Generate the gradient fields (and the refractive index when phaseshift) on the GPU.

        full_solver.py:211-234: omega, n_c, ne_nc = float32(ne/n_c), dnd{x,y,z} = -c^2/2 * np.gradient(...)."""
        self.omega = 2 * np.pi * (c / lwl)
        if self.B_on:
            self.VerdetConst = 2.62e-13 * lwl ** 2  # radians per Tesla per m^2 (full_solver.py:223)
        self._aux_ready = False
        self._volume = engine.Volume.from_ne(self.ne, self.x, self.y, self.z, lwl,
                                              probing_direction=self.probing_direction, phaseshift=self.phaseshift)
        self._fields = None

    def _field(self, k):
        if self._volume is None:
            raise RuntimeError("call calc_dndr(lwl) first")
        if self._fields is None:
            self._fields = self._volume.fields(phase=False)
        return self._fields[k]

    @property
    def dndx(self):
        return self._field(0)

    @property
    def dndy(self):
        return self._field(1)

    @property
    def dndz(self):
        return self._field(2)

    @staticmethod
    def omega_pe(ne):
        """Electron plasma frequency [rad/s] of n_e [cm^-3] (NRL formulary; full_solver.py:236-239).  The reference defines
        it in the class body without `self` and calls it as a bare name, which is why its own phaseshift / inv_brems runs stop
        with NameError; here it is callable both ways."""
        return 5.64e4 * np.sqrt(ne)

    def plot_midline_gradients(self, ax, probing_direction):
        """The three gradient components along the line through the middle of the box (full_solver.py:291-315: along x for
        'x', along z for 'z', along y otherwise, always drawn against self.y as there)."""
        m = len(self.x) // 2
        line = {"x": (slice(None), m, m), "z": (m, m, slice(None))}.get(probing_direction, (m, slice(None), m))
        for g in (self.dndx, self.dndy, self.dndz):
            ax.plot(self.y, g[line])

    def dndr(self, x):
        """Gradient at the (3, N) locations x -> (3, N) (full_solver.py:317-332), interpolated on the GPU."""
        if self._volume is None:
            raise RuntimeError("call calc_dndr(lwl) first")
        return self._volume.sample(np.asarray(x).T)[:3]

    def phase(self, x):
        """omega*(n(x) - 1) at the (3, N) locations x (full_solver.py:342-347)."""
        if not self.phaseshift:
            return 0.0
        return self.omega * self._volume.sample(np.asarray(x).T)[3]

    def atten(self, x):
        """kappa at the (3, N) locations x (full_solver.py:334-339)."""
        if not self.inv_brems:
            return 0.0
        self._check_terms()
        return self._volume.sample_aux(np.asarray(x).T)[0]

    def get_ne(self, x):
        self._check_terms()
        return self._volume.sample_aux(np.asarray(x).T)[1]

    def get_B(self, x):
        self._check_terms()
        return self._volume.sample_aux(np.asarray(x).T)[2:5]

    def neB(self, x, v):
        """VerdetConst * ne * B.v at locations x with velocities v, both (3, N) (full_solver.py:356-374)."""
        if not self.B_on:
            return 0.0
        self._check_terms()
        X = self._volume.sample_aux(np.asarray(x).T)
        return self.VerdetConst * X[1] * np.sum(X[2:5] * np.asarray(v), axis=0)

    # ---- A2 + A6 ---------------------------------------------------------------------
    def _check_terms(self):
        if (self.inv_brems or self.B_on) and not getattr(self, "_aux_ready", False):
            self.set_up_interps()

    def _solve(self, s0, t_end, return_E):
        if self._volume is None:
            raise RuntimeError("call calc_dndr(lwl) first")
        self._check_terms()
        s0 = np.asarray(s0, dtype=np.float64)
        if s0.ndim == 1:
            s0 = s0.reshape(9, -1)
        start = time()
        # the traced bundle stays in HBM: rf comes back now, Jf with return_E, and sf / Jf otherwise only when the
        # attribute is read (0.72 GB of final states per 1e7 rays that most callers never look at).  The diagnostics
        # classes find the bundle again through the arrays returned here (resident.attach) and deposit from it.
        rays = resident.acquire(s0.shape[1], self._rays)
        self._rays = rays.upload(s0)
        self.trace_stats = rays.trace(self._volume, t_end, self.extent, row_order=engine.ROWS_LEGACY, precision=self.precision,
                                      substeps=self.substeps, resident=False)  # rf goes back to the caller, who may bin it: "auto" = float64
        self._sf = self._Jf = None
        _, rf, Jf = rays.download(sf=False, Jf=return_E)
        self.rf, Jf = resident.register(rays, rf, Jf)  # the same memory, write-tracked (resident.TrackedArray)
        if return_E:
            self._Jf = Jf
        self.duration = time() - start
        return (self.rf, self._Jf) if return_E else self.rf

    @property
    def sf(self):
        """Final states (9, N) at t_end of the last solve (full_solver.py:397), fetched from the GPU on first use."""
        if self._sf is None and self._rays is not None:
            self._sf = self._rays.download(rf=False, Jf=False)[0]
        return self._sf

    @sf.setter
    def sf(self, value):
        self._sf = value

    @property
    def Jf(self):
        if self._Jf is None and self._rays is not None:
            self._Jf = self._rays.download(sf=False, rf=False)[2]
        return self._Jf

    @Jf.setter
    def Jf(self, value):
        self._Jf = value

    def solve(self, s0, return_E=False):
        """Trace s0 (9, N) for t = sqrt(8)*extent/c and back-project to the exit plane (full_solver.py:376-403)."""
        return self._solve(s0, np.sqrt(8.0) * self.extent / c, return_E)

    def solve_at_depth(self, s0, z):
        """Trace for the time length z/c (full_solver.py:405-425)."""
        return self._solve(s0, z / c, False)

    def export_scalar_field(self, property: str = "ne", fname: str = None):
        """Save n_e as <fname>.vti + <fname>.pvti (full_solver.py:442-512), written without pyvista."""
        from ..utils.handle_filetypes import export_scalar_field

        export_scalar_field(self, property, fname)

    def clear_memory(self):
        """Drop the volume and ray attributes (full_solver.py:427-440)."""
        if self._volume is not None:
            self._volume.close()
        self._volume = None
        self._fields = None
        self.ne = None
        if self._rays is not None:
            resident.release(self._rays)  # diagnostics still depositing from it take their rays to the host first
        self._rays = None
        self.sf = None
        self.rf = None


def dsdt(t, s, ScalarDomain):
    """RHS of the photon-path ODE for the flattened (9N,) state (full_solver.py:516-544): the gathers run on the GPU."""
    s = np.asarray(s, dtype=np.float64)
    Np = s.size // 9
    s = s.reshape(9, Np)
    sprime = np.zeros_like(s)
    F = ScalarDomain._volume.sample(s[:3].T)
    sprime[3:6] = F[:3]
    sprime[:3] = s[3:6]
    sprime[6] = ScalarDomain.atten(s[:3]) * s[6]
    if ScalarDomain.phaseshift:
        sprime[7] = ScalarDomain.omega * F[3]
    sprime[8] = ScalarDomain.neB(s[:3], s[3:6])
    return sprime.flatten()


def init_beam(Np, beam_size, divergence, ne_extent, beam_type, probing_direction="z"):
    """Draw the initial ray bundle s0 (9, Np) (full_solver.py:547-835).  Uses the global np.random stream in
    the reference's order, so np.random.seed(k) before the call reproduces the reference's rays."""
    Np = int(Np)
    rand, randn = np.random.rand, np.random.randn
    if beam_type == "circular":
        t = 2 * np.pi * rand(Np)          # polar angle of position
        u = rand(Np) + rand(Np)           # radial coordinate, folded at 1 -> uniform over the disc
        u[u > 1] = 2 - u[u > 1]
        phi = np.pi * rand(Np)            # azimuth of velocity
        chi = divergence * randn(Np)      # polar angle of velocity
        return _beam.assemble(beam_size * u * np.cos(t), beam_size * u * np.sin(t), chi, phi, ne_extent, probing_direction)
    if beam_type in ("square", "rectangular"):
        t = 2 * rand(Np) - 1.0
        u = 2 * rand(Np) - 1.0
        phi = np.pi * rand(Np)
        chi = divergence * randn(Np)
        b1, b2 = (beam_size, beam_size) if beam_type == "square" else (beam_size[0], beam_size[1])
        return _beam.assemble(b1 * u, b2 * t, chi, phi, ne_extent, probing_direction)
    if beam_type == "linear":
        t = 2 * rand(Np) - 1.0
        chi = divergence * randn(Np)
        return _beam.assemble_linear(t, chi, beam_size, ne_extent)
    raise ValueError(f"beam_type {beam_type!r} unrecognised or broken in the reference; accepted: circular, square, "
                     "rectangular, linear")


def ray_to_Jonesvector(ode_sol, ne_extent, probing_direction="z"):
    """(9, N) solver state -> (rf (4, N) [x, theta, y, phi] on the exit plane, Jf (2, N)) (full_solver.py:838-894)."""
    return engine.ray_to_jones(ode_sol, ne_extent, probing_direction, engine.ROWS_LEGACY, return_E=True)
