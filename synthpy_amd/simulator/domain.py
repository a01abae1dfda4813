"""Host mirror of src/simulator/domain.py: ScalarDomain(lengths, dims, ne_type=...).
Grid coordinates, analytic test profiles and external_* loaders: inputs of the hot path (host NumPy).
A plain mutable class (the reference's eqx.Module is frozen, so its external_ne cannot assign,
domain.py:310,453-461).  The reference's memory-driven split of the volume into regions along the probing
axis (domain.py:140-277) is unfinished there (hard-coded 0:65 / 64:128); here `region_count=R` is honoured by
propagator.solve: the volume is built and traced one slab of node planes at a time (R slabs sharing their boundary
planes, the rays handed over on them: same results -- bit for bit in the float64 build, to float32 rounding for a ray
exactly on a cell face at a hand-off plane in the mixed build -- 1/R of the volume in HBM at a time).  auto_batching=True
(the default, as in the reference) sizes R the reference's way -- ceil(estimate * leeway_factor / free memory),
domain.py:166-199 -- with the estimate of what THIS engine holds per node (engine.volume_bytes_estimate) and the GPU's
free HBM (sr_device_memory): 1 for every volume up to ~2300^3 on an empty MI355X.  An explicit region_count > 1 wins.
"""
from __future__ import annotations

import numpy as np


class ScalarDomain:
    def __init__(self, lengths, dims, *, ne_type=None, inv_brems=False, phaseshift=False, B_on=False,
                 probing_direction="z", auto_batching=True, iteration=1, region_count=1, leeway_factor=None,
                 coord_backup=None, future_dims=None, debug=False):
        """
        Args:
            lengths (float | 3 floats): full lengths of the box in x, y, z (m); the box spans -L/2..L/2
            dims (int | 3 ints): number of nodes per axis
            ne_type (str): 'test_null' | 'test_slab' | 'test_linear_cos' | 'test_exponential_cos' | None
        """
        self.ne = self.B = self.Te = self.Z = None
        self.inv_brems, self.phaseshift, self.B_on = inv_brems, phaseshift, B_on
        if probing_direction not in ("x", "y", "z"):
            raise ValueError(f"probing_direction must be 'x', 'y' or 'z', got {probing_direction!r}")
        self.probing_direction = probing_direction
        self.ne_type = ne_type
        self.leeway_factor = 1.1 if leeway_factor is None else leeway_factor
        self.debug = debug
        if np.ndim(lengths) == 0:
            lengths = [lengths] * 3
        if len(lengths) != 3:
            raise Exception("lengths must have len = 3: (x,y,z)")
        if np.ndim(dims) == 0:
            dims = [dims] * 3
        if len(dims) != 3:
            raise Exception("n must have len = 3: (x_n, y_n, z_n)")
        self.lengths = np.array(lengths, dtype=np.float64)
        self.dims = np.array([int(d) for d in dims])
        self.x_length, self.y_length, self.z_length = (float(v) for v in self.lengths)
        self.x_n, self.y_n, self.z_n = (int(v) for v in self.dims)
        self.region_count = max(1, int(region_count))
        self.auto_batching = bool(auto_batching)  # propagator.solve asks regions_for_memory() when region_count is left at 1
        self.coord_backup = None
        self.future_dims = None
        # domain.py:230-232
        self.x = np.float32(np.linspace(-self.x_length / 2, self.x_length / 2, self.x_n))
        self.y = np.float32(np.linspace(-self.y_length / 2, self.y_length / 2, self.y_n))
        self.z = np.float32(np.linspace(-self.z_length / 2, self.z_length / 2, self.z_n))
        self.XX, self.YY, self.ZZ = np.meshgrid(self.x, self.y, self.z, indexing="ij", sparse=True)
        self._volume_cache = None
        if self.ne_type is not None:
            self.generate_electron_density_profile()

    def regions_for_memory(self, free_bytes=None):
        """The reference's auto-batching rule (domain.py:166-199): ceil(estimated allocation * leeway_factor / free memory)
        regions along the probing axis, at most one cell layer each.  free_bytes None: the selected GPU's free HBM; without
        a GPU to ask the answer is 1 (the trace itself will say that there is no device)."""
        from math import ceil

        from .. import engine

        if free_bytes is None:
            try:
                if engine.device_count() < 1:
                    return 1
                free_bytes = engine.device_memory()[0]
            except RuntimeError:
                return 1
        need = engine.volume_bytes_estimate(int(np.prod(self.dims)), self.phaseshift, self.inv_brems, self.B_on) * self.leeway_factor
        axis = "xyz".index(self.probing_direction)
        return int(min(max(1, ceil(need / max(1, free_bytes))), max(1, int(self.dims[axis]) - 1)))

    def _full(self, a):
        return np.ascontiguousarray(np.broadcast_to(a, tuple(self.dims)))

    def generate_electron_density_profile(self):
        fn = {"test_null": self.test_null, "test_slab": self.test_slab, "test_linear_cos": self.test_linear_cos,
              "test_exponential_cos": self.test_exponential_cos}.get(self.ne_type)
        if fn is None:
            raise ValueError(f"unknown ne_type {self.ne_type!r}")
        fn()

    # profiles as domain.py:392-451 (note: scaled by the FULL length, unlike the legacy class)
    def test_null(self):
        self._set(np.zeros(tuple(self.dims), np.float32))

    def test_slab(self, s=1, ne_0=2e23):
        self._set(self._full(ne_0 * (1.0 + s * self.XX / self.x_length)))

    def test_linear_cos(self, s1=0.1, s2=0.1, ne_0=2e23, Ly=1):
        self._set(self._full(ne_0 * (1.0 + s1 * self.XX / self.x_length) * (1 + s2 * np.cos(2 * np.pi * self.YY / Ly))))

    def test_exponential_cos(self, ne_0=1e24, Ly=1e-3, s=2e-3):
        self._set(self._full(ne_0 * (10 ** (self.XX / s) * (1 + np.cos(2 * (np.pi * (self.YY / Ly)))))))

    def _set(self, ne):
        self.ne = ne
        self._volume_cache = None

    def test_B(self, Bmax=1.0):
        """B_z = Bmax * x / x_length, the other components 0 (domain.py:493-503; the reference's in-place write on a
        jnp array cannot run as written)."""
        B = np.zeros(tuple(self.dims) + (3,))
        B[..., 2] = np.broadcast_to(Bmax * self.XX / self.x_length, tuple(self.dims))
        self.B = B

    def external_ne(self, ne):
        """Load an externally generated (x_n, y_n, z_n) grid of n_e in m^-3."""
        ne = np.asarray(ne)
        if ne.shape != tuple(self.dims):
            raise ValueError(f"ne has shape {ne.shape}, the domain is {tuple(self.dims)}")
        self._set(ne)

    def external_B(self, B):
        self.B = B

    def external_Te(self, Te, Te_min=1.0):
        self.Te = np.maximum(Te_min, Te)

    def external_Z(self, Z):
        self.Z = Z

    def export_scalar_field(self, property: str = "ne", fname: str = None):
        """Save n_e as <fname>.vti + <fname>.pvti (domain.py:505-579), written without pyvista."""
        from ..utils.handle_filetypes import export_scalar_field

        export_scalar_field(self, property, fname)

    def cleanup(self):
        self.XX = self.YY = self.ZZ = None
