"""Host mirror of src/simulator/beam.py: Beam(...).s0 is the (9, Np) initial ray bundle.
An input of the hot path (host NumPy).  Draw order and seeding follow the reference
(beam.py:63-77: circular = rand for the polar angle, a discarded rand, np.random.power(2, Np) for
the radius, an UNseeded rand for the azimuth, randn for the divergence), so seeded=True gives the
reference's rays.  s0 is float64 (the reference's JAX arrays are float32 unless x64 is enabled)."""
from __future__ import annotations

import numpy as np

from .. import _beam
from .utils import random_array, random_array_n, random_inv_pow_array


class Beam:
    def __init__(self, Np, beam_size, divergence, ne_extent, *, probing_direction="z", wavelength=1064e-9,
                 beam_type="circular", seeded=False):
        """
        Args:
            Np (int): number of photons
            beam_size (float | (float, float)): beam radius / half sizes, m
            divergence (float): beam divergence, rad
            ne_extent (float): half length of the volume on the probing axis, m: rays start at -ne_extent
            probing_direction (str): 'x', 'y' or 'z'
        """
        self.Np = int(Np)
        self.beam_size = beam_size
        self.divergence = divergence
        self.probing_direction = probing_direction
        self.beam_type = beam_type
        self.wavelength = wavelength
        self.init_beam(ne_extent, seeded)

    def init_beam(self, ne_extent, seeded):
        Np, bs, div, pd = self.Np, self.beam_size, self.divergence, self.probing_direction
        if self.beam_type == "circular":
            t = 2 * np.pi * random_array(Np, seeded)
            random_array(Np, seeded)  # drawn and overwritten in the reference (beam.py:70)
            u = random_inv_pow_array(2, Np, seeded)
            phi = np.pi * random_array(Np)
            chi = div * random_array_n(Np, seeded)
            self.s0 = _beam.assemble(bs * u * np.cos(t), bs * u * np.sin(t), chi, phi, ne_extent, pd)
        elif self.beam_type in ("square", "rectangular"):
            t = 2 * random_array(Np, seeded) - 1.0
            u = 2 * random_array(Np, seeded) - 1.0
            phi = np.pi * random_array(Np, seeded)
            chi = div * random_array_n(Np, seeded)
            b1, b2 = (bs, bs) if self.beam_type == "square" else (bs[0], bs[1])
            self.s0 = _beam.assemble(b1 * u, b2 * t, chi, phi, ne_extent, pd)
        elif self.beam_type == "linear":
            t = 2 * random_array(Np, seeded) - 1.0
            chi = div * random_array_n(Np, seeded)
            self.s0 = _beam.assemble_linear(t, chi, bs, ne_extent)
        else:
            raise ValueError(f"beam_type {self.beam_type!r} unrecognised or unfinished in the reference; accepted: "
                             "circular, square, rectangular, linear")

    def save_rays_pos(self, fn=None):
        """Save s0 as .npy (beam.py:305-321)."""
        from datetime import datetime

        fn = "{} rays.npy".format(datetime.now().strftime("%Y-%m-%d_%H-%M-%S")) if fn is None else "{}.npy".format(fn)
        with open(fn, "wb") as f:
            np.save(f, self.s0)
