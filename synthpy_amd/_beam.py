"""Ray-bundle assembly shared by the two API generations (an INPUT of the hot path: host NumPy,
not accelerated).

The random draws are made by the callers in the reference's order, so a seeded call gives
the same rays as the reference; this module only places them in the (9, N) state
    rows 0-2 position (m), 3-5 velocity (m/s, |v| = c), 6 amplitude = 1, 7 phase = 0, 8 polarisation = 0
(src/solvers-legacy/full_solver.py:563-835, src/simulator/beam.py:63-300).
"""
from __future__ import annotations

import numpy as np

c = 299792458.0

# lateral axes (first, second) that carry the beam cross-section for each probing direction
_LATERAL = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}
_AXIS = {"x": 0, "y": 1, "z": 2}


def assemble(p1, p2, chi, phi, ne_extent, probing_direction):
    """Cross-section positions (p1, p2), polar angle chi and azimuth phi of the velocity -> s0 (9, N)."""
    if probing_direction not in _AXIS:
        raise ValueError(f"probing_direction must be 'x', 'y' or 'z', got {probing_direction!r}")
    n = len(chi)
    s0 = np.zeros((9, n))
    a = _AXIS[probing_direction]
    l1, l2 = _LATERAL[probing_direction]
    s0[3 + a] = c * np.cos(chi)
    s0[3 + l1] = c * np.sin(chi) * np.cos(phi)
    s0[3 + l2] = c * np.sin(chi) * np.sin(phi)
    s0[a] = -ne_extent
    s0[l1] = p1
    s0[l2] = p2
    s0[6] = 1.0
    return s0


def assemble_linear(t, chi, beam_size, ne_extent):
    """beam_type 'linear': rays along a line in x, angles in the x-z plane, whatever the probing
    direction (full_solver.py:707-720)."""
    s0 = np.zeros((9, len(chi)))
    s0[3] = c * np.sin(chi)
    s0[5] = c * np.cos(chi)
    s0[0] = beam_size * t
    s0[2] = -ne_extent
    s0[6] = 1.0
    return s0
