"""Analysis of the detector images (after the path): mirror of radial_2Dspectrum in src/utils/power_spectrum.py:372-421,
the function examples/notebooks/test_ShadowgraphyAnalysis.ipynb applies to shadowgraphs.  The 2-D FFT and the radial
binning run on the GPU (sr_radial_spectrum2d); bin edges and wavenumbers are numpy's, computed as the reference does.
"""
from __future__ import annotations

import numpy as np

from .._ffi import check, lib, ptr


def movingaverage(interval, window_size):
    """power_spectrum.py:190-192."""
    window = np.ones(int(window_size)) / float(window_size)
    return np.convolve(interval, window, "same")


def radial_2Dspectrum(r, lx, ly, smooth=False):
    """Radially averaged power spectrum of a 2-D field r (nx, ny) over a domain lx x ly:
    returns (knyquist, k_centers (99,), spectrum (99,)): |fft2(r)|^2/(nx*ny)^2 averaged in 99 log-spaced bins from the
    smallest non-zero wavenumber to the largest; an empty bin is NaN (np.mean of nothing).

    The reference builds its wavenumber grid with np.meshgrid(kx, ky) (shape (ny, nx)) against a spectrum of shape
    (nx, ny) (power_spectrum.py:398-404): it only runs for square fields and then pairs index (i, j) with
    (ky[i], kx[j]).  That pairing is kept for square fields; a non-square field (a whole 2574 x 3448 detector image)
    pairs (kx[i], ky[j])."""
    r = np.ascontiguousarray(r, dtype=np.float64)
    nx, ny = r.shape
    kx = 2.0 * np.pi * np.fft.fftfreq(nx, d=lx / nx)
    ky = 2.0 * np.pi * np.fft.fftfreq(ny, d=ly / ny)
    k0, k1 = (ky, kx) if nx == ny else (kx, ky)
    k0, k1 = np.ascontiguousarray(k0), np.ascontiguousarray(k1)
    k0s, k1s = np.sort(np.abs(k0)), np.sort(np.abs(k1))
    kmax = float(np.sqrt(k0s[-1] ** 2 + k1s[-1] ** 2))
    pos = [v for v in (k0s[k0s > 0][:1], k1s[k1s > 0][:1]) if len(v)]  # the smallest non-zero k lies on an axis
    kmin = float(min(p[0] for p in pos))
    k_bins = np.logspace(np.log10(kmin), np.log10(kmax), num=100)
    s = np.zeros(len(k_bins) - 1)
    c = np.zeros(len(k_bins) - 1, np.uint64)
    check(lib.sr_radial_spectrum2d(ptr(r), nx, ny, ptr(k0), ptr(k1), ptr(k_bins), len(k_bins), ptr(s), ptr(c)))
    with np.errstate(invalid="ignore", divide="ignore"):
        spectrum = s / c
    k_centers = np.sqrt(k_bins[:-1] * k_bins[1:])
    if smooth:
        spectrum = movingaverage(spectrum, 5)
    return kmax / 2, k_centers, spectrum
