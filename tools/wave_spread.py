#!/usr/bin/env python3
"""CPU-side design study for the per-wavefront moving window of k_trace_win (trace_win.inc): how far do the 64 rays of one
wavefront spread laterally (in cells) on BASELINE config 3, and what would a W x W window that follows them cost?

Paraxial leapfrog through the bench volume (statistics only, not the engine's integrator):
    python tools/wave_spread.py [grid=512] [rays=1e7] [every=40]
rays: the bundle size that sets rays per cell; every: one wavefront in `every` is followed.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    grid = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n_rays = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    band = 4
    ne, x = bench.make_volume(grid)
    ext, lwl = 5e-3, 1064e-9
    nc = 3.14207787e-4 * (2 * np.pi * 2.99792458e8 / lwl) ** 2
    dx = x[1] - x[0]
    rng = np.random.default_rng(0)
    # uniform disc r = 4 mm, divergence 5e-5 (gaussian angles), as init_beam('circular')
    r = 4e-3 * np.sqrt(rng.random(n_rays))
    t = 2 * np.pi * rng.random(n_rays)
    px, py = r * np.cos(t), r * np.sin(t)
    ib = np.clip(((px + ext) / dx).astype(np.int64), 0, grid - 2)
    ic = np.clip(((py + ext) / dx).astype(np.int64), 0, grid - 2)
    bnd = ib // band
    col = np.where(bnd & 1, grid - 2 - ic, ic)
    key = (bnd * (grid - 1) + col) * band + (ib - bnd * band)
    order = np.argsort(key, kind="stable")
    n_waves = n_rays // 64
    pick = np.arange(0, n_waves, every)
    idx = (pick[:, None] * 64 + np.arange(64)[None, :]).ravel()
    sel = order[idx]
    px, py = px[sel].copy(), py[sel].copy()
    vx = 5e-5 * rng.standard_normal(px.size)
    vy = 5e-5 * rng.standard_normal(px.size)
    nw = pick.size
    print(f"grid {grid}, {n_rays:.3g} rays ({n_rays / (np.pi * (4e-3 / dx) ** 2):.1f} per cell), {nw} wavefronts followed")

    spans = {}
    Ws = (4, 5, 6, 7)
    # moving-window policy: origin (ob, oc) per wave; a ray whose cell is in the window's outer ring asks for a re-centre at the step's end
    state = {W: dict(ob=None, oc=None, retile=0, lost=np.zeros(px.size, bool), wave_steps_with_retile=0) for W in Ws}
    changes = 0
    for k in range(grid - 1):
        cb = np.floor((px + ext) / dx).astype(np.int64)
        cc = np.floor((py + ext) / dx).astype(np.int64)
        cbw, ccw = cb.reshape(nw, 64), cc.reshape(nw, 64)
        if k in (0, 32, 64, 128, 256, 384, grid - 2):
            sb = cbw.max(1) - cbw.min(1) + 1
            sc = ccw.max(1) - ccw.min(1) + 1
            s = np.maximum(sb, sc)
            spans[k] = (np.percentile(s, [50, 90, 99, 99.9]), (sb * sc).mean())
        for W in Ws:
            st = state[W]
            lost = st["lost"].reshape(nw, 64)
            big = 1 << 40
            lo_b = np.where(lost, big, cbw).min(1); hi_b = np.where(lost, -big, cbw).max(1)
            lo_c = np.where(lost, big, ccw).min(1); hi_c = np.where(lost, -big, ccw).max(1)
            if st["ob"] is None:
                st["ob"] = lo_b - (W - (hi_b - lo_b + 1)) // 2
                st["oc"] = lo_c - (W - (hi_c - lo_c + 1)) // 2
            tb = cbw - st["ob"][:, None]
            tc = ccw - st["oc"][:, None]
            ring = ((tb <= 0) | (tb >= W - 1) | (tc <= 0) | (tc >= W - 1)) & ~lost
            need = ring.any(1)
            if need.any():
                # re-centre on the live rays' box; rays that cannot fit (span > W - 2 would re-trigger every step): drop the farthest
                nb_ = lo_b - (W - (hi_b - lo_b + 1)) // 2
                nc_ = lo_c - (W - (hi_c - lo_c + 1)) // 2
                st["ob"] = np.where(need, nb_, st["ob"])
                st["oc"] = np.where(need, nc_, st["oc"])
                st["retile"] += int(need.sum())
                tb = cbw - st["ob"][:, None]
                tc = ccw - st["oc"][:, None]
                out = ((tb < 0) | (tb >= W) | (tc < 0) | (tc >= W)) & ~lost
                lost |= out
        # paraxial kick + drift through one cell layer (bilinear gradient of n_e/n_c on plane k)
        gx, gy = np.gradient(ne[:, :, k] / nc, dx)
        fx = (px + ext) / dx
        fy = (py + ext) / dx
        i0 = np.clip(np.floor(fx).astype(np.int64), 0, grid - 2)
        j0 = np.clip(np.floor(fy).astype(np.int64), 0, grid - 2)
        wx, wy = fx - i0, fy - j0

        def bil(g):
            return (g[i0, j0] * (1 - wx) * (1 - wy) + g[i0 + 1, j0] * wx * (1 - wy) + g[i0, j0 + 1] * (1 - wx) * wy + g[i0 + 1, j0 + 1] * wx * wy)

        vx -= 0.5 * bil(gx) * dx
        vy -= 0.5 * bil(gy) * dx
        px += vx * dx
        py += vy * dx
        nb2 = np.floor((px + ext) / dx).astype(np.int64)
        nc2 = np.floor((py + ext) / dx).astype(np.int64)
        changes += int(((nb2 != cb) | (nc2 != cc)).reshape(nw, 64).any(1).sum())
    print(f"exit angles rms {np.sqrt((vx ** 2 + vy ** 2).mean()) * 1e3:.2f} mrad; wavefront-steps with a cell change: {changes / (nw * (grid - 1)):.2f}")
    print("plane: span of a wavefront's cells (max of both axes) p50 p90 p99 p99.9 | mean box area")
    for k, (p, a) in spans.items():
        print(f"  {k:4d}: {p}  | {a:.1f}")
    for W in Ws:
        st = state[W]
        print(f"W={W}: re-centres per wavefront {st['retile'] / nw:.2f} (one per {nw * (grid - 1) / max(1, st['retile']):.0f} steps), rays lost {st['lost'].mean() * 100:.2f} %")


if __name__ == "__main__":
    main()
