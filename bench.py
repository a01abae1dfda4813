#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json: ray-steps/s and rays/s to detector,
1e7 rays through a 512^3 turbulent n_e volume, phase integral + interferogram; config C3).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the path over one batch of rays already resident in HBM:
    bin rays by entry cell -> plane-stepping RK4 trace (+ time-stepping fallback) -> reference beam +
    two-lens optics + complex detector deposit  [-> RCCL sum of the images when N > 1]
Every rank traces its own seeded bundle of --rays rays (weak scaling, as the reference's MPI drivers
do) through its own HBM copy of the volume.  torch is used only as the launcher's control plane
(gloo rendezvous, barrier, max over ranks); the data path is libsynthray.so + RCCL.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
BYTES_PER_RAY_STEP = {False: 384, True: 512}  # 4 RHS x 8 corners x 4 B x (3 gradients [+ n])


def host_cores():
    """CPU cores this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def make_volume(grid, seed=1234):
    """n_e = 1e25 + 9e24*noise, noise = k^(-11/3) Gaussian random field of examples/jobs/run_scripts/turb_gen.py:36-50
    (gaussian3D.domain_fft(l_max=1, l_min=0.01, extent=5 mm, res=grid/2)), box +-5 mm, `grid` nodes per axis."""
    from synthpy_amd.field_generator.gaussian3D import gaussian3D

    np.random.seed(seed)
    noise = gaussian3D(lambda k: k ** (-11 / 3)).domain_fft(1.0, 0.01, 5, grid // 2, 1.0)
    ne = 1e25 + 9e24 * noise
    x = np.linspace(-5e-3, 5e-3, grid)
    return ne, x


def make_rays(n, ext, seed):
    """Circular beam, radius 4 mm, divergence 5e-5 rad (examples/jobs/run_scripts/test_SynthRayTrace.py:60-63)."""
    from synthpy_amd.solvers_legacy.full_solver import init_beam

    np.random.seed(seed)
    return init_beam(n, 4e-3, 5e-5, ext, "circular", "z")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["c2", "c3", "c4"], default="c3",
                    help="BASELINE.json configs[1..3]: c2 = 1e6 rays x 256^3, shadow + schlieren; c3 = 1e7 x 512^3, interferometry "
                         "(the headline, default); c4 = 1.25e7 rays per GPU (1e8 over 8) x 512^3, all three diagnostics")
    ap.add_argument("--rays", type=float, default=None, help="rays per GPU (overrides the workload's)")
    ap.add_argument("--grid", type=int, default=None, help="nodes per axis (overrides the workload's)")
    ap.add_argument("--substeps", type=int, default=1)
    ap.add_argument("--precision", choices=["mixed", "f64"], default="mixed",
                    help="mixed: float64 state/positions/accumulation + float32 stage arithmetic (default); f64: all float64")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-phase", action="store_true", help="shadowgraphy + schlieren deposit instead of the interferogram")
    ap.add_argument("--cpu-sample", type=float, default=2e5, help="rays traced by the CPU baseline (0 = skip)")
    args = ap.parse_args()
    wl_rays, wl_grid, wl_diag = {"c2": (1e6, 256, "shadow+schlieren"), "c3": (1e7, 512, "interferometry"),
                                 "c4": (1.25e7, 512, "all")}[args.workload]
    args.rays = wl_rays if args.rays is None else args.rays
    args.grid = wl_grid if args.grid is None else args.grid
    if args.no_phase and wl_diag == "interferometry":
        wl_diag = "shadow+schlieren"

    from synthpy_amd import engine
    from synthpy_amd.distributed import RayShardGroup

    grp = RayShardGroup()
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {grp.world}: launch with torch.distributed.run")
    engine.init(grp.local_rank if engine.device_count() > 1 else 0)

    n_rays, grid, ext, lwl = int(args.rays), args.grid, 5e-3, 1064e-9
    phase = wl_diag != "shadow+schlieren"
    t0 = time.time()
    ne, x = make_volume(grid)
    t_vol = time.time() - t0
    vol = engine.Volume.from_ne(ne, x, x, x, lwl, "z", phaseshift=phase)
    s0 = make_rays(n_rays, ext, seed=grp.rank)
    rays = engine.RayBundle(n_rays).upload(s0)  # inputs resident in HBM before the timed region
    t_end = engine.default_t_end(ext)
    images = []
    if wl_diag in ("interferometry", "all"):
        images.append((engine.DetectorImage.complex_field(bin_scale=1), engine.chain_shadow_two(),
                       dict(kwave=2 * np.pi / lwl, ref_beam=(10, 10))))
    if wl_diag in ("shadow+schlieren", "all"):
        images += [(engine.DetectorImage.counts(bin_scale=1), engine.chain_shadow_two(), {}),
                   (engine.DetectorImage.counts(bin_scale=1), engine.chain_schlieren(), {})]

    def one_step():
        st = rays.trace(vol, t_end, ext, substeps=args.substeps, sort_rays=not args.no_sort, precision=args.precision)
        dep_ms, hit = 0.0, 0
        for img, chain, kw in images:
            img.zero()
            ms, h = rays.deposit(img, chain, **kw)
            dep_ms += ms
            hit += h
            grp.reduce_image(img, root=0)
        return st, dep_ms, hit

    for _ in range(args.warmup):
        one_step()
    engine.synchronize()
    grp.barrier()
    t_start = time.perf_counter()
    k_ms, d_ms, steps_total, hits = [], [], 0, 0
    for _ in range(args.steps):
        st, dep_ms, hit = one_step()
        k_ms.append(st.trace_kernel_ms)
        d_ms.append(dep_ms)
        steps_total += st.ray_steps
        hits = hit
        fallback = st.fallback_rays
    engine.synchronize()
    grp.barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t_start)
    all_steps = grp.sum_over_ranks(float(steps_total))
    all_rays = grp.sum_over_ranks(float(n_rays * args.steps))

    # ---- correctness next to the timing + the CPU baseline (rank 0, N = 1 only for the baseline) ----
    check, cpu = None, None
    if grp.rank == 0:
        ns = int(min(args.cpu_sample, n_rays))
        if ns > 0:
            from oracle import oracle as orc  # the checker / reported CPU baseline, never the product

            orc.build()
            orc.set_num_threads(host_cores())
            sf_g, rf_g, _ = rays.download()
            dom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=phase)
            tc = time.perf_counter()
            sf_o, steps_o = orc.trace_rk4(dom, s0[:, :ns], (x[1] - x[0]) / orc.c, t_end, "z", "planes", args.substeps)
            rf_o, _ = orc.ray_to_jones(sf_o, ext, "z")
            tc = time.perf_counter() - tc
            check = {"rays": ns, "max_dx_m": float(np.max(np.abs(rf_g[0::2, :ns] - rf_o[0::2]))),
                     "max_dtheta_rad": float(np.max(np.abs(rf_g[1::2, :ns] - rf_o[1::2]))),
                     "max_dphase_rad": float(np.max(np.abs(sf_g[7, :ns] - sf_o[7]))), "vs": "oracle (CPU restatement)"}
            if args.gpus == 1:
                cpu = {"value": steps_o / tc, "unit": "ray-steps/s", "cores": orc.num_threads(), "kind": "port",
                       "rays_per_s": ns / tc,
                       "sample": f"first {ns} rays of the same bundle through the same {grid}^3 volume, trace + back-projection, "
                                 f"oracle/synthray_oracle.c with OpenMP over rays, {tc:.1f} s"}

    if grp.rank == 0:
        kern_ms = float(np.mean(k_ms))
        steps_per_launch = steps_total / args.steps
        bps = BYTES_PER_RAY_STEP[phase]
        achieved = steps_per_launch * bps / (kern_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            key = f"{grid}_{n_rays}_{'phase' if phase else 'nophase'}_{args.precision}"
            traffic = tj.get(key, {}).get("bytes_per_launch")
        out = {
            "metric": "ray-steps/sec (+ rays/sec to detector), 1e7 rays x 512^3 volume",
            "value": all_steps / elapsed,
            "unit": "ray-steps/s",
            "rays_per_s": all_rays / elapsed,
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.precision == "f64" else "f64 state+accumulation / f32 stage arithmetic",
            "data": "synthetic",
            "config": {
                "workload": (args.workload.upper() + ": " if (n_rays, grid) == (int(wl_rays), wl_grid) else "") +
                            f"{n_rays:.3g} rays/GPU x {grid}^3 k^-11/3 turbulent n_e (1e25 + 9e24*noise), RK4 {args.substeps} step/cell, " +
                            {"interferometry": "phase integral + reference beam + two-lens interferogram",
                             "shadow+schlieren": "two-lens shadowgraphy + dark-field schlieren",
                             "all": "phase integral; interferogram + two-lens shadowgraphy + dark-field schlieren"}[wl_diag] +
                            ", detector 3448x2574 (bin_scale 1)",
                "rays_per_gpu": n_rays, "grid": grid, "substeps": args.substeps, "sort_rays": not args.no_sort,
                "fallback_rays": int(fallback), "deposited_rays": int(hits),
                "volume_setup_s": round(t_vol, 1), "volume_hbm_bytes": vol.nbytes,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_trace_mixed" if args.precision == "mixed" else "k_trace_planes", "kernel_ms": kern_ms, "ray_steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_ray_step": bps, "deposit_kernel_ms": float(np.mean(d_ms))},
            "cpu_baseline": cpu,
            "check": check,
        }
        print(json.dumps(out))
    grp.barrier()  # the other ranks wait for rank 0's check before the group goes away
    grp.close()


if __name__ == "__main__":
    main()
