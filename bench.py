#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json: ray-steps/s and rays/s to detector,
1e7 rays through a 512^3 turbulent n_e volume, phase integral + interferogram; config C3).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the path over one batch of rays already resident in HBM:
    bin rays by entry cell -> plane-stepping RK4 trace (+ time-stepping fallback) -> reference beam +
    two-lens optics + complex detector deposit (accumulating into the job's image)
and the timed job = K steps + ONE RCCL sum of the per-GPU images when N > 1 (the reference's drivers sum their
chunks' images locally and reduce once, examples/jobs/run_scripts/pvti_trace_mpi.py:144-170).
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
    ne = 1e25 + 9e24 * noise * float(os.environ.get("SYNTHRAY_BENCH_NOISE", "1"))  # 0: a flat volume (kernel diagnostics)
    x = np.linspace(-5e-3, 5e-3, grid)
    return ne, x


def make_rays(n, ext, seed):
    """Circular beam, radius 4 mm, divergence 5e-5 rad (examples/jobs/run_scripts/test_SynthRayTrace.py:60-63)."""
    from synthpy_amd.solvers_legacy.full_solver import init_beam

    np.random.seed(seed)
    return init_beam(n, 4e-3, 5e-5, ext, "circular", "z")


def upsampled_slab(coarse, f, lo, hi):
    """Planes lo..hi (last axis) of `coarse` refined f times per axis by trilinear interpolation: node i of the fine
    grid sits at coarse coordinate i/f.  Only the slab is ever formed (the 1021^3 whole would be 8.5 GB of float64)."""
    n = coarse.shape[0]
    nf = f * (n - 1) + 1
    pos = np.arange(nf) / f
    i0 = np.minimum(pos.astype(np.int64), n - 2)
    w = pos - i0
    zpos = np.arange(lo, hi + 1) / f
    k0 = np.minimum(zpos.astype(np.int64), n - 2)
    wz = zpos - k0
    a = coarse[:, :, k0] * (1 - wz) + coarse[:, :, k0 + 1] * wz
    a = a[i0] * (1 - w)[:, None, None] + a[i0 + 1] * w[:, None, None]
    return a[:, i0] * (1 - w)[None, :, None] + a[:, i0 + 1] * w[None, :, None]


def bench_c5(args):
    """BASELINE configs[4]: the volume cut into slabs of node planes along the probing axis, one per GPU, chunks of
    rays handed from GPU to GPU on the shared planes (RCCL send/recv), the last GPU deposits.  At N = 1 the one GPU
    holds --slabs slabs and hands over in place: that measures what the cut costs.  A "step" = all --rays rays
    through the whole volume; the rays are drawn on the first slab's GPU (sr_rays_generate: init_beam's distributions,
    Philox stream), so no host upload sits in the pipeline (--host-rays uploads a host bundle per chunk instead, as
    the reference's drivers would)."""
    from synthpy_amd import engine
    from synthpy_amd.distributed import RayShardGroup, SlabPipeline

    grp = RayShardGroup()
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {grp.world}: launch with torch.distributed.run")
    engine.init(grp.local_rank if engine.device_count() > 1 else 0)
    coarse_n, f, ext, lwl = 256, 4, 5e-3, 1064e-9
    n_rays = int(args.rays if args.rays is not None else 1e8)
    ne_c, _ = make_volume(coarse_n)
    n = f * (coarse_n - 1) + 1
    x = np.linspace(-ext, ext, n)
    n_slabs = grp.world if grp.world > 1 else args.slabs
    cuts = engine.slab_cuts(n, n_slabs)
    mine = [cuts[grp.rank]] if grp.world > 1 else cuts

    def slab_volume(lo, hi):
        return engine.Volume.from_ne_slab(upsampled_slab(ne_c, f, max(lo - 1, 0), min(hi + 1, n - 1)), x, x, x, lwl, "z", lo, hi,
                                          phaseshift=True)

    def flags(q, count):
        return (engine.HANDOFF_ENTER if q else 0) | (engine.HANDOFF_EXIT if q + 1 < count else 0)

    t0 = time.time()
    vols = [slab_volume(lo, hi) for lo, hi in mine]
    t_vol = time.time() - t0
    chunk = int(args.chunk)
    sizes = [chunk] * (n_rays // chunk) + ([n_rays % chunk] if n_rays % chunk else [])
    t_end = engine.default_t_end(ext)
    s0_chunk = make_rays(max(sizes), ext, seed=0)  # one host bundle, re-uploaded per chunk (the upload is part of stage 0)
    img = engine.DetectorImage.complex_field(bin_scale=1)
    dep = [(img, engine.chain_shadow_two(), dict(kwave=2 * np.pi / lwl, ref_beam=(10, 10)))]
    pipe = SlabPipeline(grp, transport="rccl")
    beam = dict(beam_size=4e-3, divergence=5e-5, ne_extent=ext, beam_type="circular", probing_direction="z", seed=0)
    kern_ms = []

    def one_pass():
        img.zero()
        if grp.world > 1:
            return pipe.trace_chunks(vols[0], ext, sizes, lambda m, ci: s0_chunk[:, :m], precision=args.precision,
                                     substeps=args.substeps, deposits=dep, device_beam=None if args.host_rays else beam)[0]
        steps, rays, first = 0, {}, 0
        for m in sizes:
            r = rays.get(m) or rays.setdefault(m, engine.RayBundle(m))
            if args.host_rays:
                r.upload(s0_chunk[:, :m])
            else:
                r.generate(first_ray=first, **beam)
            first += m
            for q, v in enumerate(vols):
                st = r.trace(v, t_end, ext, precision=args.precision, substeps=args.substeps, handoff=flags(q, len(vols)))
                steps += st.ray_steps
                kern_ms.append(st.trace_kernel_ms)
            for im, chain, kw in dep:
                r.deposit(im, chain, want_stats=False, **kw)
        engine.synchronize()
        return steps

    for _ in range(args.warmup):
        one_pass()
    engine.synchronize()
    grp.barrier()
    kern_ms.clear()
    t_start = time.perf_counter()
    steps_total = 0
    for _ in range(args.steps):
        steps_total += one_pass()
    engine.synchronize()
    grp.barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t_start)
    all_steps = grp.sum_over_ranks(float(steps_total))
    check = None
    if grp.rank == 0 and grp.world == 1:  # the cut changes nothing: a sample through this chain == through a chain of 2
        ns = min(100000, sizes[0])
        r1 = engine.RayBundle(ns).upload(s0_chunk[:, :ns])
        for q, v in enumerate(vols):
            r1.trace(v, t_end, ext, precision=args.precision, substeps=args.substeps, handoff=flags(q, len(vols)))
        sf_chain = r1.download()[0]
        for v in vols[1:]:
            v.close()
        r2 = engine.RayBundle(ns).upload(s0_chunk[:, :ns])
        for q, (lo, hi) in enumerate(engine.slab_cuts(n, 2)):
            v2 = slab_volume(lo, hi)
            r2.trace(v2, t_end, ext, precision=args.precision, substeps=args.substeps, handoff=flags(q, 2))
            v2.close()
        sf_2 = r2.download()[0]
        # bit for bit in the float64 build; in the mixed build a ray exactly on a cell face at a hand-off plane may be
        # blended from the other cell (same value, float32 rounding): the largest differences are reported with the flag
        check = {"rays": ns, f"chain_of_{len(vols)}_slabs_equals_chain_of_2_bitwise": bool(np.array_equal(sf_chain, sf_2)),
                 "max_dx_m": float(np.nanmax(np.abs(sf_chain[:3] - sf_2[:3]))), "max_dphase_rad": float(np.nanmax(np.abs(sf_chain[7] - sf_2[7]))),
                 "nan_rays": int(np.isnan(sf_chain[0]).sum())}
    if grp.rank == 0:
        per_step_ms = sum(kern_ms) / args.steps if kern_ms else None
        achieved = (steps_total / args.steps) * 512 / (per_step_ms * 1e-3) / 1e9 if kern_ms else None
        out = {
            "metric": "ray-steps/sec (+ rays/sec to detector), slab-decomposed volume with ray hand-off",
            "value": all_steps / elapsed, "unit": "ray-steps/s", "rays_per_s": n_rays * args.steps / elapsed,
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if args.precision == "f64" else "f64 state+accumulation / f32 stage arithmetic", "data": "synthetic",
            "config": {"workload": f"C5: {n_rays:.3g} rays in chunks of {chunk:.3g} x {n}^3 n_e (256^3 k^-11/3 turbulence refined x4), "
                                   f"{n_slabs} slabs of node planes ({'one per GPU, RCCL hand-off' if grp.world > 1 else 'all on one GPU, hand-off in place'}), "
                                   "phase integral + interferogram on the last slab's GPU",
                       "grid": n, "slabs": n_slabs, "chunk": chunk, "volume_setup_s": round(t_vol, 1),
                       "volume_hbm_bytes_this_rank": int(sum(v.nbytes for v in vols[:1])) if check else int(sum(v.nbytes for v in vols))},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                         "kernel": "k_trace_mixed" if args.precision == "mixed" else "k_trace_planes",
                         "kernel_ms_per_step": per_step_ms, "algorithmic_bytes_per_ray_step": 512},
            "cpu_baseline": None, "check": check,
        }
        print(json.dumps(out))
    grp.barrier()
    grp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c3",
                    help="BASELINE.json configs[1..3]: c2 = 1e6 rays x 256^3, shadow + schlieren; c3 = 1e7 x 512^3, interferometry "
                         "(the headline, default); c4 = 1.25e7 rays per GPU (1e8 over 8) x 512^3, all three diagnostics; "
                         "c5 = 1021^3 volume cut into slabs of node planes, one per GPU (--slabs on one GPU when N = 1), rays handed "
                         "from slab to slab (--rays = total rays, default 1e8)")
    ap.add_argument("--rays", type=float, default=None, help="rays per GPU (overrides the workload's)")
    ap.add_argument("--grid", type=int, default=None, help="nodes per axis (overrides the workload's)")
    ap.add_argument("--substeps", type=int, default=1)
    ap.add_argument("--precision", choices=["mixed", "f64"], default="mixed",
                    help="mixed: float64 state/positions/accumulation + float32 stage arithmetic (default); f64: all float64")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-phase", action="store_true", help="shadowgraphy + schlieren deposit instead of the interferogram")
    ap.add_argument("--cpu-sample", type=float, default=2e5, help="rays traced by the CPU baseline (0 = skip)")
    ap.add_argument("--chunk", type=float, default=2.5e6, help="c5: rays per pipeline chunk")
    ap.add_argument("--slabs", type=int, default=8, help="c5 at N = 1: slabs held by the one GPU")
    ap.add_argument("--host-rays", action="store_true", help="c5: upload a host ray bundle per chunk instead of drawing the rays on the GPU")
    args = ap.parse_args()
    if args.workload == "c5":
        return bench_c5(args)
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
        for img, chain, kw in images:  # the image ACCUMULATES over the steps of a job, as the reference's drivers sum
            ms, h = rays.deposit(img, chain, **kw)  # their chunks' images (pvti_trace_mpi.py:144-163)
            dep_ms += ms
            hit += h
        return st, dep_ms, hit

    def reduce_images():  # ONE sum over the ranks per job (pvti_trace_mpi.py:169-170), inside the timed region
        for img, _, _ in images:
            grp.reduce_image(img, root=0)

    for img, _, _ in images:
        img.zero()
    reduce_images()  # untimed: creates the RCCL communicator and its rings whatever --warmup is
    for _ in range(args.warmup):
        one_step()
    for img, _, _ in images:
        img.zero()
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
    reduce_images()
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
            # the detector end of the path: the fused GPU deposit against the oracle's optics + binning fed the SAME ray
            # states (the GPU's own rf / Jf of the sample).  Counts must be equal; the complex image agrees to the
            # rounding of k*|dr| (~6e6 rad per mm as the reference writes it, wavelength in m against mm -- 1e-8 rad of
            # exit angle moves that phase by ~25 rad, which is why images are compared on identical ray states)
            rs = engine.RayBundle(ns).upload(s0[:, :ns])
            rs.trace(vol, t_end, ext, substeps=args.substeps, precision=args.precision)
            _, rf_s, Jf_s = rs.download()
            hc = engine.DetectorImage.counts(bin_scale=1)
            rs.deposit(hc, engine.chain_shadow_two())
            r_mm = orc.optics(rf_s, [(orc.SCALE, 1e3)])[0]
            Ho = orc.histogram(orc.optics(r_mm, orc.chain_shadow_two())[0], bin_scale=1)
            dH = np.abs(hc.download().astype(np.int64) - Ho.astype(np.int64))
            check["H_counts_equal"] = bool(dH.sum() == 0)
            if phase:
                hi_ = engine.DetectorImage.complex_field(bin_scale=1)
                rs.deposit(hi_, engine.chain_shadow_two(), kwave=2 * np.pi / lwl, ref_beam=(10, 10))
                E_o = orc.interfere_ref_beam(rf_s, Jf_s, 10, 10)  # on rf in metres, as diagnostics.py:579-581
                r_o, E_o = orc.optics(r_mm, orc.chain_shadow_two(), E_o, 2 * np.pi / lwl)
                Io = orc.interferogram(r_o, E_o, bin_scale=1)
                check["interferogram_max_dH_over_max_H"] = float(np.max(np.abs(hi_.amplitude() - Io)) / np.max(Io))
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
                         "algorithmic_bytes_per_ray_step": bps, "deposit_kernel_ms": float(np.mean(d_ms)),
                         "note": "frac > 1: the algorithmic bytes (8 corners x 4 stages per ray-step) are served from registers and L1 -- "
                                 "a ray keeps its cell's planes across steps -- so HBM sees `traffic` bytes per launch, not `achieved` x time; "
                                 "what bounds the kernel is VALU issue (DESIGN.md section 8: instructions per wavefront-step, VALU busy %)"},
            "cpu_baseline": cpu,
            "check": check,
        }
        print(json.dumps(out))
    grp.barrier()  # the other ranks wait for rank 0's check before the group goes away
    grp.close()


if __name__ == "__main__":
    main()
