"""Ray-chunked, ray-sharded driver: the job scripts of the reference on one or several MI355X.

Replaces the flow of examples/jobs/run_scripts/pvti_trace_mpi.py:111-187, interference_MPI.py:118-200 and
test_SynthRayTrace.py (argparse -d/--domain, -r/--rays): every rank draws its own ray bundles in chunks (the
reference's 5e5 rays, `Np_ray_split`, pvti_trace_mpi.py:27; 1e7 here by default, see DEFAULT_CHUNK), traces each chunk, pushes it through the diagnostics and ADDS
the chunk's image to the running image; at the end the per-rank images are summed onto rank 0
(`comm.reduce(H, op=MPI.SUM)`, :169-170).  Here the images stay in HBM from the first chunk to the reduce
(RCCL), and the volume is built once per GPU instead of being broadcast with every chunk (:115).

    python -m synthpy_amd.run_trace -d 256 -r 2e6 --ne-type turbulence --diagnostics shadow,schlieren -o out.npz
    python -m torch.distributed.run --nproc-per-node 8 -m synthpy_amd.run_trace -r 1e8 -d 512 ...

As a library:  images = chunked_trace(volume, extent, n_rays, ray_source, diagnostics=[...], group=grp)
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np

from . import engine
from .distributed import RayShardGroup

NP_RAY_SPLIT = int(5e5)  # pvti_trace_mpi.py:27: the reference's chunk (its workers' memory)
# The driver's own default: a GPU traces a DENSE bundle faster per ray -- every chunk reads the part of the volume
# under the beam from HBM once whatever its ray count (measured through 512^3: 3.4 GB + 1.07 GB per 1e6 rays, so 4.7 ms
# per 1e6 rays in chunks of 5e5, 3.1 ms in chunks of 1e7) -- and 1e7 rays hold 2 GB of HBM.  With the
# host ray source each chunk is one seeded init_beam draw, so the chunk size is part of what defines the sample; with
# --device-beam the image does not depend on it.
DEFAULT_CHUNK = int(1e7)
MIN_JOB_CHUNKS = 32  # host ray source: at least this many chunks per job, see default_chunk
# Consecutive chunks of one rank are MERGED on the device into bundles of at least this many rays before they are traced
# (merge_groups): the chunk stays what defines the sample -- every chunk is still its own seeded draw of its own size, so the rays
# are those of the unmerged job -- but the GPU sees dense bundles: the volume under the beam is read from HBM once per bundle,
# and from 8 rays per lateral cell of the beam on the tile kernel runs.  Rays are independent and the images are sums over rays:
# counts are the unmerged job's exactly, the field sums to rounding.  4 * 2^20 rays are 25 rays per cell of the 4 mm beam on 512^3.
MERGE_RAYS = 4 << 20


def default_chunk(n_rays, per_ray_stream=False):
    """The chunk size a job uses when --chunk is not given.  Device beam (the ray index keys the stream, rays are sharded
    one by one): DEFAULT_CHUNK.  Host ray source: whole seeded chunks are dealt to the ranks (rank_chunks), so the job is
    cut into at least MIN_JOB_CHUNKS of them -- 32 whole chunks over 2..8 ranks differ by at most 1.25 between the busiest
    and the idlest rank, where one 1e7-ray chunk would leave seven of eight GPUs without work.  The rule looks at the ray
    count only, never at the number of GPUs: the sample, and with it the image, stays independent of the GPU count."""
    n_rays = int(n_rays)
    if per_ray_stream:
        return DEFAULT_CHUNK
    return max(1, min(DEFAULT_CHUNK, -(-n_rays // MIN_JOB_CHUNKS)))


class Diagnostic:
    """One detector: an optic chain, an image kind and its deposit options."""

    def __init__(self, name, chain, *, complex_field=False, bin_scale=1, pix_x=3448, pix_y=2574, Lx=18.0, Ly=13.5, **deposit):
        self.name, self.chain, self.deposit = name, chain, deposit
        mk = engine.DetectorImage.complex_field if complex_field else engine.DetectorImage.counts
        self.image = mk(bin_scale=bin_scale, pix_x=pix_x, pix_y=pix_y, Lx=Lx, Ly=Ly)
        self.complex_field = complex_field

    def result(self):
        """H as the reference's classes hold it: float64 counts [y, x], or sqrt(Re^2 + Re^2) of the field sums."""
        return self.image.amplitude() if self.complex_field else self.image.download().astype(np.float64)


def standard_diagnostics(names, lwl, bin_scale=1, L=400.0, R=25.0):
    """'shadow' | 'shadow1' | 'schlieren' | 'schlieren_lf' | 'refract' | 'interf' -> Diagnostic objects (the reference's
    fixed chains, rtm_solver.py:197-286, 376-422; reference beam of diagnostics.py:616)."""
    out = []
    for n in names:
        if n == "shadow":
            out.append(Diagnostic(n, engine.chain_shadow_two(L, R), bin_scale=bin_scale))
        elif n == "shadow1":
            out.append(Diagnostic(n, engine.chain_shadow_single(L, R), bin_scale=bin_scale))
        elif n == "schlieren":
            out.append(Diagnostic(n, engine.chain_schlieren(L, R), bin_scale=bin_scale))
        elif n == "schlieren_lf":
            out.append(Diagnostic(n, engine.chain_schlieren(L, R, dark_field=False), bin_scale=bin_scale))
        elif n == "refract":
            out.append(Diagnostic(n, engine.chain_refractometry(L, R), bin_scale=bin_scale))
        elif n == "interf":
            out.append(Diagnostic(n, engine.chain_shadow_two(L, R), complex_field=True, bin_scale=bin_scale,
                                  kwave=2 * np.pi / lwl, ref_beam=(10, 10)))
        else:
            raise ValueError(f"unknown diagnostic {n!r}")
    return out


def chunk_sizes(n_rays, chunk=NP_RAY_SPLIT):
    """The reference's split: the remainder first, then the full chunks (pvti_trace_mpi.py:144-163)."""
    n_rays, chunk = int(n_rays), int(chunk)
    sizes = [n_rays % chunk] if n_rays % chunk else []
    return sizes + [chunk] * (n_rays // chunk)


def rank_chunks(n_rays, chunk, rank, world, per_ray_stream=False):
    """What one rank traces: [(chunk_index, n, first_ray)].  Host ray source (one seeded init_beam draw per chunk): the
    job's GLOBAL chunk list is cut into contiguous runs of whole chunks, so chunk c holds the same rays whatever the
    number of GPUs and the summed image does not depend on it.  Device beam (per_ray_stream: the ray index keys the
    Philox stream): contiguous ray shards, chunked locally."""
    from .distributed import shard_range

    if per_ray_stream:
        lo, hi = shard_range(int(n_rays), rank, world)
        out, first = [], lo
        for ci, n in enumerate(chunk_sizes(hi - lo, chunk)):
            out.append((ci, n, first))
            first += n
        return out
    sizes = chunk_sizes(n_rays, chunk)
    starts = np.concatenate(([0], np.cumsum(sizes)))[:-1] if sizes else []
    c_lo, c_hi = shard_range(len(sizes), rank, world)
    return [(ci, sizes[ci], int(starts[ci])) for ci in range(c_lo, c_hi)]


def merge_groups(chunks, merge_rays=MERGE_RAYS):
    """Consecutive chunks [(chunk_index, n, first_ray)] of one rank grouped into bundles of at least merge_rays rays (the last
    group takes what is left): [[positions in `chunks`], ...].  merge_rays <= 0 or chunks that are large already: one per group."""
    groups, cur, tot = [], [], 0
    for q, (_, n, _) in enumerate(chunks):
        cur.append(q)
        tot += n
        if merge_rays <= 0 or tot >= merge_rays:
            groups.append(cur)
            cur, tot = [], 0
    if cur:
        if groups and merge_rays > 0 and tot < merge_rays // 2:
            groups[-1].extend(cur)  # a short tail joins the bundle before it
        else:
            groups.append(cur)
    return groups


_FARM_SOURCE, _FARM_SLOTS = None, None  # what the forked workers of a RayFarm inherit


def _farm_job(slot, n, ci):
    out = np.ndarray((9, n), np.float64, buffer=_FARM_SLOTS[slot].buf)
    out[...] = _FARM_SOURCE(n, ci)


def host_cores():
    """CPU cores this process may use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


class RayFarm:
    """A host ray source evaluated AHEAD of the trace by forked worker processes.  The reference's drivers draw every chunk
    with NumPy on the host (init_beam after np.random.seed, pvti_trace_mpi.py:144-163): 45 ms per 5e5 rays on one core
    against 4 ms of GPU time.  Every chunk is its own seeded draw (ray_source(n, chunk_index)), so chunks can be drawn
    side by side and the rays are the same whatever the number of workers.  Chunks arrive through shared memory.

    Create it BEFORE the process initialises the GPU or starts threads (it forks); close() it (it owns /dev/shm blocks)."""

    def __init__(self, ray_source, chunks, workers):
        global _FARM_SOURCE, _FARM_SLOTS
        import multiprocessing as mp
        from multiprocessing import shared_memory

        self.order = [(ci, n) for ci, n, _ in chunks]
        self.slots = [shared_memory.SharedMemory(create=True, size=max(8, 72 * max(n for _, n in self.order)))
                      for _ in range(workers + 2)]
        _FARM_SOURCE, _FARM_SLOTS = ray_source, self.slots
        self.pool = mp.get_context("fork").Pool(workers)
        self.pending, self.free, self.next = {}, list(range(len(self.slots))), 0
        self._fill()

    def _fill(self):
        while self.free and self.next < len(self.order):
            slot = self.free.pop()
            ci, n = self.order[self.next]
            self.pending[self.next] = (self.pool.apply_async(_farm_job, (slot, n, ci)), slot, n)
            self.next += 1

    def get(self, q):
        """Chunk number q of the list given at creation (in that order): (s0 view on a shared block, slot)."""
        res, slot, n = self.pending.pop(q)
        res.get()
        return np.ndarray((9, n), np.float64, buffer=self.slots[slot].buf), slot

    def release(self, slot):
        self.free.append(slot)
        self._fill()

    def close(self):
        global _FARM_SOURCE, _FARM_SLOTS
        self.pool.terminate()
        self.pool.join()
        for sm in self.slots:
            sm.close()
            sm.unlink()
        self.slots, _FARM_SOURCE, _FARM_SLOTS = [], None, None


def chunked_trace(volume, extent, n_rays, ray_source, diagnostics, *, chunk=None, group=None, t_end=None,
                  precision=engine.DEFAULT_PRECISION, substeps=1, row_order=engine.ROWS_LEGACY, device_beam=None, streams=None,
                  ray_farm=None, merge_rays=None):
    """Trace this rank's share of n_rays in chunks and accumulate every diagnostic's image in HBM.

    merge_rays: consecutive chunks are put together on the device into bundles of at least this many rays before the trace
    (default MERGE_RAYS; 0: every chunk is traced on its own, as the reference's drivers do) -- see MERGE_RAYS.

    ray_source(n, chunk_index) -> s0 (9, n) with chunk_index counted over the WHOLE job (rank_chunks: the image does
    not depend on the number of GPUs), or device_beam = dict(beam_size, divergence, ne_extent, beam_type,
    probing_direction, seed) to draw the rays on the GPU (RayBundle.generate; the ray index, not the chunking, keys the
    stream, so the image depends neither on chunk size nor on GPU count).  ray_farm: a RayFarm made for this rank's
    rank_chunks(...) — the chunks then come from its workers instead of ray_source.  Returns a dict of totals."""
    group = group or RayShardGroup(rank=0, world=1)
    t_end = engine.default_t_end(extent) if t_end is None else t_end
    chunk = default_chunk(n_rays, device_beam is not None) if chunk is None else int(chunk)
    chunks = rank_chunks(n_rays, chunk, group.rank, group.world, per_ray_stream=device_beam is not None)
    # Two streams, alternating by chunk: chunk i+1's binning and start-up run beside chunk i's tail and deposit, which is
    # what keeps the GPU full when the chunks are small (one 5e5-ray chunk alone runs at ~70 % of the dense rate).  A bundle
    # belongs to one stream; the images are shared (atomic sums) and read only after synchronize().
    groups = merge_groups(chunks, MERGE_RAYS if merge_rays is None else int(merge_rays))
    n_streams = 2 if (streams is None and len(groups) > 1) or (streams or 0) > 1 else 1
    bundles = {}
    tot = dict(rays=0, ray_steps=0, fallback_rays=0, seconds=0.0, bundles=len(groups), chunks=len(chunks))
    engine.synchronize()  # volumes and images created on stream 0 are complete before stream 1 touches them
    t0 = time.perf_counter()
    for g, members in enumerate(groups):
        sid = g % n_streams
        engine.select_stream(sid)
        n = sum(chunks[q][1] for q in members)
        rays = bundles.get((n, sid)) or bundles.setdefault((n, sid), engine.RayBundle(n))
        if device_beam is not None:  # drawn on the GPU (Philox stream keyed by the ray index: consecutive chunks are one draw)
            rays.generate(first_ray=chunks[members[0]][2], **device_beam)
        else:
            at = 0
            for q in members:  # every chunk its own seeded draw, side by side in the bundle (sr_rays_upload_part)
                ci, m, _ = chunks[q]
                if ray_farm is not None:
                    s0, slot = ray_farm.get(q)
                else:
                    s0, slot = ray_source(m, ci), None
                if len(members) == 1:
                    rays.upload(s0)  # returns when the copy is done: the block can be drawn into again
                else:
                    rays.upload_part(s0, at, last=q == members[-1])
                if slot is not None:
                    ray_farm.release(slot)
                at += m
        # no host round trip per chunk: the kernels are queued and the bundle's counters keep adding up
        rays.trace(volume, t_end, extent, substeps=substeps, precision=precision, row_order=row_order, want_stats=False)
        counts = [(d.image, d.chain) for d in diagnostics if not d.complex_field]
        refined = len(counts) > 1
        for k in range(0, len(counts) if refined else 0, 4):  # one float64 re-trace for (up to four) counts diagnostics
            rays.refine(counts[k:k + 4], want_stats=False)
        for d in diagnostics:
            rays.deposit(d.image, d.chain, want_stats=False, exact_counts=not refined, **d.deposit)
        tot["rays"] += n
    engine.synchronize()
    for (n, sid), rays in bundles.items():  # the totals each bundle's counters hold
        engine.select_stream(sid)
        st = rays.trace_stats()
        tot["ray_steps"] += st.ray_steps
        tot["fallback_rays"] += st.fallback_rays
    engine.select_stream(0)
    for d in diagnostics:
        group.reduce_image(d.image, root=0)
    engine.synchronize()
    tot["seconds"] = time.perf_counter() - t0
    return tot


def _load_field(path):
    """n_e (nx, ny, nz) from .pvti / .vti (pvti_readin, as pvti_trace_mpi.py:73-85), a FLASH file (hdf_readin), .npy or
    .npz ('ne'); coordinates +-5 mm unless the .npz carries 'x', 'y', 'z' (m)."""
    if path.endswith((".pvti", ".vti")):
        from .utils.handle_filetypes import pvti_readin

        ne, _, _ = pvti_readin(path)
        xs = (None, None, None)
    elif path.endswith((".h5", ".hdf5")) or "hdf5_plt" in os.path.basename(path) or "hdf5_chk" in os.path.basename(path):
        from .utils.handle_filetypes import hdf_readin  # FLASH AMR file (h5py, or utils/hdf5_lite.py without it); values as the file holds them

        ne, _, _ = hdf_readin(path)
        xs = (None, None, None)
    elif path.endswith(".npz"):
        z = np.load(path)
        ne = z["ne"]
        xs = tuple(z[k] if k in z else None for k in "xyz")
    else:
        ne, xs = np.load(path), (None, None, None)
    coords = [np.linspace(-5e-3, 5e-3, n) if c is None else np.asarray(c) for c, n in zip(xs, ne.shape)]
    return ne, coords


def build_parser():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-d", "--domain", type=int, default=512, help="nodes per axis of a generated volume")
    ap.add_argument("-r", "--rays", type=float, default=1e7, help="total number of rays (all ranks)")
    ap.add_argument("-f", "--force-device", type=str, default=None,
                    help="GPU index (default: LOCAL_RANK); the reference's 'gpu' is accepted, its 'cpu' is refused (no CPU path)")
    ap.add_argument("-c", "--cores", type=int, default=None,
                    help="host cores this job may use (the reference's core_limit, test_SynthRayTrace.py:14): cores - 1 ray workers")
    ap.add_argument("-m", "--memory", type=str, default=None, help="accepted for the reference's command lines; not used")
    ap.add_argument("--field", type=str, default=None, help=".pvti / .vti / .npy / .npz or a FLASH file (.h5 / hdf5_plt_cnt / hdf5_chk) with n_e [m^-3] instead of a generated volume")
    ap.add_argument("--ne-type", default="turbulence",
                    help="turbulence | test_null | test_slab | test_linear_cos | test_exponential_cos")
    ap.add_argument("--diagnostics", default="shadow", help="comma list of shadow,shadow1,schlieren,schlieren_lf,refract,interf")
    ap.add_argument("--bin-scale", type=int, default=1)
    ap.add_argument("--chunk", type=float, default=None,
                    help="rays per chunk (the reference's scripts use 5e5; default: 1e7 with --device-beam, else rays/32 capped at 1e7 "
                         "so that whole seeded chunks spread evenly over the GPUs)")
    ap.add_argument("--merge-rays", type=float, default=None,
                    help=f"consecutive chunks are merged on the device into bundles of at least this many rays before the trace "
                         f"(default {MERGE_RAYS}; 0: every chunk on its own, as the reference's drivers trace them); the rays, and the "
                         f"counts images, do not depend on it")
    ap.add_argument("--beam-size", type=float, default=4e-3)
    ap.add_argument("--divergence", type=float, default=5e-5)
    ap.add_argument("--wavelength", type=float, default=1064e-9)
    ap.add_argument("--probing-direction", default="z", choices=["x", "y", "z"])
    ap.add_argument("--precision", default=engine.DEFAULT_PRECISION, choices=["auto"] + sorted(engine.PRECISIONS))
    ap.add_argument("--substeps", type=int, default=1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device-beam", action="store_true",
                    help="draw the rays on the GPU (same distributions, Philox stream) instead of init_beam on the host")
    ap.add_argument("--streams", type=int, default=None, choices=[1, 2],
                    help="HIP streams the chunks alternate on (default: 2 when there is more than one chunk)")
    ap.add_argument("--ray-workers", type=int, default=None,
                    help="processes that draw the host ray chunks ahead of the trace (default: this rank's share of the cores less "
                         "one when the job has more than two chunks; 0 = draw them in the driver, one after the other)")
    ap.add_argument("-o", "--output", default="synthray_out.npz")
    return ap


def device_choice(force_device):
    """-f / --force-device: a GPU index, or None for the reference's 'gpu' / nothing (= this rank's own GPU,
    engine.init_rank); 'cpu' has no counterpart here."""
    if force_device is None or str(force_device).lower() in ("gpu", "cuda", "rocm", "hip"):
        return None
    if str(force_device).isdigit():
        return int(force_device)
    raise SystemExit(f"--force-device {force_device!r}: a GPU index or 'gpu' (synthpy_amd has no CPU path)")


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.cores is not None and args.ray_workers is None:
        args.ray_workers = max(0, args.cores - 1)

    from .distributed import env_rank
    from .solvers_legacy.full_solver import ScalarDomain, init_beam

    names = [n for n in args.diagnostics.split(",") if n]
    phase = "interf" in names
    pd = args.probing_direction
    if args.field:
        ne, (x, y, z) = _load_field(args.field)
    elif args.ne_type == "turbulence":
        from .field_generator.gaussian3D import gaussian3D

        np.random.seed(1234)
        ne = 1e25 + 9e24 * gaussian3D(lambda k: k ** (-11 / 3)).domain_fft(1.0, 0.01, 5, args.domain // 2, 1.0)
        x = y = z = np.linspace(-5e-3, 5e-3, ne.shape[0])
    else:
        x = y = z = np.linspace(-5e-3, 5e-3, args.domain)
        dom = ScalarDomain(x, y, z, 5e-3)
        getattr(dom, args.ne_type)()
        ne = dom.ne
    extent = float(np.max(np.abs((x, y, z)["xyz".index(pd)])))

    def ray_source(n, ci):
        np.random.seed(args.seed + ci)  # ci counts the job's chunks, not this rank's: the sample does not depend on the GPU count
        return init_beam(n, args.beam_size, args.divergence, extent, "circular", pd)

    # the workers that draw the chunks are forked NOW: nothing has touched the GPU or started a thread yet
    rank, _, world = env_rank()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    args.chunk = default_chunk(args.rays, args.device_beam) if args.chunk is None else int(args.chunk)
    mine = rank_chunks(int(args.rays), int(args.chunk), rank, world)
    from . import _ffi

    if _ffi.gpu_touched:  # main() called from a process that already uses the GPU: no fork from here
        if args.ray_workers:
            raise SystemExit("--ray-workers: this process has already used the GPU; the workers must be forked before that")
        workers = 0
    else:
        workers = args.ray_workers if args.ray_workers is not None else (max(0, host_cores() // local_world - 1) if len(mine) > 2 else 0)
    if workers > 0 and mine:  # the chunks in flight sit in /dev/shm, 72 B per ray, workers + 2 blocks: no more than ~4 GB of them
        workers = min(workers, int(4e9 // (72 * max(n for _, n, _ in mine))) - 2)  # <= 0: even three blocks are too many, draw in the driver
    farm = RayFarm(ray_source, mine, min(workers, len(mine))) if workers > 0 and mine and not args.device_beam else None
    try:
        grp = RayShardGroup()
        engine.init_rank(grp.local_rank, grp.local_world, device=device_choice(args.force_device))
        vol = engine.Volume.from_ne(ne, x, y, z, args.wavelength, pd, phaseshift=phase)
        diags = standard_diagnostics(names, args.wavelength, args.bin_scale)
        dev = dict(beam_size=args.beam_size, divergence=args.divergence, ne_extent=extent, beam_type="circular",
                   probing_direction=pd, seed=args.seed) if args.device_beam else None
        tot = chunked_trace(vol, extent, int(args.rays), ray_source, diags, chunk=int(args.chunk), group=grp,
                            precision=args.precision, substeps=args.substeps, device_beam=dev, streams=args.streams, ray_farm=farm,
                            merge_rays=None if args.merge_rays is None else int(args.merge_rays))
    finally:
        if farm is not None:
            farm.close()
    rays_all = grp.sum_over_ranks(tot["rays"])
    steps_all = grp.sum_over_ranks(tot["ray_steps"])
    secs = grp.max_over_ranks(tot["seconds"])
    if grp.rank == 0:
        out = {d.name: d.result() for d in diags}
        np.savez_compressed(args.output, rays=rays_all, ray_steps=steps_all, seconds=secs, world=grp.world, **out)
        print(f"{int(rays_all)} rays, {int(steps_all)} ray-steps on {grp.world} GPU(s) in {secs:.3f} s "
              f"({rays_all / secs:.3e} rays/s incl. host ray generation and upload) -> {args.output}")
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
