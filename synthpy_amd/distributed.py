"""Ray-sharded multi-GPU runs: one process per GPU, contiguous ray shards, the volume replicated in
every GPU's HBM, and ONE collective — the sum of the per-GPU detector images.

Mirrors the reference's MPI drivers (examples/jobs/run_scripts/pvti_trace_mpi.py:111-170,
interference_MPI.py:160-189): every rank traces its own bundle and rank 0 receives
comm.reduce(H, op=SUM).  The per-chunk bcast of the pickled field (pvti_trace_mpi.py:115) is an
artefact of that driver and is not reproduced.

Image sums run in HBM through RCCL over xGMI (sr_image_reduce).  The control plane -- rendezvous, the
128-byte RCCL id hand-off, barriers, the max-over-ranks of a timing -- is plain TCP (_rendezvous.TcpGroup:
standard library only, no torch in the product).  `control="module:callable"` (or SYNTHRAY_CONTROL_PLANE) plugs another
one in from outside the package: the CPU tests of the N > 1 path (world_size 2 and 3, no GPU) run over
torch.distributed's gloo backend that way (tests/gloo_plane.py), next to the same tests over TCP.  Either control plane
can also sum host images, which is what those tests and the one-GPU rehearsal use instead of RCCL.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous shard [lo, hi) of n_items for `rank` of `world`: sizes differ by at most one and the
    concatenation over ranks is 0..n_items in order, so results do not depend on the world size."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_stripe(coord, rank: int, world: int):
    """Indices (ascending) of the rays of `rank`'s STRIPE: the bundle cut into `world` equal-count stripes along one lateral
    coordinate (`coord`: that coordinate of every ray, e.g. s0[0] for a beam probing along z).

    Rays never interact, so any partition gives the same summed image -- the reference cuts by index (`pvti_trace_mpi.py:144-170`,
    shard_range above), where every rank's share covers the whole beam at 1 / world of its density.  On this GPU the density is
    what the trace's speed depends on (the tile path shares a cell's coefficient records between the cell's rays: at 7 rays per
    cell, BASELINE's 1e7 rays over 8 GPUs, a rank runs at 0.66 of the full bundle's rate).  A stripe keeps the full bundle's
    density on 1 / world of the area: the strong-scaling share that costs 1 / world of the time.  Equal
    counts (cuts at the coordinate's quantiles; ties broken by index), NaN coordinates last."""
    coord = np.asarray(coord)
    if coord.ndim != 1:
        raise ValueError("shard_stripe: one coordinate per ray")
    lo, hi = shard_range(coord.shape[0], rank, world)
    if world == 1:
        return np.arange(coord.shape[0])
    order = np.argsort(coord, kind="stable")  # NaNs sort to the end
    return np.sort(order[lo:hi])


def env_rank():
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


def _load_plane(spec, rank, world, timeout_s):
    """A control plane from outside the package: "module:attr" names a callable (rank, world, timeout_s) -> object with
    barrier(), bcast_bytes(data), allreduce(value, op), reduce_array(a, root), send(a, dst, tag), recv(src, tag) or
    recv_into(shape, src, tag), close().  The tests plug torch.distributed's gloo backend in this way (tests/gloo_plane.py);
    the package itself ships the TCP plane only."""
    import importlib

    mod, _, attr = spec.partition(":")
    if not mod or not attr:
        raise ValueError(f"control plane {spec!r}: 'tcp' or 'module:callable'")
    return getattr(importlib.import_module(mod), attr)(rank, world, timeout_s)


class _native_stdout_to_stderr:
    """RCCL announces itself on the C stdout of rank 0 when a communicator is made ("RCCL version : ...", five lines); a job's
    stdout carries its result (bench.py: ONE JSON line), so file descriptor 1 points at stderr meanwhile.  C stdio is flushed on
    both sides of the switch: on a pipe it is block-buffered, and what it still held would come out on the restored descriptor."""

    def __enter__(self):
        import ctypes as C
        import sys

        self._libc = C.CDLL(None)
        sys.stdout.flush()
        self._libc.fflush(None)
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys

        sys.stdout.flush()
        self._libc.fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


class RayShardGroup:
    """The process group of a ray-sharded run."""

    def __init__(self, rank=None, world=None, *, device_images=True, timeout_s=600, control=None):
        erank, elocal, eworld = env_rank()
        self.rank = erank if rank is None else int(rank)
        self.world = eworld if world is None else int(world)
        self.local_rank = elocal
        self.local_world = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))  # ranks on this node (torchrun sets it)
        self.control = control or os.environ.get("SYNTHRAY_CONTROL_PLANE", "tcp")
        self._plane = None
        self._comm = None
        if self.world > 1:
            if self.control.lower() != "tcp":
                self._plane = _load_plane(self.control, self.rank, self.world, timeout_s)
            else:
                from ._rendezvous import TcpGroup

                self._plane = TcpGroup(self.rank, self.world, timeout_s=timeout_s)
        self._device_images = bool(device_images)

    def _init_rccl(self):
        """Create the RCCL communicator on the CURRENT device (call engine.init_rank first);
        collective: every rank reaches it at its first reduce_image."""
        import ctypes as C

        from ._ffi import check, lib

        ident = b""
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            with _native_stdout_to_stderr():
                check(lib.sr_comm_unique_id(buf))
            ident = buf.raw
        ident = self._plane.bcast_bytes(ident)
        h = C.c_void_p()
        with _native_stdout_to_stderr():
            check(lib.sr_comm_create(C.byref(h), ident, self.rank, self.world))
        self._comm = h

    def shard(self, n_items: int):
        return shard_range(n_items, self.rank, self.world)

    def stripe(self, coord):
        """This rank's stripe of a bundle (shard_stripe): indices into the bundle."""
        return shard_stripe(coord, self.rank, self.world)

    def barrier(self):
        if self._plane is not None:
            self._plane.barrier()

    def max_over_ranks(self, value: float) -> float:
        return float(value) if self._plane is None else self._plane.allreduce(value, "max")

    def sum_over_ranks(self, value: float) -> float:
        return float(value) if self._plane is None else self._plane.allreduce(value, "sum")

    def reduce_image(self, image, root=0):
        """Sum a DetectorImage over the ranks in HBM (RCCL); the result lands on `root` (all ranks if root < 0)."""
        if self.world == 1:
            return
        if not self._device_images:
            raise RuntimeError("group was created with device_images=False")
        if self._comm is None:
            self._init_rccl()
        from ._ffi import check, lib

        check(lib.sr_image_reduce(image._h, self._comm, int(root)))

    def comm_ranks(self):
        """(rank, size) as the data-path communicator itself reports them (ncclCommUserRank / ncclCommCount; the control
        plane's when the images are host images): what a job prints as `ranks_seen`."""
        if self.world == 1:
            return 0, 1
        if not self._device_images:
            return self.rank, int(round(self._plane.allreduce(1.0, "sum")))
        if self._comm is None:
            self._init_rccl()
        import ctypes as C

        from ._ffi import check, lib

        r, n = C.c_int(-1), C.c_int(-1)
        check(lib.sr_comm_ranks(self._comm, C.byref(r), C.byref(n)))
        return r.value, n.value

    def reduce_host(self, H: np.ndarray, root=0):
        """Sum a host image over the ranks through the control plane.  Integer counts are summed as int64 (exact);
        returns the sum on `root` (all ranks if root < 0), None elsewhere."""
        if self.world == 1:
            return H
        kind = H.dtype
        work = H.astype(np.int64) if np.issubdtype(kind, np.integer) else np.ascontiguousarray(H)
        out = self._plane.reduce_array(work, root)
        if out is None:
            return None
        return out.astype(kind) if np.issubdtype(kind, np.integer) else out

    def send_host(self, a, dst, tag=0):
        self._plane.send(np.ascontiguousarray(a, dtype=np.float64), dst, tag)

    def recv_host(self, shape, src, tag=0):
        if hasattr(self._plane, "recv_into"):
            return self._plane.recv_into(shape, src, tag)
        a = self._plane.recv(src, tag)
        if tuple(a.shape) != tuple(shape):
            raise RuntimeError(f"received an array of shape {a.shape} from rank {src}, expected {tuple(shape)}")
        return a

    def close(self):
        if self._comm is not None:
            from ._ffi import lib

            lib.sr_comm_destroy(self._comm)
            self._comm = None
        if self._plane is not None:
            self._plane.close()
            self._plane = None


CHUNK_FIXED_COST_RAYS = 3.2e5  # what a chunk costs beside its rays, in rays: measured on BASELINE config 5 (profiles/r05_c5_stripes.txt)


def plan_chunks(n_rays, world, beam_cells, *, min_density=None, chunk=None, cut="index"):
    """How a slab pipeline cuts its job: dict(chunk, sizes, chunks, ranks, fill_fraction, rays_per_beam_cell, cut).

    Rank g starts its first chunk g steps late and idles world-1-g steps at the end: of the chunks + world - 1 time steps a job
    takes, every rank works `chunks` -- fill_fraction = chunks / (chunks + world - 1), which wants MANY chunks.  The kernels
    want DENSE chunks: below sr_tile_min_density() rays per lateral cell of the beam's bounding box a chunk falls back from
    the tile kernel to the per-ray kernel.  beam_cells: lateral cells of the volume under the beam's bounding box
    (beam_cells_of).  The reference's drivers cut at a fixed 5e5 rays (pvti_trace_mpi.py:27).  `chunk` overrides either rule.

    cut="index" (chunks are index ranges of the job's rays -- the device beam's Philox stream, the reference's seeded draws: every
    chunk covers the whole beam at chunk / n_rays of the job's density): the chunk is the smallest that is still dense (rounded up
    to 2^16 rays), the whole job if that is less.

    cut="stripe" (chunks are stripes of the beam, stripe_chunks below: every chunk at the JOB's density, however small): what limits
    a chunk from below is only what it costs beside its rays -- CHUNK_FIXED_COST_RAYS, the binnings and launches of its slab traces.
    Work per rank ~ (chunk + c0) * (n_rays / chunk + world - 1) is least at chunk = sqrt(c0 * n_rays / (world - 1)) (rounded to 2^16;
    BASELINE config 5, 1e8 rays over 8 ranks: 2.16e6 -- 47 chunks, fill 0.87, at 0.87 of the one-chunk rate, where the index rule's 19
    chunks of 5.4e6 run at 0.70 of it, fill 0.73).  A job that is sparse as a whole is cut by the index rule."""
    n_rays, world = int(n_rays), int(world)
    if cut not in ("index", "stripe"):
        raise ValueError("cut must be 'index' or 'stripe'")
    if min_density is None:
        from ._ffi import lib

        min_density = float(lib.sr_tile_min_density())
    if cut == "stripe" and n_rays < min_density * float(beam_cells):
        cut = "index"
    if chunk is None:
        if cut == "index":
            chunk = int(np.ceil(min_density * float(beam_cells) / 65536.0)) * 65536
        elif world > 1:
            chunk = max(1, int(round(np.sqrt(CHUNK_FIXED_COST_RAYS * n_rays / (world - 1)) / 65536.0))) * 65536
        else:
            chunk = n_rays
    chunk = max(1, min(int(chunk), n_rays))
    sizes = [chunk] * (n_rays // chunk) + ([n_rays % chunk] if n_rays % chunk else [])
    return {"chunk": chunk, "sizes": sizes, "chunks": len(sizes), "ranks": world, "cut": cut,
            "fill_fraction": len(sizes) / (len(sizes) + world - 1),
            "rays_per_beam_cell": (n_rays if cut == "stripe" else chunk) / float(beam_cells)}


def stripe_chunks(s0, sizes, axis=0):
    """A host bundle s0 (9, N) cut into chunks of `sizes` rays (their sum N) by POSITION: consecutive equal-count stripes along
    lateral coordinate `axis` (row of s0), each chunk's rays in the bundle's order (shard_stripe's cut for ragged sizes).  Returns the
    list of chunks' index arrays: s0[:, idx] is what SlabPipeline.trace_chunks' ray_source hands out for chunk ci."""
    s0 = np.asarray(s0)
    sizes = [int(m) for m in sizes]
    if s0.ndim != 2 or sum(sizes) != s0.shape[1]:
        raise ValueError("stripe_chunks: the chunk sizes must add up to the bundle's rays")
    order = np.argsort(s0[axis], kind="stable")
    edges = np.concatenate(([0], np.cumsum(sizes)))
    return [np.sort(order[edges[q]:edges[q + 1]]) for q in range(len(sizes))]


def beam_cells_of(bbox, x, y, z, probing_axis=2):
    """Lateral cells of the grid (x, y, z node coordinates) under a beam's bounding box (min x, y, z, max x, y, z): the
    library's own measure (trace.hip: beam_cells)."""
    cells = 1.0
    for q, g in enumerate((x, y, z)):
        if q == probing_axis:
            continue
        n, lo, hi = len(g), max(bbox[q], float(g[0])), min(bbox[3 + q], float(g[-1]))
        width = (float(g[-1]) - float(g[0])) / (n - 1)
        cells *= min((hi - lo) / width + 1.0 if hi > lo else 1.0, n - 1)
    return cells


_BOX_TAG = 1 << 30  # the beam's bounding box on the control plane (chunks use their index as tag)


class SlabPipeline:
    """Slab-decomposed runs (BASELINE config 5; the reference's region loop, propagator.py:366-452): rank g holds the
    node planes cuts[g] of the probing axis, chunks of rays enter at rank 0 and are handed from rank to rank on the
    shared node planes; rank world-1 finishes them (sf / rf / Jf, detector deposit).

    The exchange step is point to point: `send(chunk)` to rank+1 after a chunk's slab is traced, `recv(chunk)` from
    rank-1 before it.  While rank g traces chunk i, rank g-1 traces chunk i+1: after world-1 chunks every GPU is busy.
    Transports: "rccl" (ray records go HBM to HBM over xGMI, sr_rays_handoff_send/_recv on the library stream) and
    "host" (the (10, N) records through the control plane: the CPU tests, and boxes without peer access).
    """

    def __init__(self, group: RayShardGroup, transport="rccl"):
        if transport not in ("rccl", "host"):
            raise ValueError("transport must be 'rccl' or 'host'")
        self.group, self.transport = group, transport
        self.rank, self.world = group.rank, group.world

    @property
    def first(self):
        return self.rank == 0

    @property
    def last(self):
        return self.rank == self.world - 1

    # ---- the schedule, independent of what a stage does (the CPU tests drive it with the oracle's slab trace) ----
    def run(self, n_chunks, stage, send, recv):
        """for every chunk: recv (ranks > 0) -> stage -> send (ranks < world-1).  stage(ci, incoming) returns what
        send gets; returns the list of the last rank's stage results."""
        done = []
        for ci in range(int(n_chunks)):
            incoming = recv(ci) if not self.first else None
            out = stage(ci, incoming)
            if not self.last:
                send(ci, out)
            else:
                done.append(out)
        return done

    # ---- host transport: the records as float64 arrays through the control plane ----
    def send_host(self, ci, rec):
        self.group.send_host(rec, self.rank + 1, tag=ci)

    def recv_host(self, n_rays):
        def recv(ci):
            return self.group.recv_host((10, int(n_rays(ci) if callable(n_rays) else n_rays)), self.rank - 1, tag=ci)

        return recv

    # ---- the GPU stage ----
    def trace_chunks(self, volume, extent, chunk_sizes, ray_source, *, t_end=None, precision="auto", substeps=1,
                     deposits=(), row_order=0, device_beam=None, overlap=None):
        """Trace chunks of rays through this rank's slab `volume`.  ray_source(n, ci) -> s0 (rank 0 only), or
        device_beam = dict(beam_size, divergence, ne_extent, ...) to draw them on rank 0's GPU (RayBundle.generate);
        deposits: [(DetectorImage, chain, kwargs)] applied by the last rank.  Returns (ray_steps, rays_finished).

        Two ray bundles (two sets of hand-off records) alternate from chunk to chunk.  With the RCCL transport and `overlap`
        (SYNTHRAY_SLAB_OVERLAP=1; OFF by default since round 5: the two-stream schedule has only ever run as a recorded graph on
        CPU, never between GPUs -- the serial schedule below it is the one every transport test ran; `self.schedule` says
        which one a call used) the traces run on the library's stream 0 and every ncclSend /
        ncclRecv on stream 1, ordered by events (sr_stream_wait), the receive of chunk k+1 posted before the trace of chunk k
        is queued: the records of chunk k leave, and those of chunk k+1 arrive, while chunk k / k+1 is being traced.  Per chunk k:
          A. stream 1 waits for stream 0 so far (trace k-1 is done with the bundle chunk k+1 arrives in), then recv(k+1)
          B. stream 0: [draw / upload,] trace(k) [, deposits]      (it waited for recv(k) in step C of chunk k-1)
          C. stream 0 waits for stream 1 so far: recv(k+1) has arrived and send(k-1) has left before trace(k+1) touches that bundle
          D. stream 1 waits for stream 0 so far (trace k), then send(k)
        Every wait names work queued EARLIER, so the two queues cannot wait for each other in a circle; between ranks the sends
        and receives of one pair go in chunk order on one stream each.  The host transport (CPU tests, ranks sharing a GPU)
        receives chunk k when it needs it and sends it when it is traced: host-synchronous copies, one stream."""
        from . import engine

        t_end = engine.default_t_end(extent) if t_end is None else t_end
        flags = (0 if self.first else engine.HANDOFF_ENTER) | (0 if self.last else engine.HANDOFF_EXIT)
        rccl = self.transport == "rccl"
        if self.world > 1 and rccl and self.group._comm is None:
            self.group._init_rccl()
        if overlap is None:
            overlap = os.environ.get("SYNTHRAY_SLAB_OVERLAP", "0") == "1"
        overlap = bool(overlap) and rccl and self.world > 1
        self.schedule = "two streams: hand-offs beside the traces" if overlap else "one stream: recv -> trace -> send per chunk"
        n_chunks = len(chunk_sizes)
        bundles, totals = {}, [0, 0]
        # The beam's bounding box goes round ONCE, over the control plane: rays that arrive by hand-off carry none, and a rank
        # that judged their density by the whole lateral grid would trace a dense chunk of a narrow beam with the per-ray
        # kernel (sr_rays_set_bbox; round 4's ranks > 0 did).  Rank 0 knows it from its first chunk: the beam's parameters
        # (device_beam) or the launch positions of the host bundle, which stage(0) then uploads instead of asking again.
        first_s0, have_first, box = None, False, None
        if self.world > 1 and n_chunks:
            if self.first:
                if device_beam is not None:
                    probe = engine.RayBundle(1)
                    probe.generate(first_ray=0, **device_beam)
                    box = probe.bbox
                    probe.close()
                else:
                    first_s0, have_first = ray_source(chunk_sizes[0], 0), True
                    a = None if first_s0 is None else np.asarray(first_s0, dtype=np.float64)
                    if a is not None and a.ndim == 2 and a.shape[0] >= 3 and np.isfinite(a[:3]).all(axis=0).any():
                        ok = np.isfinite(a[:3]).all(axis=0)
                        box = np.concatenate([a[:3, ok].min(axis=1), a[:3, ok].max(axis=1)])
            msg = np.full(7, np.nan) if box is None else np.concatenate([[1.0], box])
            for dst in range(1, self.world) if self.first else ():
                self.group.send_host(msg, dst, tag=_BOX_TAG)
            if not self.first:
                msg = self.group.recv_host((7,), 0, tag=_BOX_TAG)
            box = msg[1:] if msg[0] == 1.0 else None
        self.beam_bbox = box

        def bundle(ci):  # two bundles per chunk size, taken in turn
            key = (chunk_sizes[ci], ci & 1 if self.world > 1 else 0)
            b = bundles.get(key)
            if b is None:
                b = bundles[key] = engine.RayBundle(chunk_sizes[ci])
                if box is not None and not self.first:
                    b.bbox = box
            return b

        def recv(ci):
            rays = bundle(ci)
            if rccl:
                rays.handoff_recv(self.group._comm, self.rank - 1)
            else:
                rays.handoff_upload(self.recv_host(chunk_sizes[ci])(ci))
            return rays

        def send(ci, rays):
            if rccl:
                rays.handoff_send(self.group._comm, self.rank + 1)
            else:
                self.send_host(ci, rays.handoff_download())

        def stage(ci, rays):
            if rays is None:
                rays = bundle(ci)
                if device_beam is not None:
                    rays.generate(first_ray=int(sum(chunk_sizes[:ci])), **device_beam)
                else:
                    rays.upload(first_s0 if (ci == 0 and have_first) else ray_source(chunk_sizes[ci], ci))
            # queued, not waited for: the step counts stay on the device until the end (RayBundle.trace_stats)
            rays.trace(volume, t_end, extent, precision=precision, substeps=substeps, handoff=flags, row_order=row_order,
                       want_stats=False)
            if self.last:
                totals[1] += chunk_sizes[ci]
                for img, chain, kw in deposits:
                    rays.deposit(img, chain, want_stats=False, **kw)
            return rays

        if not overlap:
            self.run(n_chunks, stage, send, recv)
        else:
            TRACE, COMM = 0, 1
            arrived = None
            if not self.first and n_chunks:
                engine.select_stream(COMM)
                arrived = recv(0)
                engine.stream_wait(TRACE, COMM)
            for ci in range(n_chunks):
                nxt = None
                if not self.first and ci + 1 < n_chunks:          # A
                    engine.stream_wait(COMM, TRACE)
                    engine.select_stream(COMM)
                    nxt = recv(ci + 1)
                engine.select_stream(TRACE)                         # B
                rays = stage(ci, arrived)
                engine.stream_wait(TRACE, COMM)                     # C
                if not self.last:                                   # D
                    engine.stream_wait(COMM, TRACE)
                    engine.select_stream(COMM)
                    send(ci, rays)
                arrived = nxt
            engine.select_stream(TRACE)
        engine.synchronize()
        totals[0] = sum(r.trace_stats().ray_steps for r in bundles.values())  # waits for the stream
        engine.synchronize()
        return totals[0], totals[1]
