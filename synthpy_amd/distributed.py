"""Ray-sharded multi-GPU runs: one process per GPU, contiguous ray shards, the volume replicated in
every GPU's HBM, and ONE collective — the sum of the per-GPU detector images.

Mirrors the reference's MPI drivers (examples/jobs/run_scripts/pvti_trace_mpi.py:111-170,
interference_MPI.py:160-189): every rank traces its own bundle and rank 0 receives
comm.reduce(H, op=SUM).  The per-chunk bcast of the pickled field (pvti_trace_mpi.py:115) is an
artefact of that driver and is not reproduced.

Image sums run in HBM through RCCL over xGMI (sr_image_reduce).  The control plane -- rendezvous, the
128-byte RCCL id hand-off, barriers, the max-over-ranks of a timing -- is plain TCP (_rendezvous.TcpGroup:
standard library only, no torch in the product).  `control="gloo"` (or SYNTHRAY_CONTROL_PLANE=gloo) puts
torch.distributed's gloo backend in its place: that is what the CPU tests of the N > 1 path run (world_size 2
and 3, no GPU), next to the same tests over TCP.  Either control plane can also sum host images, which is what
those tests and the one-GPU rehearsal use instead of RCCL.
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


class stdout_to_stderr:
    """gloo announces its connections on the C++ stdout ("[Gloo] Rank 0 is connected to ..."); a job's stdout carries
    its result line, so file descriptor 1 points at stderr while the process group is being set up."""

    def __enter__(self):
        import sys

        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys

        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def env_rank():
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when not launched by it."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))


class _GlooPlane:
    """torch.distributed / gloo as the control plane (tests; SYNTHRAY_CONTROL_PLANE=gloo)."""

    def __init__(self, rank, world, timeout_s):
        import datetime

        import torch.distributed as dist

        self.rank, self.world = rank, world
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29513")
            with stdout_to_stderr():
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s))
        self._dist = dist
        self._connected = False

    def _first_contact(self):
        """gloo connects its pairs (and prints) at the first collective: do that one with stdout pointed at stderr."""
        if not self._connected:
            self._connected = True
            with stdout_to_stderr():
                self._dist.barrier()

    def barrier(self):
        self._first_contact()
        self._dist.barrier()

    def bcast_bytes(self, data=b""):
        self._first_contact()
        box = [data if self.rank == 0 else None]
        self._dist.broadcast_object_list(box, src=0)
        return box[0]

    def allreduce(self, value, op="sum"):
        import torch

        self._first_contact()
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX if op == "max" else self._dist.ReduceOp.SUM)
        return float(t[0])

    def reduce_array(self, a, root=0):
        import torch

        self._first_contact()
        work = np.ascontiguousarray(a)
        t = torch.from_numpy(work.view(np.float64).copy() if np.iscomplexobj(work) else work.copy())
        if root < 0:
            self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        else:
            self._dist.reduce(t, dst=root, op=self._dist.ReduceOp.SUM)
            if self.rank != root:
                return None
        out = t.numpy()
        return out.view(np.complex128) if np.iscomplexobj(work) else out

    def send(self, a, dst, tag=0):
        import torch

        self._dist.send(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)), dst=dst, tag=tag)

    def recv_into(self, shape, src, tag=0):
        import torch

        t = torch.empty(tuple(shape), dtype=torch.float64)
        self._dist.recv(t, src=src, tag=tag)
        return t.numpy()

    def close(self):
        if self._dist.is_initialized():
            self._dist.destroy_process_group()


class RayShardGroup:
    """The process group of a ray-sharded run."""

    def __init__(self, rank=None, world=None, *, device_images=True, timeout_s=600, control=None):
        erank, elocal, eworld = env_rank()
        self.rank = erank if rank is None else int(rank)
        self.world = eworld if world is None else int(world)
        self.local_rank = elocal
        self.local_world = int(os.environ.get("LOCAL_WORLD_SIZE", self.world))  # ranks on this node (torchrun sets it)
        self.control = (control or os.environ.get("SYNTHRAY_CONTROL_PLANE", "tcp")).lower()
        if self.control not in ("tcp", "gloo"):
            raise ValueError("control plane must be 'tcp' or 'gloo'")
        self._plane = None
        self._comm = None
        if self.world > 1:
            if self.control == "gloo":
                self._plane = _GlooPlane(self.rank, self.world, timeout_s)
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
            check(lib.sr_comm_unique_id(buf))
            ident = buf.raw
        ident = self._plane.bcast_bytes(ident)
        h = C.c_void_p()
        check(lib.sr_comm_create(C.byref(h), ident, self.rank, self.world))
        self._comm = h

    def shard(self, n_items: int):
        return shard_range(n_items, self.rank, self.world)

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
        if self.control == "gloo":
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
                     deposits=(), row_order=0, device_beam=None):
        """Trace chunks of rays through this rank's slab `volume`.  ray_source(n, ci) -> s0 (rank 0 only), or
        device_beam = dict(beam_size, divergence, ne_extent, ...) to draw them on rank 0's GPU (RayBundle.generate);
        deposits: [(DetectorImage, chain, kwargs)] applied by the last rank.  Returns (ray_steps, rays_finished)."""
        from . import engine

        t_end = engine.default_t_end(extent) if t_end is None else t_end
        flags = (0 if self.first else engine.HANDOFF_ENTER) | (0 if self.last else engine.HANDOFF_EXIT)
        if self.world > 1 and self.transport == "rccl" and self.group._comm is None:
            self.group._init_rccl()
        bundles, totals = {}, [0, 0]

        def bundle(n):
            return bundles.get(n) or bundles.setdefault(n, engine.RayBundle(n))

        def recv(ci):
            rays = bundle(chunk_sizes[ci])
            if self.transport == "rccl":
                rays.handoff_recv(self.group._comm, self.rank - 1)
            else:
                rays.handoff_upload(self.recv_host(chunk_sizes[ci])(ci))
            return rays

        def stage(ci, rays):
            if rays is None:
                rays = bundle(chunk_sizes[ci])
                if device_beam is not None:
                    rays.generate(first_ray=int(sum(chunk_sizes[:ci])), **device_beam)
                else:
                    rays.upload(ray_source(chunk_sizes[ci], ci))
            # queued, not waited for: the step counts stay on the device until the end (RayBundle.trace_stats)
            rays.trace(volume, t_end, extent, precision=precision, substeps=substeps, handoff=flags, row_order=row_order,
                       want_stats=False)
            if self.last:
                totals[1] += chunk_sizes[ci]
                for img, chain, kw in deposits:
                    rays.deposit(img, chain, want_stats=False, **kw)
            return rays

        def send(ci, rays):
            if self.transport == "rccl":
                rays.handoff_send(self.group._comm, self.rank + 1)
            else:
                self.send_host(ci, rays.handoff_download())

        self.run(len(chunk_sizes), stage, send, recv)
        totals[0] = sum(r.trace_stats().ray_steps for r in bundles.values())  # waits for the stream
        engine.synchronize()
        return totals[0], totals[1]
