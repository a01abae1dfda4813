"""torch.distributed / gloo as the control plane of a RayShardGroup: test-side only (the product's own is the TCP plane of
synthpy_amd/_rendezvous.py).  Plugged in with SYNTHRAY_CONTROL_PLANE=gloo_plane:GlooPlane (tests/ on PYTHONPATH) or
RayShardGroup(control="gloo_plane:GlooPlane")."""
import os

import numpy as np


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


class GlooPlane:
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
