"""The launcher's control plane without torch: rendezvous, barrier, small reductions, the 128-byte RCCL id hand-off and
(for CPU tests and the one-GPU rehearsal) host arrays between ranks -- plain TCP sockets, standard library only.

The reference's drivers get this from mpi4py (`comm.barrier()`, `comm.bcast`, `comm.reduce`,
examples/jobs/run_scripts/pvti_trace_mpi.py:23-25, 97, 115, 169-170); the data path here is RCCL (csrc/comm.hip), this
module only gets the ranks to the point where they can create the communicator, and keeps them in step.

Rendezvous.  Ranks are started by `python -m torch.distributed.run`, by bench.py's own spawner or by hand, with
RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment.  MASTER_PORT itself may be taken (torchrun's agent
keeps its own store there), so rank 0 listens on an EPHEMERAL port and publishes "host port token" in a file every
rank can derive, <tmp>/synthray_rdv_<MASTER_ADDR>_<MASTER_PORT>, written atomically once the socket listens.  The other
ranks poll the file and connect; a stale file of an earlier job (nobody listening, or somebody who does not answer with
the token) just means reading it again, until `timeout_s`.  Ranks on other nodes need the file on a shared directory
(SYNTHRAY_RDV_DIR); one node is what the benchmark contract asks for.

Topology: every rank holds one connection to rank 0 (collectives go through it: gather, combine in rank order, send
back -- deterministic sums) and, on demand, direct connections to its pipeline neighbours (send / recv).
"""
from __future__ import annotations

import json
import os
import secrets
import socket
import struct
import tempfile
import time

import numpy as np

_HDR = struct.Struct("<Q")


def _send_msg(sock, payload: bytes):
    sock.sendall(_HDR.pack(len(payload)))
    if payload:
        sock.sendall(payload)


def _recv_exact(sock, n: int) -> bytes:
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("control plane: peer closed the connection")
        got += k
    return bytes(buf)


def _recv_msg(sock) -> bytes:
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return _recv_exact(sock, n) if n else b""


def _array_to_msg(a: np.ndarray) -> bytes:
    a = np.ascontiguousarray(a)
    head = json.dumps({"dtype": a.dtype.str, "shape": list(a.shape)}).encode()
    return _HDR.pack(len(head)) + head + a.tobytes()


def _msg_to_array(m: bytes) -> np.ndarray:
    (h,) = _HDR.unpack(m[:_HDR.size])
    head = json.loads(m[_HDR.size:_HDR.size + h].decode())
    return np.frombuffer(m, dtype=np.dtype(head["dtype"]), offset=_HDR.size + h).reshape(head["shape"]).copy()


def rendezvous_path(addr: str, port: str) -> str:
    d = os.environ.get("SYNTHRAY_RDV_DIR") or tempfile.gettempdir()
    return os.path.join(d, f"synthray_rdv_{addr}_{port}")


class TcpGroup:
    """world ranks joined over TCP.  Collective calls must be made by every rank in the same order."""

    def __init__(self, rank: int, world: int, *, addr=None, port=None, timeout_s: float = 600.0):
        self.rank, self.world, self.timeout_s = int(rank), int(world), float(timeout_s)
        self.addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        self.port = str(port or os.environ.get("MASTER_PORT", "29513"))
        self._path = rendezvous_path(self.addr, self.port)
        self._peers = {}      # rank 0: {rank: socket}; others: {0: socket}
        self._direct = {}     # pipeline neighbours: {rank: socket}
        self._listener = None
        self._table = {}      # rank -> (host, port) of every rank's own listener (direct connections)
        if self.world > 1:
            self._join()

    # ---- rendezvous --------------------------------------------------------------------------------------------
    def _listen(self):
        s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        s.bind((self.addr if self.addr not in ("localhost",) else "127.0.0.1", 0))
        s.listen(max(8, self.world))
        s.settimeout(self.timeout_s)
        return s

    def _join(self):
        deadline = time.time() + self.timeout_s
        self._listener = self._listen()  # every rank: where its pipeline neighbours reach it
        me = (self._listener.getsockname()[0], self._listener.getsockname()[1])
        if self.rank == 0:
            try:
                os.remove(self._path)  # an earlier job's
            except OSError:
                pass
            token = secrets.token_hex(8)
            master = self._listen()
            # the join token is in this file: created exclusively (a planted file or symlink of that name is not followed),
            # readable by this user only, then moved into place
            tmp = f"{self._path}.{os.getpid()}.{secrets.token_hex(4)}.tmp"
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as f:
                f.write(f"{master.getsockname()[0]} {master.getsockname()[1]} {token}")
            os.replace(tmp, self._path)
            self._token = token
            table = {0: me}
            while len(self._peers) < self.world - 1:
                master.settimeout(max(0.1, deadline - time.time()))
                try:
                    c, _ = master.accept()
                except socket.timeout:
                    raise TimeoutError(f"control plane: {self.world - 1 - len(self._peers)} of {self.world} ranks never arrived "
                                       f"({self._path})") from None
                # the hello comes with the connection: a short wait, so that a stray connection that says nothing cannot hold the
                # accept loop beyond the ranks' own 5 s wait for their acknowledgement (they would retry into a queue)
                c.settimeout(min(5.0, self.timeout_s))
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    hello = json.loads(_recv_msg(c).decode())
                except (OSError, ValueError):
                    c.close()
                    continue
                if hello.get("token") != token or hello.get("world") != self.world or not (0 < hello.get("rank", 0) < self.world):
                    c.close()  # not one of this job's ranks
                    continue
                try:
                    _send_msg(c, b"joined " + token.encode())  # at once: the rank knows it reached THIS job's rank 0 and may wait for the table
                except OSError:
                    c.close()  # the rank gave up on this connection (its 5 s were over): it is on its way again
                    continue
                c.settimeout(self.timeout_s)
                old = self._peers.get(hello["rank"])
                if old is not None:  # a rank that came back after its first attempt timed out: the new connection is the live one
                    old.close()
                self._peers[hello["rank"]] = c
                table[hello["rank"]] = tuple(hello["listen"])
            master.close()
            self._table = table
            msg = json.dumps({str(k): list(v) for k, v in table.items()}).encode()
            for r in sorted(self._peers):
                _send_msg(self._peers[r], msg)
        else:
            last_err = None
            while True:
                if time.time() > deadline:
                    raise TimeoutError(f"control plane: rank {self.rank} could not reach rank 0 through {self._path}: {last_err}")
                try:
                    host, port, token = open(self._path).read().split()
                    c = socket.create_connection((host, int(port)), timeout=5.0)
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    _send_msg(c, json.dumps({"rank": self.rank, "world": self.world, "token": token, "listen": list(me)}).encode())
                    # a stale file may name a port that now belongs to somebody else: rank 0 of THIS job acknowledges at once, so
                    # the short timeout stays until that is in (ValueError -> next attempt), the long one only for the table
                    if _recv_msg(c) != b"joined " + token.encode():
                        raise ValueError("the listener is not this job's rank 0")
                    c.settimeout(self.timeout_s)
                    table = json.loads(_recv_msg(c).decode())  # rank 0 sends it once everybody is in
                    self._peers[0] = c
                    self._table = {int(k): tuple(v) for k, v in table.items()}
                    self._token = token
                    break
                except (OSError, ValueError, ConnectionError) as e:  # no file yet, stale file, nobody listening, wrong listener
                    last_err = e
                    time.sleep(0.05)

    # ---- collectives through rank 0 ------------------------------------------------------------------------------
    def _gather(self, payload: bytes):
        """rank 0: [payload of rank 0, 1, ...]; others: None"""
        if self.rank == 0:
            return [payload] + [_recv_msg(self._peers[r]) for r in range(1, self.world)]
        _send_msg(self._peers[0], payload)
        return None

    def _bcast(self, payload):
        if self.rank == 0:
            for r in range(1, self.world):
                _send_msg(self._peers[r], payload)
            return payload
        return _recv_msg(self._peers[0])

    def barrier(self):
        if self.world > 1:
            self._gather(b"")
            self._bcast(b"")

    def bcast_bytes(self, data: bytes = b"") -> bytes:
        """rank 0's bytes on every rank"""
        return self._bcast(data if self.rank == 0 else None) if self.world > 1 else data

    def allreduce(self, value: float, op: str = "sum") -> float:
        if self.world == 1:
            return float(value)
        parts = self._gather(struct.pack("<d", float(value)))
        out = b""
        if self.rank == 0:
            vals = [struct.unpack("<d", p)[0] for p in parts]
            out = struct.pack("<d", max(vals) if op == "max" else sum(vals))
        return struct.unpack("<d", self._bcast(out))[0]

    def reduce_array(self, a: np.ndarray, root: int = 0):
        """Sum of `a` over the ranks (rank order: the same sum on every run) on `root` (every rank if root < 0), None elsewhere."""
        if self.world == 1:
            return a
        parts = self._gather(_array_to_msg(a))
        total = None
        if self.rank == 0:
            total = _msg_to_array(parts[0])
            for p in parts[1:]:
                total = total + _msg_to_array(p)
        if root < 0:
            return _msg_to_array(self._bcast(_array_to_msg(total) if self.rank == 0 else None))
        if root == 0:
            return total
        # the sum goes from rank 0 to the root
        if self.rank == 0:
            _send_msg(self._peers[root], _array_to_msg(total))
            return None
        return _msg_to_array(_recv_msg(self._peers[0])) if self.rank == root else None

    # ---- point to point (the slab pipeline's host transport) -------------------------------------------------------
    def _link(self, peer: int):
        """The direct connection to `peer`: the lower rank connects, the higher rank accepts."""
        if peer in self._direct:
            return self._direct[peer]
        if self.rank < peer:
            c = socket.create_connection(self._table[peer], timeout=self.timeout_s)
            c.settimeout(self.timeout_s)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send_msg(c, json.dumps({"rank": self.rank, "token": self._token}).encode())
        else:
            while True:
                c, _ = self._listener.accept()
                c.settimeout(min(self.timeout_s, 10.0))
                try:  # anybody can connect to the port: a hello that does not parse or carry the token is dropped
                    hello = json.loads(_recv_msg(c).decode())
                    ok = hello.get("token") == self._token and isinstance(hello.get("rank"), int)
                except (ConnectionError, ValueError, socket.timeout, OSError):
                    ok = False
                if not ok:
                    c.close()
                    continue
                c.settimeout(self.timeout_s)
                self._direct[hello["rank"]] = c
                if hello["rank"] == peer:
                    break
            return self._direct[peer]
        self._direct[peer] = c
        return c

    def send(self, a: np.ndarray, dst: int, tag: int = 0):
        _send_msg(self._link(dst), struct.pack("<q", int(tag)) + _array_to_msg(a))

    def recv(self, src: int, tag: int = 0) -> np.ndarray:
        m = _recv_msg(self._link(src))
        (t,) = struct.unpack("<q", m[:8])
        if t != int(tag):
            raise RuntimeError(f"control plane: expected message {tag} from rank {src}, got {t}")
        return _msg_to_array(m[8:])

    def close(self):
        for s in list(self._peers.values()) + list(self._direct.values()):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._direct = {}, {}
        if self._listener is not None:
            self._listener.close()
            self._listener = None
        if self.rank == 0 and self.world > 1:
            try:
                os.remove(self._path)
            except OSError:
                pass
