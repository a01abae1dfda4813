"""The N > 1 path on CPU: world_size 2 (and 3), over gloo (torch.distributed as the control plane: tests only) and over the
product's own TCP control plane (synthpy_amd._rendezvous).  The product's sharding and image-sum logic
(synthpy_amd.distributed) is exercised for real; the per-rank tracing is done by the oracle, since this
box has no GPU (the oracle is the checker here, never the product)."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    from synthpy_amd.distributed import RayShardGroup
    from oracle import oracle as orc

    grp = RayShardGroup(device_images=False, timeout_s=120)
    n, ext, lwl, N = 20, 5e-3, 1064e-9, 3001
    x = np.linspace(-ext, ext, n)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij", sparse=True)
    ne = 1e25 * np.exp(-(X**2 + Y**2 + Z**2) / (1.5e-3) ** 2)
    rng = np.random.default_rng(5)                     # every rank draws the SAME bundle, then takes its shard
    s0 = np.zeros((9, N)); s0[0] = rng.uniform(-3e-3, 3e-3, N); s0[1] = rng.uniform(-3e-3, 3e-3, N)
    s0[2] = -ext; s0[5] = orc.c; s0[6] = 1
    dom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=True)

    def image(rays):
        sf, _ = orc.trace_rk4(dom, rays, (x[1] - x[0]) / orc.c, orc.default_t_end(ext), "z", "planes", 1)
        rf, Jf = orc.ray_to_jones(sf, ext, "z")
        r, E = orc.optics(orc.m_to_mm(rf), orc.chain_shadow_two(), E=Jf, kwave=2 * np.pi / 532e-9)
        return orc.histogram(r, bin_scale=20), orc.interferogram_sums(r, E, bin_scale=20)

    lo, hi = grp.shard(N)
    H, A = image(s0[:, lo:hi])
    Hs = grp.reduce_host(H, root=0)
    As = grp.reduce_host(A, root=-1)
    # the same job cut into stripes of the beam (distributed.shard_stripe, bench.py's strong-scaling cut): the same images
    idx = grp.stripe(s0[0])
    H2, A2 = image(np.ascontiguousarray(s0[:, idx]))
    Hs2 = grp.reduce_host(H2, root=0)
    As2 = grp.reduce_host(A2, root=-1)
    assert grp.sum_over_ranks(float(idx.size)) == N
    assert np.max(np.abs(As2 - As)) <= 1e-9 * np.max(np.abs(As))
    if grp.rank == 0:
        assert np.array_equal(Hs2, Hs), "stripes and index ranges are two partitions of the same rays"
    tmax = grp.max_over_ranks(float(grp.rank + 1))
    tot = grp.sum_over_ranks(float(hi - lo))
    grp.barrier()
    assert tmax == grp.world and tot == N
    if grp.rank == 0:
        H1, A1 = image(s0)
        assert Hs.dtype == np.uint32 and np.array_equal(Hs, H1), "integer counts must be exact at any world size"
        assert np.max(np.abs(As - A1)) <= 1e-9 * np.max(np.abs(A1))
        print("RANK0 OK", int(Hs.sum()))
    else:
        assert Hs is None
    assert np.max(np.abs(As)) > 0
    grp.close()
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_workers(tmp_path, text, world, control="gloo"):
    """control "gloo": torch.distributed's gloo backend plugged in from tests/gloo_plane.py (the package ships the TCP plane only)."""
    script = tmp_path / "worker.py"
    script.write_text(text.format(root=ROOT))
    port = str(_free_port())
    procs = []
    plane = "gloo_plane:GlooPlane" if control == "gloo" else control
    pypath = os.pathsep.join([os.path.join(ROOT, "tests")] + ([os.environ["PYTHONPATH"]] if os.environ.get("PYTHONPATH") else []))
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="2", SYNTHRAY_CONTROL_PLANE=plane, PYTHONPATH=pypath)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
    return outs


@pytest.mark.parametrize("world", [2, 3])
def test_ray_sharding_and_image_sum_over_gloo(tmp_path, orc, world):
    outs = _run_workers(tmp_path, WORKER, world)
    assert "RANK0 OK" in outs[0]


@pytest.mark.parametrize("world", [2, 3])
def test_ray_sharding_and_image_sum_over_tcp(tmp_path, orc, world):
    """The same job over the product's own control plane (no torch in the ranks: checked in the worker)."""
    text = WORKER.replace("grp.close()\n", "grp.close()\nassert 'torch' not in sys.modules, 'the TCP control plane must not import torch'\n")
    outs = _run_workers(tmp_path, text, world, control="tcp")
    assert "RANK0 OK" in outs[0]


SLAB_WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    from synthpy_amd.distributed import RayShardGroup, SlabPipeline
    from synthpy_amd.engine import slab_cuts
    from oracle import oracle as orc

    grp = RayShardGroup(device_images=False, timeout_s=120)
    n, ext, lwl = 22, 5e-3, 1064e-9
    x = np.linspace(-ext, ext, n)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij", sparse=True)
    ne = 1e25 * np.exp(-(X**2 + Y**2 + Z**2) / (1.5e-3) ** 2) * (1 + 0.3 * np.cos(2e3 * X) * np.sin(1.5e3 * Y + 900 * Z))
    dom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=True)   # the oracle indexes planes of the whole domain
    sizes = [300, 257, 300, 41]                                    # ragged chunks

    def rays(ci):
        rng = np.random.default_rng(100 + ci)
        N = sizes[ci]
        s0 = np.zeros((9, N)); s0[0] = rng.uniform(-3e-3, 3e-3, N); s0[1] = rng.uniform(-3e-3, 3e-3, N)
        s0[2] = -ext; s0[3] = 2e4 * rng.standard_normal(N); s0[5] = orc.c; s0[6] = 1
        return s0

    pipe = SlabPipeline(grp, transport="host")
    lo, hi = slab_cuts(n, grp.world)[grp.rank]
    t_end = orc.default_t_end(ext)

    def stage(ci, rec):   # this rank's slab of planes, by the oracle (no GPU here)
        out, _ = orc.trace_slab(dom, t_end, "z", lo, hi, s0=rays(ci) if rec is None else None, rec=rec, last=pipe.last)
        return out

    done = pipe.run(len(sizes), stage, pipe.send_host, pipe.recv_host(lambda ci: sizes[ci]))
    grp.barrier()
    if pipe.last:
        assert len(done) == len(sizes)
        for ci, sf in enumerate(done):
            whole, _ = orc.trace_rk4(dom, rays(ci), (x[1] - x[0]) / orc.c, t_end, "z", "planes", 1)
            assert sf.shape == (9, sizes[ci]) and np.array_equal(sf, whole), ci
        print("LAST RANK OK", grp.world)
    else:
        assert done == []
    grp.close()
""")


@pytest.mark.parametrize("control", ["gloo", "tcp"])
@pytest.mark.parametrize("world", [2, 3])
def test_slab_pipeline_hand_off_over_gloo(tmp_path, orc, world, control):
    """Config 5's exchange step on CPU: rank g holds slab g, ragged chunks flow down the pipeline through the control
    plane's send/recv (gloo, and the product's TCP), the last rank's final states equal the single-pass trace bit for bit."""
    outs = _run_workers(tmp_path, SLAB_WORKER, world, control=control)
    assert f"LAST RANK OK {world}" in outs[-1]


def test_tcp_control_plane_survives_a_stale_rendezvous_file(tmp_path):
    """A rendezvous file left by an earlier job with the same MASTER_PORT (nobody listening on its port) does not stop
    the next job: the ranks read it again until this job's rank 0 has replaced it."""
    from synthpy_amd._rendezvous import rendezvous_path

    port = _free_port()
    dead = _free_port()
    with open(rendezvous_path("127.0.0.1", str(port)), "w") as f:
        f.write(f"127.0.0.1 {dead} deadbeefdeadbeef")
    worker = textwrap.dedent("""
        import os, sys, time
        sys.path.insert(0, {root!r})
        from synthpy_amd._rendezvous import TcpGroup
        import numpy as np
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        if rank == 0:
            time.sleep(0.5)  # the other ranks find the stale file first
        g = TcpGroup(rank, world, timeout_s=60)
        assert g.allreduce(rank + 1.0, "sum") == world * (world + 1) / 2 and g.allreduce(float(rank), "max") == world - 1
        assert g.bcast_bytes(b"id-of-128-bytes" if rank == 0 else b"") == b"id-of-128-bytes"
        tot = g.reduce_array(np.full((3, 2), rank + 1, np.int64), root=1)
        assert (tot is None) == (rank != 1) and (rank != 1 or int(tot[0, 0]) == world * (world + 1) // 2)
        if rank + 1 < world:
            g.send(np.arange(5.0) + rank, rank + 1, tag=7)
        if rank > 0:
            assert np.array_equal(g.recv(rank - 1, tag=7), np.arange(5.0) + rank - 1)
        g.barrier()
        g.close()
        print("OK", rank)
    """)
    script = tmp_path / "w.py"
    script.write_text(worker.format(root=ROOT))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1",
                                                                      MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(3)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert not os.path.exists(rendezvous_path("127.0.0.1", str(port)))  # rank 0 removes its file at close()


def test_shard_range_properties():
    from synthpy_amd.distributed import shard_range

    for n in (0, 1, 7, 10 ** 7, 10 ** 8 + 3):
        for world in (1, 2, 3, 4, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(10, 2, 2)


# ---------------------------------------------------------------- bench.py as the driver launches it
def _run_bench(cmd, env_drop=("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    return subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)


@pytest.mark.parametrize("world", [2, 3])
def test_bench_spawns_its_own_ranks(world):
    """`python bench.py --gpus N` with no torchrun environment: the parent starts N ranks itself (before any GPU call),
    they rendezvous on 127.0.0.1 (the TCP control plane), and stdout is exactly ONE JSON line, exit code 0.  --dry-control-plane
    keeps the library and the GPU out of it: what is tested is that the command line of the driver's scaling leg
    cannot fail on launch."""
    import json

    r = _run_bench([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "1",
                    "--dry-control-plane"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["ranks_seen"] == world and d["steps"] == 2 and d["warmup"] == 1 and d["dry"] is True


def test_bench_under_torchrun_is_one_of_the_ranks():
    """The driver's other form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N (no second spawn)."""
    import json

    r = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                    "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2",
                    "--dry-control-plane"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0])["ranks_seen"] == 2


def test_bench_spawn_reports_a_dead_rank():
    """A rank that dies must end the job with a non-zero code, quickly, instead of leaving the others in a barrier."""
    import time

    import bench

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["SYNTHRAY_BENCH_FAIL_RANK"] = "1"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-control-plane"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0 and time.time() - t0 < 100
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]  # and no result line
    args = bench.parse_args(["--gpus", "8", "--scaling", "strong", "--workload", "c3"])
    assert args.gpus == 8 and args.scaling == "strong"  # the driver's command line parses without a launcher around it


def test_shard_stripe_is_an_equal_count_partition_by_position():
    """distributed.shard_stripe: the strong-scaling shares bench.py cuts by default -- every ray in exactly one stripe, counts that
    differ by at most one, stripes ordered along the coordinate (a rank's rays keep the bundle's density on 1 / world of its area),
    indices ascending (a stripe is the bundle's rays in the bundle's order), NaN positions in the last stripe; one rank: the bundle."""
    from synthpy_amd.distributed import shard_stripe

    rng = np.random.default_rng(11)
    x = rng.normal(size=10007) * 1e-3
    x[[5, 777]] = np.nan
    x[100:110] = x[100]  # ties
    for world in (1, 2, 3, 8):
        parts = [shard_stripe(x, r, world) for r in range(world)]
        assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(x.size))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
        assert all(np.all(np.diff(p) > 0) for p in parts)
        assert 5 in parts[-1] and 777 in parts[-1]
        tops = [np.nanmax(x[p]) for p in parts]
        bots = [np.nanmin(x[p]) for p in parts]
        assert all(tops[k] <= bots[k + 1] for k in range(world - 1))
    assert np.array_equal(shard_stripe(x, 0, 1), np.arange(x.size))
    with pytest.raises(ValueError):
        shard_stripe(x, 3, 3)
    with pytest.raises(ValueError):
        shard_stripe(np.zeros((2, 4)), 0, 2)
    # a circular beam of 1e5 rays in 8 stripes: every stripe at the bundle's density or near it (rays per unit area of its bounding box)
    t, u = rng.random(100000) * 2 * np.pi, np.sqrt(rng.random(100000))
    bx, by = 4e-3 * u * np.cos(t), 4e-3 * u * np.sin(t)
    whole = bx.size / ((bx.max() - bx.min()) * (by.max() - by.min()))
    for r in range(8):
        idx = shard_stripe(bx, r, 8)
        box = (bx[idx].max() - bx[idx].min()) * (by[idx].max() - by[idx].min())
        assert idx.size / box > 0.6 * whole, r  # an index range of the same size: 0.125
    import bench

    args = bench.parse_args(["--gpus", "8", "--scaling", "strong"])
    assert args.shard == "stripe" and args.share_of == 0
    args = bench.parse_args(["--share-of", "8", "--share-rank", "4", "--shard", "index"])
    assert (args.share_of, args.share_rank, args.shard) == (8, 4, "index")


def test_slab_pipeline_plan_with_chunks_cut_by_position():
    """plan_chunks(cut="stripe") + stripe_chunks: a slab pipeline whose chunks are stripes of the beam (every chunk at the job's
    density) is limited from below only by a chunk's fixed cost: BASELINE config 5 over 8 ranks goes in 47 chunks of 2.16e6 rays
    (fill 0.87) where index ranges need 5.4e6 rays to stay dense (19 chunks, 0.73); a sparse job falls back to the index rule."""
    from synthpy_amd.distributed import CHUNK_FIXED_COST_RAYS, plan_chunks, stripe_chunks

    cells = 671_000.0  # lateral cells of 1024^3 under the 4 mm beam's bounding box
    by_index = plan_chunks(10 ** 8, 8, cells, min_density=8.0)
    by_pos = plan_chunks(10 ** 8, 8, cells, min_density=8.0, cut="stripe")
    assert by_index["cut"] == "index" and by_index["chunks"] == 19 and 0.72 < by_index["fill_fraction"] < 0.74
    assert by_pos["cut"] == "stripe" and by_pos["chunk"] == 33 * 65536 and by_pos["chunks"] == 47 and 0.86 < by_pos["fill_fraction"] < 0.88
    assert sum(by_pos["sizes"]) == 10 ** 8 and by_pos["rays_per_beam_cell"] == 10 ** 8 / cells
    rate = lambda c: c / (c + CHUNK_FIXED_COST_RAYS)  # profiles/r05_c5_stripes.txt
    assert by_pos["fill_fraction"] * rate(by_pos["chunk"]) > 1.4 * by_index["fill_fraction"] * 0.705
    assert plan_chunks(10 ** 8, 1, cells, min_density=8.0, cut="stripe")["chunks"] == 1
    assert plan_chunks(10 ** 8, 8, cells, min_density=8.0, cut="stripe", chunk=2_500_000)["chunks"] == 40
    assert plan_chunks(10 ** 6, 8, cells, min_density=8.0, cut="stripe")["cut"] == "index"  # 1.5 rays per cell: sparse whatever the cut
    with pytest.raises(ValueError):
        plan_chunks(10, 2, 1.0, min_density=8.0, cut="diagonal")
    rng = np.random.default_rng(3)
    s0 = rng.normal(size=(9, 1000))
    sizes = [300, 300, 300, 100]
    parts = stripe_chunks(s0, sizes)
    assert [len(p) for p in parts] == sizes and np.array_equal(np.sort(np.concatenate(parts)), np.arange(1000))
    assert all(np.all(np.diff(p) > 0) for p in parts)
    assert all(s0[0, parts[k]].max() <= s0[0, parts[k + 1]].min() for k in range(3))
    with pytest.raises(ValueError):
        stripe_chunks(s0, [500, 400])


def test_bench_strong_scaling_shards_one_bundle():
    """--scaling strong: the ranks' bundles are the contiguous shards of ONE seeded bundle, whatever the world size."""
    from synthpy_amd.distributed import shard_range

    n = 1003
    for world in (1, 2, 3, 8):
        parts = [shard_range(n, r, world) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[k][1] == parts[k + 1][0] for k in range(world - 1))


# ---- SlabPipeline.trace_chunks with the RCCL transport: the two-stream schedule, checked without a GPU ---------------------------
def _schedule_of(rank, world, n_chunks, monkeypatch):
    """Run trace_chunks(rank of world) against a recording stand-in for the engine: every call is an operation queued on the
    selected stream; engine.stream_wait(w, on) makes everything queued on w afterwards depend on what `on` held then.
    Returns (ops, hb): ops = [(stream, kind, chunk, bundle id)], hb(a, b) = "a is complete before b starts" (stream order + waits)."""
    import types

    from synthpy_amd import distributed as dist

    ops, sel, deps, last, counts = [], [0], {}, {0: None, 1: None}, {}

    def queue(kind, bundle=None):
        k = counts.get(kind, 0)
        counts[kind] = k + 1
        idx = len(ops)
        ops.append((sel[0], kind, k, id(bundle) if bundle is not None else None))
        deps[idx] = set() if last[sel[0]] is None else {last[sel[0]]}
        last[sel[0]] = idx
        return idx

    class Bundle:
        bbox = None

        def __init__(self, n):
            self.n = n

        def close(self):
            pass

        def generate(self, **kw):
            queue("draw", self)

        def upload(self, s0):
            queue("draw", self)

        def trace(self, *a, **kw):
            queue("trace", self)

        def deposit(self, *a, **kw):
            queue("deposit", self)

        def handoff_send(self, comm, peer):
            queue("send", self)

        def handoff_recv(self, comm, peer):
            queue("recv", self)

        def trace_stats(self):
            return types.SimpleNamespace(ray_steps=0)

    def stream_wait(waiting, on):
        keep = sel[0]
        sel[0] = waiting
        i = queue("wait")
        if last[on] is not None and last[on] != i:
            deps[i].add(last[on])
        sel[0] = keep

    fake = types.SimpleNamespace(RayBundle=Bundle, select_stream=lambda i: sel.__setitem__(0, i), stream_wait=stream_wait,
                                 synchronize=lambda: None, default_t_end=lambda ext: 1.0, HANDOFF_ENTER=1, HANDOFF_EXIT=2)
    import synthpy_amd

    monkeypatch.setattr(synthpy_amd, "engine", fake, raising=False)
    monkeypatch.setitem(sys.modules, "synthpy_amd.engine", fake)
    grp = types.SimpleNamespace(rank=rank, world=world, _comm=object(), _init_rccl=lambda: None,  # the beam's box: control plane, not a stream
                                send_host=lambda a, dst, tag=0: None, recv_host=lambda shape, src, tag=0: np.full(shape, np.nan))
    pipe = dist.SlabPipeline(grp, transport="rccl")
    pipe.trace_chunks(object(), 1.0, [100] * n_chunks, lambda n, ci: None, deposits=[(object(), [], {})], overlap=True)

    memo = {}

    def before(i):  # everything that is complete before operation i starts
        if i not in memo:
            acc = set()
            for d in deps[i]:
                acc |= {d} | before(d)
            memo[i] = acc
        return memo[i]

    return ops, (lambda a, b: a in before(b))


@pytest.mark.parametrize("world,rank", [(2, 0), (2, 1), (3, 1), (8, 0), (8, 4), (8, 7)])
def test_slab_pipeline_two_stream_schedule(monkeypatch, world, rank):
    """What RCCL between ranks would run, as a graph: (1) every two operations on the SAME bundle (its records are written by the
    receive and the trace, read by the trace and the send) are ordered, in program order; (2) the order is chunk order per peer;
    (3) the overlap is there: the send of chunk k does not hold up the trace of chunk k + 1, and the receive of chunk k + 1 does
    not wait for the trace of chunk k; (4) traces on stream 0, hand-offs on stream 1."""
    n = 6
    ops, hb = _schedule_of(rank, world, n, monkeypatch)
    first, last = rank == 0, rank == world - 1
    where = {(kind, k): i for i, (s, kind, k, b) in enumerate(ops) if kind != "wait"}
    by_bundle = {}
    for i, (s, kind, k, b) in enumerate(ops):
        if b is not None:
            by_bundle.setdefault(b, []).append(i)
    assert len(by_bundle) == 2  # two sets of records, taken in turn
    for seq in by_bundle.values():
        for a, b in zip(seq, seq[1:]):
            assert hb(a, b), (ops[a], ops[b], "two operations on one bundle are not ordered")
    for k in range(n):
        assert ops[where[("trace", k)]][0] == 0
        if not first:
            assert ops[where[("recv", k)]][0] == 1 and hb(where[("recv", k)], where[("trace", k)])
        if not last:
            assert ops[where[("send", k)]][0] == 1 and hb(where[("trace", k)], where[("send", k)])
        else:
            assert hb(where[("trace", k)], where[("deposit", k)])
        if k + 1 < n:
            assert hb(where[("trace", k)], where[("trace", k + 1)])
            if not last:
                assert hb(where[("send", k)], where[("send", k + 1)])
                assert not hb(where[("send", k)], where[("trace", k + 1)]), "the send of a chunk holds up the next chunk's trace"
            if not first:
                assert hb(where[("recv", k)], where[("recv", k + 1)])
                assert not hb(where[("trace", k)], where[("recv", k + 1)]), "the next chunk's receive waits for this chunk's trace"


def test_slab_pipeline_plan_for_eight_ranks():
    """round 4's bench gave 8 ranks 8 chunks of 1.25e7 rays (fill 8 / 15 = 0.53).  The plan is now
    explicit: the smallest chunk the library still traces with the tile kernel (sr_tile_min_density = 8 rays per lateral cell of
    the beam's bounding box: 5.4e6 rays under the 4 mm beam on 1024^3), so 1e8 rays are 19 chunks and eight ranks are busy 73 % of
    the job's time steps.  (The review's 4.2e6 = 8 rays per cell of the beam's DISC is 6.3 per cell of its box, where the per-ray
    kernel is the faster one: profiles/r05_density_crossover.txt -- tie at 10.4 per cell of the box, per-ray kernel 8 % ahead at
    7.4.)  The box is what ranks > 0 are told over the control plane (SlabPipeline.beam_bbox)."""
    from synthpy_amd import distributed as dist
    from synthpy_amd._ffi import lib

    n, ext = 1024, 5e-3
    x = np.linspace(-ext, ext, n)
    box = [-4e-3, -4e-3, -ext, 4e-3, 4e-3, -ext]  # sr_rays_generate's box of a circular beam of radius 4 mm probing z
    cells = dist.beam_cells_of(box, x, x, x, 2)
    assert abs(cells - (8e-3 / (2 * ext / (n - 1)) + 1) ** 2) < 1e-6 * cells
    plan = dist.plan_chunks(1e8, 8, cells)
    assert plan["ranks"] == 8 and plan["chunks"] == len(plan["sizes"]) and sum(plan["sizes"]) == 10 ** 8
    assert plan["rays_per_beam_cell"] >= lib.sr_tile_min_density() and plan["chunk"] % 65536 == 0
    assert plan["fill_fraction"] == plan["chunks"] / (plan["chunks"] + 7) >= 0.72 and plan["chunks"] >= 19, plan
    # the same job cut at round 4's 1.25e7: what the plan replaced
    assert dist.plan_chunks(1e8, 8, cells, chunk=1.25e7)["fill_fraction"] == 8 / 15
    # a job smaller than one dense chunk is one chunk; one rank is always full
    assert dist.plan_chunks(1e6, 8, cells)["chunks"] == 1 and dist.plan_chunks(1e8, 1, cells)["fill_fraction"] == 1.0
    # a beam that overfills the grid: the whole lateral grid
    assert dist.beam_cells_of([-1, -1, -ext, 1, 1, -ext], x, x, x, 2) == (n - 1) ** 2


@pytest.mark.parametrize("world", [2, 3])
def test_slab_pipeline_tells_every_rank_the_beams_box(world, tmp_path):
    """Ranks > 0 receive their rays by hand-off and would judge their density by the whole lateral grid: rank 0 sends the box of
    its first chunk's launch positions round once (host rays here; the control plane is the TCP rendezvous)."""
    worker = textwrap.dedent("""
        import os, sys, types
        sys.path.insert(0, {root!r})
        import numpy as np
        from synthpy_amd import distributed as dist
        import synthpy_amd

        class Bundle:  # no GPU: what trace_chunks asks of the engine
            def __init__(self, n): self.n, self.bbox, self.rec = n, None, None
            def upload(self, s0): self.s0 = s0
            def trace(self, *a, **k): pass
            def deposit(self, *a, **k): pass
            def handoff_download(self): return np.zeros((10, self.n))
            def handoff_upload(self, rec): self.rec = rec
            def trace_stats(self): return types.SimpleNamespace(ray_steps=0)
            def close(self): pass
        made = []
        def make(n):
            made.append(Bundle(n)); return made[-1]
        fake = types.SimpleNamespace(RayBundle=make, synchronize=lambda: None, default_t_end=lambda e: 1.0, HANDOFF_ENTER=1, HANDOFF_EXIT=2)
        synthpy_amd.engine = fake
        sys.modules["synthpy_amd.engine"] = fake
        grp = dist.RayShardGroup(device_images=False, timeout_s=60)
        pipe = dist.SlabPipeline(grp, transport="host")
        calls = []
        def source(n, ci):
            calls.append(ci)
            s0 = np.zeros((9, n)); s0[0] = np.linspace(-3e-3, 2e-3, n); s0[1] = np.linspace(-1e-3, 1e-3, n); s0[2] = -5e-3
            s0[0, 0] = np.nan  # a NaN position does not spoil the box
            return s0
        pipe.trace_chunks(object(), 5e-3, [50, 50, 20], source)
        want = np.array([-3e-3 + 5e-3 / 49, -1e-3 + 2e-3 / 49, -5e-3, 2e-3, 1e-3, -5e-3])  # without the NaN ray
        assert np.allclose(pipe.beam_bbox, want, rtol=0, atol=1e-15), pipe.beam_bbox
        if grp.rank == 0:
            assert calls == [0, 1, 2]  # chunk 0 is asked for once
        else:
            assert not calls and made and all(np.array_equal(b.bbox, pipe.beam_bbox) for b in made)
        grp.close()
        print("OK", grp.rank)
    """)
    script = tmp_path / "w.py"
    script.write_text(worker.format(root=ROOT))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                                                                      MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
