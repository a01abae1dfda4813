"""The N > 1 path on CPU: world_size 2 (and 3) over gloo.  The product's sharding and image-sum logic
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


def _run_workers(tmp_path, text, world):
    script = tmp_path / "worker.py"
    script.write_text(text.format(root=ROOT))
    port = str(_free_port())
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=port, OMP_NUM_THREADS="2")
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


@pytest.mark.parametrize("world", [2, 3])
def test_slab_pipeline_hand_off_over_gloo(tmp_path, orc, world):
    """Config 5's exchange step on CPU: rank g holds slab g, ragged chunks flow down the pipeline through gloo
    send/recv, the last rank's final states equal the single-pass trace bit for bit."""
    outs = _run_workers(tmp_path, SLAB_WORKER, world)
    assert f"LAST RANK OK {world}" in outs[-1]


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
