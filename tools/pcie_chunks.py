"""sr_trace on host arrays (C3), page-locked result arrays recycled between calls: the chunk size of the pipelined path.
    python tools/pcie_chunks.py [chunk ...] [--rays N ...]     (most rays per chunk, default 2^21, 3.4e6, 5e6; bundles of N rays, default 1e7)"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine
engine.init(0)
ne, x = bench.make_volume(512)
vol = engine.Volume.from_ne(ne, x, x, x, 1064e-9, "z", phaseshift=True)
argv = sys.argv[1:]
sizes = [10 ** 7]
if "--rays" in argv:
    sizes = [int(float(v)) for v in argv[argv.index("--rays") + 1:]]
    argv = argv[:argv.index("--rays")]
t_end = engine.default_t_end(5e-3)
chunks = [int(float(c)) for c in argv] or [1 << 21, 3_400_000, 5_000_000]
s0 = ref = None
for n_rays, chunk in [(n, c) for n in sizes for c in chunks]:
    if s0 is None or s0.shape[1] != n_rays:
        s0 = bench.make_rays(n_rays, 5e-3, 0)
        ref = None
    os.environ["SYNTHRAY_TRACE_CHUNK"] = str(chunk)
    ts = []
    for rep in range(5):
        t = time.perf_counter()
        out = engine.trace(vol, s0, t_end, 5e-3)
        ts.append((time.perf_counter() - t) * 1e3)
        if rep == 0:
            if ref is None:
                ref = [a.copy() for a in out[:3]]
            else:
                assert all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, out[:3])), "chunk size changed the arrays"
        del out
    print(f"{n_rays} rays, chunk {chunk}: calls {' '.join(f'{t:.1f}' for t in ts)} ms (the first two page-lock the result arrays)", flush=True)
