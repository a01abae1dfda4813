"""sr_trace on host arrays (C3): pageable result arrays (what a single call gets) against page-locked ones (a loop), new
arrays each call; the phases of the last call (SYNTHRAY_TRACE_DEBUG)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine
engine.init(0)
ne, x = bench.make_volume(512)
vol = engine.Volume.from_ne(ne, x, x, x, 1064e-9, "z", phaseshift=True)
s0 = bench.make_rays(10 ** 7, 5e-3, 0)
t_end = engine.default_t_end(5e-3)
for mode in ("0", "auto"):
    engine.PINNED_RESULTS = mode
    for rep in range(4):
        t = time.perf_counter()
        out = engine.trace(vol, s0, t_end, 5e-3)
        dt = (time.perf_counter() - t) * 1e3
        del out
        print(f"SYNTHRAY_PINNED_RESULTS={mode} call {rep}: {dt:.1f} ms", flush=True)
