"""PCIe-inclusive rate of the host-buffer entry point (sr_trace: NumPy in, NumPy out) on C3 — DESIGN.md section 8.
Single pass (SYNTHRAY_TRACE_CHUNK=0) against the pipelined chunks (default 2^21 rays, and 2^20)."""
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
os.environ["SYNTHRAY_TRACE_CHUNK"] = "0"
one = engine.trace(vol, s0, t_end, 5e-3)
os.environ["SYNTHRAY_TRACE_CHUNK"] = str(1 << 21)
many = engine.trace(vol, s0, t_end, 5e-3)
print("pipelined == single pass, bit for bit (sf, rf, Jf):", [bool(np.array_equal(a, b, equal_nan=True)) for a, b in zip(one[:3], many[:3])],
      "ray-steps", one[3].ray_steps, many[3].ray_steps, flush=True)
del one, many
for chunk in ("0", str(1 << 21), str(1 << 20), "0", str(1 << 21)):
    os.environ["SYNTHRAY_TRACE_CHUNK"] = chunk
    for rep in range(3):
        t = time.perf_counter()
        sf, rf, Jf, st = engine.trace(vol, s0, t_end, 5e-3)
        dt = time.perf_counter() - t
        print(f"chunk {chunk:>8}: sr_trace host in/out {dt * 1e3:.1f} ms  ({st.ray_steps / dt:.3e} ray-steps/s, {1e7 / dt:.3e} rays/s); on-device part {st.total_ms:.1f} ms", flush=True)
    t = time.perf_counter()
    _, rf, _, st = engine.trace(vol, s0, t_end, 5e-3, return_E=False, return_sf=False)
    dt = time.perf_counter() - t
    print(f"chunk {chunk:>8}: rf only out {dt * 1e3:.1f} ms ({st.ray_steps / dt:.3e} ray-steps/s)", flush=True)
