"""Cost of the tracer's second level: a beam that overfills a 256^3 volume (rays entering through the lateral faces in
mid-step are passed by the mixed kernel to the float64 plane kernel), against the same beam held inside the volume.
    python tools/overfill_rate.py [n_rays]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from synthpy_amd import engine as eng
from synthpy_amd.solvers_legacy.full_solver import init_beam

eng.init(0)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2_000_000
n, ext = 256, 5e-3
x = np.linspace(-ext, ext, n)
rng = np.random.default_rng(3)
ne = (1e25 * (1 + 0.3 * np.tanh(rng.standard_normal((n, n, n)).astype(np.float32)))).astype(np.float32)  # n_e > 0
vol = eng.Volume.from_ne(ne, x, x, x, 1064e-9, "z", phaseshift=True)
for name, size, div in (("inside", 0.8 * ext, 5e-5), ("overfill 1.2x, 0.02 rad", 1.2 * ext, 0.02), ("overfill 1.5x, 0.08 rad", 1.5 * ext, 0.08)):
    np.random.seed(5)
    s0 = init_beam(N, size, div, ext, "square", "z")
    rays = eng.RayBundle(N).upload(s0)
    for prec in ("mixed", "f64"):
        rays.trace(vol, eng.default_t_end(ext), ext, precision=prec)
        eng.synchronize()
        t0 = time.perf_counter()
        st = rays.trace(vol, eng.default_t_end(ext), ext, precision=prec)
        eng.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        sf = rays.download()[0]
        print(f"{name:26s} {prec:5s} passed on {st.fallback_rays:8d} of {N}  first kernel {st.trace_kernel_ms:7.2f} ms  "
              f"all levels {st.total_ms:7.2f} ms (wall {wall:7.2f})  finite {np.isfinite(sf).all()}")
