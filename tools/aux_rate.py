"""Rate of the optional-terms path (inverse bremsstrahlung + Faraday rotation): 1e6 rays x 256^3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine

engine.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10 ** 6
ne, x = bench.make_volume(n)
lwl, ext = 1064e-9, 5e-3
vol = engine.Volume.from_ne(ne, x, x, x, lwl, "z", phaseshift=True)
rays = engine.RayBundle(N).upload(bench.make_rays(N, ext, 0))
t_end = engine.default_t_end(ext)
for prec in ("mixed", "f64"):
    for _ in range(2):
        st = rays.trace(vol, t_end, ext, precision=prec)
    print(f"plain {prec}: kernel {st.trace_kernel_ms:.2f} ms, {st.ray_steps / st.trace_kernel_ms / 1e6:.2f} G ray-steps/s")
X, Y, Z = np.meshgrid(x, x, x, indexing="ij", sparse=True)
B = np.stack(np.broadcast_arrays(3.0 * Y / ext, -2.0 * X / ext + 1.0, 8.0 * (1 + Z / ext) + 0 * X), -1)
kappa = 1e9 * (ne / 1e25) ** 2
for label, kw in (("kappa only", dict(kappa=kappa)), ("Faraday only", dict(ne=ne, B=np.ascontiguousarray(B), verdet=2.62e-13 * lwl ** 2)),
                  ("both", dict(kappa=kappa, ne=ne, B=np.ascontiguousarray(B), verdet=2.62e-13 * lwl ** 2))):
    vol.attach_aux(**kw)
    for _ in range(2):
        st = rays.trace(vol, t_end, ext)
    print(f"{label}: kernel {st.trace_kernel_ms:.2f} ms (whole trace {st.total_ms:.2f} ms: {st.ray_steps / st.total_ms / 1e6:.2f} G ray-steps/s), {st.ray_steps / st.trace_kernel_ms / 1e6:.2f} G ray-steps/s in the kernels, "
          f"tile segments {rays.tile_segments}, {st.fallback_rays} rays through the per-ray kernel, volume {vol.nbytes / 1e9:.2f} GB")
vol2 = engine.Volume.from_ne(ne, x, x, x, lwl, "z", phaseshift=True)
for sub in (2, 4):
    for _ in range(2):
        st = rays.trace(vol2, t_end, ext, substeps=sub, precision="f64")
    print(f"substeps {sub} (no optional terms): kernel {st.trace_kernel_ms:.2f} ms, {st.ray_steps / st.trace_kernel_ms / 1e6:.2f} G ray-steps/s")
