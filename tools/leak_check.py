"""Repeated create / trace / deposit / destroy cycles: device memory in use must not grow."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthpy_amd import engine as eng
from synthpy_amd.solvers_legacy.full_solver import init_beam

eng.init(0)
hip = C.CDLL("libamdhip64.so")
def used():
    free, total = C.c_size_t(), C.c_size_t()
    hip.hipMemGetInfo(C.byref(free), C.byref(total))
    return (total.value - free.value) / 2 ** 20
x = np.linspace(-5e-3, 5e-3, 64)
X, Y, Z = np.meshgrid(x, x, x, indexing="ij", sparse=True)
ne = 1e25 * np.exp(-(X ** 2 + Y ** 2 + Z ** 2) / (1.5e-3) ** 2)
np.random.seed(0)
s0 = init_beam(100000, 4e-3, 5e-5, 5e-3, "circular", "z")
marks = []
for it in range(60):
    vol = eng.Volume.from_ne(ne, x, x, x, 1064e-9, "z", phaseshift=True)
    if it % 3 == 0:
        vol.attach_aux(np.ones_like(ne), ne, np.zeros(ne.shape + (3,)), 1e-25)
    rays = eng.RayBundle(s0.shape[1]).upload(s0)
    rays.trace(vol, eng.default_t_end(5e-3), 5e-3, precision="mixed" if it % 2 else "f64")
    for img in (eng.DetectorImage.counts(bin_scale=4), eng.DetectorImage.complex_field(bin_scale=4)):
        rays.deposit(img, eng.chain_shadow_two(), kwave=5.9e6 if img.kind else 0.0)
        img.download(); img.close()
    rays.download(); rays.close(); vol.close()
    eng.trace(eng.Volume.from_ne(ne, x, x, x, 1064e-9, "z"), s0[:, :1000], eng.default_t_end(5e-3), 5e-3)
    if it in (9, 59):
        eng.synchronize(); marks.append(used())
print("device MiB in use after 10 and 60 cycles:", [round(m, 1) for m in marks])
assert marks[1] - marks[0] < 64, "device memory grows with the number of cycles"
print("LEAK CHECK OK")
