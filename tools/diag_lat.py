import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import golden
from synthpy_amd import engine as eng
from synthpy_amd.solvers_legacy.full_solver import init_beam
from oracle import oracle as orc
eng.init(0)
g = golden("g2_trace_blob32_z_s0")
x, ext = g["x"], float(g["extent"])
np.random.seed(11)
s0 = init_beam(6000, 1.5 * ext, 0.08, ext, "square", "z")
vol = eng.Volume.from_ne(g["ne"], x, x, x, float(g["lwl"]), "z", phaseshift=True)
dom = orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), phaseshift=True)
dt = float(np.float32(x)[1] - np.float32(x)[0]) / orc.c
so, _ = orc.trace_rk4(dom, s0, dt, orc.default_t_end(ext), "z", "planes", 1)
ro, _ = orc.ray_to_jones(so, ext, "z")
for prec in ("mixed", "f64"):
    sf, rf, Jf, st = eng.trace(vol, s0, eng.default_t_end(ext), ext, precision=prec, dt=dt)
    dpos = np.max(np.abs(rf[0::2] - ro[0::2]), axis=0)
    ang = np.hypot(s0[3], s0[4]) / s0[5]
    print(prec, st.fallback_rays, "quantiles", np.quantile(dpos, [0.5, 0.9, 0.99, 0.999, 1.0]))
    for lo, hi in ((0, 0.02), (0.02, 0.05), (0.05, 0.1), (0.1, 1)):
        m = (ang >= lo) & (ang < hi)
        print("  angle", lo, hi, m.sum(), dpos[m].max() if m.any() else None)
