"""Prints, for every golden trace fixture, the difference between the float64 build and the mixed-precision
build of the tracer and of both against the reference's tight solution (run on the GPU box; not a test)."""
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from synthpy_amd import engine  # noqa: E402

engine.init(0)
for p in sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "g2_trace_*.npz"))):
    g = np.load(p)
    x, ext, pdir = g["x"], float(g["extent"]), str(g["pdir"])
    vol = engine.Volume.from_ne(g["ne"], x, x, x, float(g["lwl"]), pdir, phaseshift=bool(g["phaseshift"]))
    a = engine.trace(vol, g["s0"], engine.default_t_end(ext), ext, precision="f64")
    b = engine.trace(vol, g["s0"], engine.default_t_end(ext), ext, precision="mixed")
    rt, st = g["rf_tight"], g["sf_tight"]
    d = lambda u, v: float(np.max(np.abs(u - v)))
    print(f"{os.path.basename(p)[:-4]:26s} mixed-f64: pos {d(a[1][0::2], b[1][0::2]):.1e} ang {d(a[1][1::2], b[1][1::2]):.1e} "
          f"ph {d(a[0][7], b[0][7]):.1e} v {d(a[0][3:6], b[0][3:6]):.1e} | vs tight f64: pos {d(a[1][0::2], rt[0::2]):.1e} ang {d(a[1][1::2], rt[1::2]):.1e} "
          f"ph {d(a[0][7], st[7]):.1e} | mixed: pos {d(b[1][0::2], rt[0::2]):.1e} ang {d(b[1][1::2], rt[1::2]):.1e} ph {d(b[0][7], st[7]):.1e}")
