"""The reference's documented flow through the mirror classes on C3-sized inputs (host arrays at every step, as a user of
synthPy writes it, full_solver.py:13-82): where the time goes.   python tools/legacy_flow_rate.py [rays=1e7]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine
from synthpy_amd.solvers_legacy import full_solver as fs
from synthpy_amd.solvers_legacy import rtm_solver as rtm

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10 ** 7
engine.init(0)
ne, x = bench.make_volume(512)
s0 = bench.make_rays(N, 5e-3, 0)


def lap(label, t=[time.perf_counter()]):
    now = time.perf_counter()
    print(f"  {label:<46s} {1e3 * (now - t[0]):8.1f} ms", flush=True)
    t[0] = time.perf_counter()


for rep in range(3):
    print(f"pass {rep}:")
    lap("")
    dom = fs.ScalarDomain(x, x, x, 5e-3, phaseshift=True)
    dom.external_ne(ne)
    dom.calc_dndr(1064e-9)
    lap("ScalarDomain + external_ne + calc_dndr")
    rf, Jf = dom.solve(s0, return_E=True)
    lap("solve(s0, return_E=True)")
    sh = rtm.Shadowgraphy(rf)
    sh.two_lens_solve()
    lap("Shadowgraphy.two_lens_solve")
    sh.histogram(bin_scale=1)
    lap("Shadowgraphy.histogram")
    it = rtm.Interferometry(rf, E=Jf)
    it.two_lens_solve(wl=1064e-9)
    lap("Interferometry.two_lens_solve")
    it.interferogram(bin_scale=1)
    lap("Interferometry.interferogram")
    del rf, Jf, sh, it, dom
