"""The reference's documented flow through the mirror classes on BASELINE-sized inputs (host arrays at every step, as a user
of synthPy writes it: full_solver.py:13-82, pvti_trace_mpi.py:111-131): where the time goes, with the diagnostics depositing
from the bundle solve() left in HBM (synthpy_amd/resident.py) and -- SYNTHRAY_RESIDENT=0 -- through host arrays as before.
    python tools/legacy_flow_rate.py [c3|c2] [rays]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine, resident
from synthpy_amd.solvers_legacy import full_solver as fs
from synthpy_amd.solvers_legacy import rtm_solver as rtm

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
grid, N, phase = {"c3": (512, 10 ** 7, True), "c2": (256, 10 ** 6, False)}[wl]
if len(sys.argv) > 2:
    N = int(float(sys.argv[2]))
engine.init(0)
ne, x = bench.make_volume(grid)
s0 = bench.make_rays(N, 5e-3, 0)
print(f"# {wl}: {N} rays x {grid}^3, {engine._ffi.lib.sr_version().decode()}")


def lap(label, t=[time.perf_counter()]):
    now = time.perf_counter()
    if label:
        print(f"  {label:<58s} {1e3 * (now - t[0]):8.1f} ms", flush=True)
    t[0] = time.perf_counter()
    return 1e3 * (now - t[0])


for mode in ("1", "0"):
    resident.MODE = mode
    print(f"SYNTHRAY_RESIDENT={mode}: " + ("diagnostics deposit from the bundle solve() left in HBM" if mode == "1" else "host arrays at every step (round 3)"))
    for rep in range(3):
        print(f" pass {rep}:")
        t_pass = time.perf_counter()
        lap("")
        dom = fs.ScalarDomain(x, x, x, 5e-3, phaseshift=phase)
        dom.external_ne(ne)
        dom.calc_dndr(1064e-9)
        lap("ScalarDomain + external_ne + calc_dndr")
        t_flow = time.perf_counter()
        if phase:
            rf, Jf = dom.solve(s0, return_E=True)
            lap(f"solve(s0, return_E=True)   [trace kernels {dom.trace_stats.trace_kernel_ms:.1f} ms, tile segments {dom._rays.tile_segments}]")
        else:
            rf, Jf = dom.solve(s0), None
            lap(f"solve(s0)   [trace kernels {dom.trace_stats.trace_kernel_ms:.1f} ms, tile segments {dom._rays.tile_segments}]")
        sh = rtm.Shadowgraphy(rf)
        sh.two_lens_solve()
        lap(f"Shadowgraphy(rf).two_lens_solve   [on device: {sh.on_device}]")
        sh.histogram(bin_scale=1)
        lap("Shadowgraphy.histogram(bin_scale=1)")
        sc = rtm.Schlieren(rf)
        sc.DF_solve()
        sc.histogram(bin_scale=1)
        lap(f"Schlieren(rf).DF_solve + histogram   [on device: {sc.on_device}]")
        if phase:
            it = rtm.Interferometry(rf, E=Jf)
            it.two_lens_solve(wl=1064e-9)
            lap(f"Interferometry(rf, E=Jf).two_lens_solve   [on device: {it.on_device}]")
            it.interferogram(bin_scale=1)
            lap("Interferometry.interferogram(bin_scale=1)")
            _ = it.rf
            lap("reading Interferometry.rf / .rE (host copies, only when asked for)")
            del it
        print(f"  {'solve + every diagnostic above':<58s} {1e3 * (time.perf_counter() - t_flow):8.1f} ms")
        del rf, Jf, sh, sc
        dom.clear_memory()
        del dom
