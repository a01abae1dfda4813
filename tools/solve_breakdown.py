"""Where ScalarDomain.solve()'s wall-clock goes on BASELINE-sized inputs: upload of s0 (pageable NumPy), binning + trace, download of
rf (+ Jf) into page-locked arrays, and the diagnostics' histogram() pieces.   python tools/solve_breakdown.py [c3|c2]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
grid, N, phase = {"c3": (512, 10 ** 7, True), "c2": (256, 10 ** 6, False)}[wl]
engine.init(0)
ne, x = bench.make_volume(grid)
s0 = bench.make_rays(N, 5e-3, 0)
vol = engine.Volume.from_ne(ne, x, x, x, 1064e-9, "z", phaseshift=phase)
rays = engine.RayBundle(N)
t_end = engine.default_t_end(5e-3)
print(f"# {wl}: {N} rays x {grid}^3, {engine._ffi.lib.sr_version().decode()}")
for rep in range(4):
    t = [time.perf_counter()]
    def lap():
        engine.synchronize()
        t.append(time.perf_counter())
        return 1e3 * (t[-1] - t[-2])
    rays.upload(s0); up = lap()
    st = rays.trace(vol, t_end, 5e-3, precision="f64", resident=False); tr = lap()
    _, rf, Jf = rays.download(sf=False, Jf=phase); dn = lap()
    img = engine.DetectorImage.counts(bin_scale=1); ic = lap()
    rays.deposit(img, engine.chain_shadow_two(), want_stats=False, exact_counts=False); dp = lap()
    H = img.counts_f64(); hd = lap()
    img.close(); cl = lap()
    print(f"pass {rep}: upload {up:.2f}  trace {tr:.2f} (kernels {st.trace_kernel_ms:.2f}, stream total {st.total_ms:.2f})  download rf{'+Jf' if phase else ''} {dn:.2f}  "
          f"image create {ic:.2f}  deposit {dp:.2f}  counts_f64 {hd:.2f}  image close {cl:.2f}   [GB/s up {s0.nbytes / up / 1e6:.1f}, down {(rf.nbytes + (Jf.nbytes if phase else 0)) / dn / 1e6:.1f}, H {H.nbytes / hd / 1e6:.1f}]")
    del rf, Jf, H
