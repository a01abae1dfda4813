"""Job of many small chunks (the reference's 5e5-ray chunks, pvti_trace_mpi.py:27) on one or two streams against one dense
chunk: rays/s of the whole job, bundle drawn on the GPU.   python tools/chunk_rate.py [total rays] [precision]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench
from synthpy_amd import engine as eng
from synthpy_amd import run_trace as rt

eng.init(0)
total = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
prec = sys.argv[2] if len(sys.argv) > 2 else "mixed"
ne, x = bench.make_volume(512)
ext, lwl = 5e-3, 1064e-9
vol = eng.Volume.from_ne(ne, x, x, x, lwl, "z", phaseshift=prec == "f64")
beam = dict(beam_size=4e-3, divergence=5e-5, ne_extent=ext, beam_type="circular", probing_direction="z", seed=0)
names = ["shadow"] if prec != "f64" else ["interf"]
ref = None
for chunk, streams in ((total, 1), (500_000, 1), (500_000, 2), (2_000_000, 1), (2_000_000, 2)):
    best = None
    for rep in range(3):
        diags = rt.standard_diagnostics(names, lwl, 1)
        t = rt.chunked_trace(vol, ext, total, None, diags, chunk=chunk, precision=prec, device_beam=beam, streams=streams)
        best = t if best is None or t["seconds"] < best["seconds"] else best
        H = diags[0].result()
    if ref is None:
        ref = H
    same = bool(np.array_equal(H, ref)) if prec != "f64" else float(np.max(np.abs(H - ref)) / np.max(ref))
    print(f"{prec} chunk {chunk:>9d} streams {streams}: {best['seconds'] * 1e3:8.2f} ms  {total / best['seconds']:.3e} rays/s  "
          f"{best['ray_steps'] / best['seconds']:.3e} ray-steps/s  image vs one chunk: {same}")
