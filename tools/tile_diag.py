"""Where the tile path's lost rays come from (C3).  Needs a DIAGNOSTIC build of the library: copy trace_tile.inc / trace.hip, add the
counters described in DESIGN.md section 8 ("Where the lost rays come from": atomicAdd on sr_rays.counters[9, 12..15] after the tile's
placement, a line on stderr in sr_rays_trace_stats), `bash tools/build_variant.sh diag`, restore the sources (the build id of the shipped
library must not change), then  SYNTHRAY_LIB=$PWD/ab/libsynthray_diag.so python tools/tile_diag.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from synthpy_amd import engine

engine.init(0)
ne, x = bench.make_volume(512)
s0 = bench.make_rays(10 ** 7, 5e-3, 0)
vol = engine.Volume.from_ne(ne, x, x, x, 1064e-9, "z", phaseshift=True)
rays = engine.RayBundle(10 ** 7).upload(s0)
st = rays.trace(vol, engine.default_t_end(5e-3), 5e-3, precision="f64")
print("tile segments", rays.tile_segments, "rays through k_trace_f64 (summed over the segments)", st.fallback_rays)
