#!/bin/bash
# usage (GPU box): bash tools/r05_bounds.sh -> what the records kernel's barrier and cell changes cost: DIAGNOSTIC builds (wrong results, timing only):
# -DSR_TILE_EXP_NOBARRIER (no barrier in the step loop), -DSR_TILE_EXP_NORELOC (a ray never changes its cell), -DSR_TILE_EXP_NOPRODUCE (no DMA and no
# mid record in the loop), bare = all three (the stage arithmetic with its record reads alone)
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 10 --warmup 2"
for n in cur noproduce bare nobarrier noreloc cur; do
  lib=ab/libsynthray_$n.so; [ "$n" = cur ] && lib=synthpy_amd/libsynthray.so
  SYNTHRAY_LIB=$lib timeout -k 10 200 $B > gpurun_out/r05_bounds_$n.json 2> gpurun_out/r05_bounds_$n.err || { echo $n failed; tail -3 gpurun_out/r05_bounds_$n.err; continue; }
  python - $n <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r05_bounds_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print(sys.argv[1], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
done
