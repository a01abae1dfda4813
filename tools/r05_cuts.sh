#!/bin/bash
# usage (GPU box): bash tools/r05_cuts.sh -> where the tile path cuts the volume into segments (SYNTHRAY_TILE_CUTS = the segments' shares
# of the node planes): the LAST segment's stragglers have no tile launch to run beside, the earlier ones' do
T=r05e
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for cuts in "1,1,1" "1.2,1,0.8" "1.3,1,0.7" "1.4,1.05,0.55" "1.5,1.1,0.4" "1,1,0.6,0.4" "1,1,1"; do
  SYNTHRAY_TILE_CUTS=$cuts timeout -k 10 200 $B > gpurun_out/${T}_cuts.json 2> gpurun_out/${T}_cuts.err || exit 1
  python - "$cuts" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r05e_cuts.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("cuts", sys.argv[1], "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
done
