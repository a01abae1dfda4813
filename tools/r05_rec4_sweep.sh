#!/bin/bash
# usage (GPU box): bash tools/r05_rec4_sweep.sh -> geometry of the four-wavefront records kernel on C3: rows per band, planes per segment, cuts
T=r05i
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
run() { # name tile cuts
  SYNTHRAY_LIB=ab/libsynthray_rec4.so SYNTHRAY_TILE_RECORDS=1 SYNTHRAY_TILE=$2 SYNTHRAY_TILE_CUTS=$3 timeout -k 10 200 $B > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.err || { echo "$1 failed"; tail -3 gpurun_out/${T}_$1.err; return; }
  python - $T $1 "$2" "$3" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("tile", sys.argv[3], "cuts", sys.argv[4] or "ramp 1.3..0.7", "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
}
run a "8,7,2,4,128" ""
run b "8,7,2,4,128" "1.2,1.07,0.93,0.8"
run c "8,7,2,4,128" "1,1,1,1"
run d "8,7,2,5,128" ""
run e "8,7,2,4,103" ""
run f "8,7,2,5,103" ""
run g "8,7,2,4,171" ""
run h "8,7,2,4,128" "1.4,1.1,0.9,0.6"
run i "8,7,2,3,128" ""
run j "8,7,2,4,128" ""
