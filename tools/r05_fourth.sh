#!/bin/bash
# round 5, fourth GPU call: (b) the trimmed cell-change region against the selects (ab/libsynthray_select.so), then where to cut the
# last segment and which priority the side stream gets
T=r05d
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print(sys.argv[2], "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
}
for rep in 1 2; do
  for n in cur select; do
    lib=ab/libsynthray_$n.so; [ "$n" = cur ] && lib=synthpy_amd/libsynthray.so
    SYNTHRAY_LIB=$lib SYNTHRAY_TILE_LAST_SPLIT=0 timeout -k 10 200 $B > gpurun_out/${T}_reloc_${n}_$rep.json 2> gpurun_out/${T}_reloc_${n}_$rep.err || exit 1
    line gpurun_out/${T}_reloc_${n}_$rep.json "reloc $n pass $rep"
  done
done
for pr in normal high; do
  for sp in 0 0.5 0.65 0.75; do
    SYNTHRAY_SIDE_PRIORITY=$pr SYNTHRAY_TILE_LAST_SPLIT=$sp timeout -k 10 200 $B > gpurun_out/${T}_split_${pr}_$sp.json 2> gpurun_out/${T}_split_${pr}_$sp.err || exit 1
    line gpurun_out/${T}_split_${pr}_$sp.json "side priority $pr, last segment cut at $sp"
  done
done
