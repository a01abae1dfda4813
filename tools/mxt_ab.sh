#!/bin/bash
# A/B of the mixed tile path (trace_mxt.inc) against k_trace_mx on one box: C3 shape (--precision mixed) and C2
out=gpurun_out/mxt_ab.txt; : > $out
run() {
  timeout -k 10 200 python bench.py --steps ${STEPS:-5} --warmup 2 --other-steps 0 --cpu-sample 0 "${@:2}" > gpurun_out/mxt_ab_last.json 2> gpurun_out/mxt_ab_last.err || { echo "$1: FAILED" >> $out; tail -3 gpurun_out/mxt_ab_last.err >> $out; return; }
  python - "$1" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/mxt_ab_last.json"))
print(sys.argv[1], "ms/step %.3f" % d["ms_per_step"], d["roofline"]["kernel"], "kernel_ms %.3f" % d["roofline"].get("kernel_ms", float("nan")), "fallback", d["config"]["fallback_rays"], "guard", d["config"]["edge_guard_retraced_rays_per_step"], "value %.3e" % d["value"])
PY
}
SYNTHRAY_MX_TILE=0 run "c3 mixed per-ray" --precision mixed
for g in ${GEOMS:-"12,16,4,4,256" "12,16,4,4,171" "12,16,4,4,128"}; do
  SYNTHRAY_MX_TILE=1 SYNTHRAY_TILE=$g run "c3 mixed tile $g" --precision mixed
done
SYNTHRAY_MX_TILE=0 STEPS=20 run "c2 per-ray" --workload c2
SYNTHRAY_MX_TILE=1 STEPS=20 run "c2 tile" --workload c2
SYNTHRAY_MX_TILE=1 SYNTHRAY_TILE=12,16,4,4,128 STEPS=20 run "c2 tile 128" --workload c2
cat $out
