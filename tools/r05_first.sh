#!/bin/bash
# round 5, first GPU call: parity tests on the straggler path, then C3 with the stragglers beside / behind the tile launches
T=r05a
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -3 gpurun_out/${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for mode in beside serial; do
  SYNTHRAY_STRAGGLERS=$mode timeout -k 10 200 $B > gpurun_out/${T}_c3_$mode.json 2> gpurun_out/${T}_c3_$mode.err || exit 1
done
for pr in low high; do
  SYNTHRAY_SIDE_PRIORITY=$pr timeout -k 10 200 $B > gpurun_out/${T}_c3_prio_$pr.json 2> gpurun_out/${T}_c3_prio_$pr.err || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05a_c3_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(f, "%.3e" % d["value"], "%.2f ms/step" % d["ms_per_step"], "kernel_ms", r.get("kernel_ms"))
PY
