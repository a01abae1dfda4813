#!/bin/bash
# usage (GPU box): bash tools/r05_rec.sh -> (a) the tile kernel with ready-made records by LDS-DMA (SYNTHRAY_TILE_RECORDS=1) against the
# producers' wavefront: every GPU test with it, then C3 A/B
T=r05f
SYNTHRAY_TILE_RECORDS=1 timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest_rec.log 2>&1; rc=$?; echo pytest with records rc $rc; tail -3 gpurun_out/${T}_pytest_rec.log
[ $rc -ne 0 ] && exit $rc
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for rep in 1 2; do
  for recs in 0 1; do
    SYNTHRAY_TILE_RECORDS=$recs timeout -k 10 200 $B > gpurun_out/${T}_rec_${recs}_$rep.json 2> gpurun_out/${T}_rec_${recs}_$rep.err || exit 1
    python - $recs $rep <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r05f_rec_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("records", sys.argv[1], "pass", sys.argv[2], "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
  done
done
SYNTHRAY_TILE_RECORDS=1 timeout -k 10 300 python bench.py --cpu-sample 2e5 --other-steps 0 --api-flow-reps 0 --steps 5 --warmup 2 > gpurun_out/${T}_rec_check.json 2> gpurun_out/${T}_rec_check.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05f_rec_check.json").read().strip().splitlines()[-1])
print("check with records:", json.dumps(d.get("check"))[:1500])
PY
