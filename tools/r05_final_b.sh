#!/bin/bash
# final GPU pass 3: every bench workload with this build's kernel model, the share curve, the optional terms' rate, the reference's
# flow through the mirror classes, c5 at the pipeline's chunk size, the job driver with the reference's 5e5-ray chunks merged / unmerged
bash tools/r05_workloads.sh r05
bash tools/share_curve.sh 20 > gpurun_out/r05_share_curve.log 2>&1; tail -7 gpurun_out/r05_share_curve.log
timeout -k 10 300 python tools/aux_rate.py > gpurun_out/r05_aux_rate.txt 2>&1; cat gpurun_out/r05_aux_rate.txt
timeout -k 10 300 python tools/pcie_dbg.py 2>/dev/null > gpurun_out/r05_pcie_pageable.txt; cat gpurun_out/r05_pcie_pageable.txt
timeout -k 10 300 python tools/legacy_flow_rate.py > gpurun_out/r05_legacy_flow.txt 2>&1; tail -12 gpurun_out/r05_legacy_flow.txt
timeout -k 10 500 python bench.py --workload c5 --chunk 5373952 --rays 21495808 --steps 2 --warmup 1 --cpu-sample 1e5 --api-flow-reps 0 > gpurun_out/r05_bench_c5_plan_chunk.json 2> gpurun_out/r05_bench_c5_plan_chunk.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_c5_plan_chunk.json").read().strip().splitlines()[-1])
print("c5 at the pipeline's chunk:", "%.3e" % d["value"], "%.1f ms/step" % d["ms_per_step"], d["config"]["kernel"][:90], "| pipeline", d["config"]["pipeline"], "| check", json.dumps(d["check"])[:300])
PY
for m in 0 4194304; do
  timeout -k 10 300 python -m synthpy_amd.run_trace -d 512 -r 1e7 --chunk 5e5 --merge-rays $m --diagnostics interf --ray-workers 0 --device-beam -o /tmp/o_$m.npz > gpurun_out/r05_driver_merge_$m.txt 2>&1; tail -2 gpurun_out/r05_driver_merge_$m.txt
done
