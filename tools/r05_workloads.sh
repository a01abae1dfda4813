#!/bin/bash
# usage (GPU box): bash tools/r05_workloads.sh <tag> -> every bench workload on one GPU -> gpurun_out/<tag>_bench_*.json
T=${1:-r05w}
timeout -k 10 300 python bench.py > gpurun_out/${T}_bench_c3.json 2> gpurun_out/${T}_bench_c3.err
timeout -k 10 200 python bench.py --workload c2 --steps 20 > gpurun_out/${T}_bench_c2.json 2> gpurun_out/${T}_bench_c2.err
timeout -k 10 300 python bench.py --workload c4 > gpurun_out/${T}_bench_c4.json 2> gpurun_out/${T}_bench_c4.err
timeout -k 10 600 python bench.py --workload c5 --steps 3 --warmup 1 > gpurun_out/${T}_bench_c5.json 2> gpurun_out/${T}_bench_c5.err
python - $T <<'PY'
import json, sys
for w in ("c3", "c2", "c4", "c5"):
    try:
        d = json.loads(open(f"gpurun_out/{sys.argv[1]}_bench_{w}.json").read().strip().splitlines()[-1]); r = d["roofline"] or {}
        print(w, "%.3e" % d["value"], "%.2f ms" % d["ms_per_step"], r.get("kernel"), r.get("kernel_ms"), "frac", r.get("frac"),
              r.get("model") if isinstance(r.get("model"), str) else "model ok", "| check", json.dumps(d.get("check"))[:600])
    except Exception as e:
        print(w, "ERR", e)
PY
