#!/bin/bash
# usage (GPU box): bash tools/r05_check.sh <tag> -> the GPU tests, C3 as the driver runs it, the per-rank share curve, one step as a timeline
T=${1:-r05j}
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -3 gpurun_out/${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > gpurun_out/${T}_bench_c3.json 2> gpurun_out/${T}_bench_c3.err; echo bench rc $?
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for n in 1e7 5e6 2.5e6 1.25e6; do
  timeout -k 10 200 $B --rays $n > gpurun_out/${T}_share_$n.json 2> gpurun_out/${T}_share_$n.err || exit 1
done
python - $T <<'PY'
import json, sys
T = sys.argv[1]
d = json.loads(open(f"gpurun_out/{T}_bench_c3.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("c3", "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms", r.get("kernel_ms"), "api_flow", (d.get("api_flow") or {}).get("ms"), "check", json.dumps(d.get("check"))[:400])
t = {}
for n in ("1e7", "5e6", "2.5e6", "1.25e6"):
    d = json.loads(open(f"gpurun_out/{T}_share_{n}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    t[n] = d["ms_per_step"]
    print(n, "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms", r.get("kernel_ms"), d["config"].get("kernel"))
print("predicted strong-scaling efficiency t(1e7) / (N t(1e7 / N)): N=2 %.3f  N=4 %.3f  N=8 %.3f" % (t["1e7"] / (2 * t["5e6"]), t["1e7"] / (4 * t["2.5e6"]), t["1e7"] / (8 * t["1.25e6"])))
PY
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && out=$R/gpurun_out/${T}_tl && rm -rf $out && mkdir -p $out &&
  timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 > $out.log 2>&1;
  f=$(ls $out/*/*kernel_trace.csv $out/*kernel_trace.csv 2>/dev/null | head -1); python3 $R/tools/timeline.py $f > $R/gpurun_out/${T}_timeline.txt 2>&1; tail -24 $R/gpurun_out/${T}_timeline.txt; rm -rf $out )
