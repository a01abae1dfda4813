#!/bin/bash
# round 5, second GPU call: the write-tracking guard's GPU tests, one C3 step as a timeline, the per-rank share curve
T=r05b
timeout -k 10 400 python -m pytest tests/test_api_flow.py tests/test_abi_and_host.py -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -5 gpurun_out/${T}_pytest.log
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && out=$R/gpurun_out/${T}_tl && rm -rf $out && mkdir -p $out &&
  timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 > $out.log 2>&1;
  f=$(ls $out/*/*kernel_trace.csv $out/*kernel_trace.csv 2>/dev/null | head -1); python3 $R/tools/timeline.py $f > $R/gpurun_out/${T}_timeline.txt 2>&1; tail -25 $R/gpurun_out/${T}_timeline.txt; rm -rf $out )
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for n in 1e7 5e6 2.5e6 1.25e6; do
  timeout -k 10 200 $B --rays $n > gpurun_out/${T}_share_$n.json 2> gpurun_out/${T}_share_$n.err || exit 1
done
python - <<'PY'
import json
for n in ("1e7", "5e6", "2.5e6", "1.25e6"):
    d = json.loads(open(f"gpurun_out/r05b_share_{n}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
    print(n, "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms", r.get("kernel_ms"), d["config"].get("kernel"))
PY
