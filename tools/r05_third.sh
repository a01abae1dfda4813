#!/bin/bash
# round 5, third GPU call: the last segment in two launches (A/B), timeline, tile path against per-ray kernel at low densities
T=r05c
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -3 gpurun_out/${T}_pytest.log
[ $rc -ne 0 ] && exit $rc
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for sp in 0 0.75 0.6 0.85 0.75; do
  SYNTHRAY_TILE_LAST_SPLIT=$sp timeout -k 10 200 $B > gpurun_out/${T}_c3_split_$sp.json 2> gpurun_out/${T}_c3_split_$sp.err || exit 1
  python - $sp <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r05c_c3_split_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("split", sys.argv[1], "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms", r.get("kernel_ms"))
PY
done
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && out=$R/gpurun_out/${T}_tl && rm -rf $out && mkdir -p $out &&
  timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 > $out.log 2>&1;
  f=$(ls $out/*/*kernel_trace.csv $out/*kernel_trace.csv 2>/dev/null | head -1); python3 $R/tools/timeline.py $f > $R/gpurun_out/${T}_timeline.txt 2>&1; rm -rf $out )
out=gpurun_out/${T}_density.txt; : > $out
q="--steps 10 --warmup 2 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --precision f64"
for rays in 2.5e6 1.75e6 1.25e6 1e6 7.5e5 5e5; do
  for tile in 1 0; do
    echo -n "rays $rays (per cell of the beam's box: $(python3 -c "print(round($rays / (8e-3 / (1e-2 / 511) + 1) ** 2, 2))"))  SYNTHRAY_F64_TILE=$tile: " >> $out
    SYNTHRAY_F64_TILE=$tile timeout -k 10 120 python3 bench.py $q --rays $rays | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  kernel_ms %.3f  ray-steps/s %.4g  stragglers %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['fallback_rays']))" >> $out 2>&1
  done
done
cat $out
