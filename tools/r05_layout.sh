#!/bin/bash
# usage (GPU box): bash tools/r05_layout.sh -> the records kernel's LDS layout: records 128 bytes apart in plain order (ab/libsynthray_linear.so)
# against piece-major column blocks 1104 bytes apart (the tree's library); tile tests with the new layout
T=r05m
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
for rep in 1 2; do
  for n in vaddr cur; do
    lib=ab/libsynthray_$n.so; [ "$n" = cur ] && lib=synthpy_amd/libsynthray.so
    SYNTHRAY_LIB=$lib timeout -k 10 200 $B > gpurun_out/${T}_${n}_$rep.json 2> gpurun_out/${T}_${n}_$rep.err || { echo $n failed; tail -3 gpurun_out/${T}_${n}_$rep.err; exit 1; }
    python - $T $n $rep <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/{sys.argv[1]}_{sys.argv[2]}_{sys.argv[3]}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print("layout", sys.argv[2], "pass", sys.argv[3], "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
  done
done
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or trace_vs or c3_shaped or slab or interferometry" > gpurun_out/${T}_pytest.log 2>&1; echo pytest rc $?; tail -2 gpurun_out/${T}_pytest.log
