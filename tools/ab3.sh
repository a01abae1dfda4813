# usage (GPU box): bash tools/ab3.sh <label> <bench args...>   -> one line: ms/step, kernel ms of headline + other build
lab=$1; shift
timeout -k 10 300 python bench.py --cpu-sample 0 "$@" > gpurun_out/ab3_$lab.json 2> gpurun_out/ab3_$lab.err || { echo "$lab failed"; tail -3 gpurun_out/ab3_$lab.err; }
python - $lab <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab3_{sys.argv[1]}.json")); o = d.get("other_build") or {}
print(sys.argv[1], d["config"]["precision"], "kernel %.3f ms" % d["roofline"]["kernel_ms"], "value %.4g" % d["value"], "|", o.get("precision"), "kernel", o.get("kernel_ms"), "value %.4g" % o.get("value", 0))
PY
