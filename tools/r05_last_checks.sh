#!/bin/bash
# the round's last GPU pass: smoke() as the driver runs it, every GPU test, bench.py exactly as the driver types it (N = 1), and the driver's own
# torchrun form at two ranks on the one GPU (tools/rehearse_torchrun.sh)
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo smoke rc $?; tail -4 gpurun_out/r05_smoke.log
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r05_last_pytest.log 2>&1; rc=$?; echo pytest rc $rc; tail -3 gpurun_out/r05_last_pytest.log
[ $rc -ne 0 ] && exit $rc
t0=$(date +%s); timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_as_the_driver.json 2> gpurun_out/r05_bench_as_the_driver.err; echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
lines = [l for l in open("gpurun_out/r05_bench_as_the_driver.json").read().splitlines() if l.strip()]
d = json.loads(lines[-1])
print(len(lines), "line(s) on stdout;", "%.4g" % d["value"], d["unit"], "%.2f ms/step" % d["ms_per_step"], "kernel_ms %.2f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"],
      "cpu_baseline", d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], "| check", json.dumps(d["check"])[:260])
PY
bash tools/rehearse_torchrun.sh > gpurun_out/r05_rehearse_torchrun.txt 2>&1; tail -6 gpurun_out/r05_rehearse_torchrun.txt
