#!/bin/bash
# smoke() as the driver runs it, the new bbox test, c5 in the pipeline's chunks (4 whole chunks: every slab trace on the tile path)
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; echo smoke rc $?; tail -4 gpurun_out/r05_smoke.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "beam_box or chunked_driver or driver_cli" > gpurun_out/r05_last_pytest.log 2>&1; echo pytest rc $?; tail -3 gpurun_out/r05_last_pytest.log
timeout -k 10 500 python bench.py --workload c5 --chunk 5373952 --rays 21495808 --steps 2 --warmup 1 --cpu-sample 1e5 --api-flow-reps 0 > gpurun_out/r05_bench_c5_plan_chunk.json 2> gpurun_out/r05_bench_c5_plan_chunk.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_bench_c5_plan_chunk.json").read().strip().splitlines()[-1])
print("c5 at the pipeline's chunk:", "%.3e" % d["value"], "%.1f ms/step" % d["ms_per_step"], d["config"]["kernel"][:100], "| pipeline", d["config"]["pipeline"], "| check", json.dumps(d["check"])[:200])
PY
