#!/bin/bash
# c5: where does the time outside the tile kernels go?
q="--workload c5 --steps 2 --warmup 1 --cpu-sample 0 --api-flow-reps 0"
run() { label=$1; shift; env "$@" timeout -k 10 400 python3 bench.py $q 2> gpurun_out/r05c5_dbg_$label.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'ms_per_step %.1f' % d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])"; grep "c5 pass" gpurun_out/r05c5_dbg_$label.err | tail -2; }
run first SYNTHRAY_BENCH_DEBUG=1
run second X=1
