#!/bin/bash
# c5 on one GPU: one segment per 128-plane slab (the default) against two (the first segment's stragglers run beside the second)
q="--workload c5 --steps 2 --warmup 1 --cpu-sample 0 --api-flow-reps 0"
run() { label=$1; shift; env "$@" timeout -k 10 400 python3 bench.py $q 2> gpurun_out/r05c5_segs.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', 'ms_per_step %.1f' % d['ms_per_step'], 'kernel_ms %.1f' % d['roofline']['kernel_ms'])"; }
run "default (one segment per slab)" X=1
run "two segments per slab, 4 rows" SYNTHRAY_TILE=8,7,2,4,64
run "two segments per slab, 5 rows" SYNTHRAY_TILE=8,7,2,5,64
run "two segments, equal cuts" SYNTHRAY_TILE=8,7,2,4,64 SYNTHRAY_TILE_CUTS=1,1
run "one segment, 5 rows" SYNTHRAY_TILE=8,7,2,5,128
