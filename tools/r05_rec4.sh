#!/bin/bash
# usage (GPU box): bash tools/r05_rec4.sh -> the ready-made-records kernel at FOUR wavefronts per SIMD (ab/libsynthray_rec4.so: 128
# registers, 8 x 7 tile = 36 KB of LDS, four workgroups per CU) against today's kernel
T=r05g
B="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3"
run() { # name lib records tile
  SYNTHRAY_LIB=$2 SYNTHRAY_TILE_RECORDS=$3 SYNTHRAY_TILE=$4 timeout -k 10 200 $B > gpurun_out/${T}_$1.json 2> gpurun_out/${T}_$1.err || { echo "$1 failed"; tail -3 gpurun_out/${T}_$1.err; return; }
  python - $1 <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r05g_{sys.argv[1]}.json").read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print(sys.argv[1], "%.3e" % d["value"], "%.3f ms/step" % d["ms_per_step"], "kernel_ms %.3f" % r.get("kernel_ms"), "stragglers", d["config"].get("fallback_rays"))
PY
}
cur=synthpy_amd/libsynthray.so; r4=ab/libsynthray_rec4.so
run producers_8x8 $cur 0 ""
run rec4_8x7 $r4 1 "8,7,2,2,171"
run rec4_8x7_band3 $r4 1 "8,7,2,3,171"
run rec4_8x7_seg128 $r4 1 "8,7,2,2,128"
run rec4_8x8 $r4 1 "8,8,2,2,171"
run rec3_8x7 $cur 1 "8,7,2,2,171"
run producers_8x7 $cur 0 "8,7,2,2,171"
run producers_8x8_again $cur 0 ""
SYNTHRAY_LIB=$r4 SYNTHRAY_TILE_RECORDS=1 SYNTHRAY_TILE="8,7,2,2,171" timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tile or trace_vs_oracle or c3_shaped or slab" > gpurun_out/${T}_pytest.log 2>&1; echo pytest rec4 rc $?; tail -2 gpurun_out/${T}_pytest.log
