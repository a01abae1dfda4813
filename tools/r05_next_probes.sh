#!/bin/bash
# usage (GPU box): bash tools/r05_next_probes.sh -> two probes for the next round, by environment only (no library change):
#  (a) the N = 8 stripe share of C3 (1.25e6 rays at the bundle's density) in 3 segments of 171 planes instead of 4 of 128: fewer launches and binnings
#      for a small bundle against more lost rays;  (b) C5 with an 8 x 8 tile (three workgroups per CU): wider margins for the deep slabs' rays
out=gpurun_out/r05_next_probes.txt; : > $out
one() {
  local name=$1; shift
  "$@" > gpurun_out/r05_probe_$name.json 2> gpurun_out/r05_probe_$name.err || { echo "$name failed" >> $out; return; }
  python - $name >> $out <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r05_probe_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} {d['value']:.3e} ray-steps/s  {d['ms_per_step']:8.3f} ms/step  trace kernels {d['roofline']['kernel_ms']:8.3f} ms  stragglers {d['config'].get('fallback_rays')}")
PY
}
S="python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps 20 --warmup 3 --rays 1e7 --share-of 8 --share-rank 4"
one share8_default $S
SYNTHRAY_TILE="8,7,2,4,171" one share8_3_segments $S
SYNTHRAY_TILE="8,7,2,4,256" one share8_2_segments $S
SYNTHRAY_TILE="8,7,2,3,128" one share8_3_rows $S
C="python bench.py --workload c5 --steps 2 --warmup 1 --cpu-sample 0 --api-flow-reps 0"
one c5_default $C
SYNTHRAY_TILE="8,8,2,4,128" one c5_tile_8x8 $C
SYNTHRAY_TILE="8,7,2,3,128" one c5_3_rows $C
cat $out
