#!/bin/bash
# usage (GPU box): bash tools/band_ab.sh -> gpurun_out/r04_band_ab.txt : rows per band of the tile path's ray order at low ray densities (512^3, 4 mm beam)
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/r04_band_ab.txt; : > $out
q="--steps 5 --warmup 1 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --precision f64"
for rays in 4e6 2e6 1e6; do
  echo -n "rays $rays per-ray kernel: " >> $out
  SYNTHRAY_F64_TILE=0 python3 bench.py $q --rays $rays | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'])" >> $out 2>&1
  for geom in 8,8,2,2,171 8,8,2,3,171 8,8,2,4,171 8,8,2,5,171 8,8,2,4,128 8,8,2,6,128; do
    echo -n "rays $rays SYNTHRAY_TILE=$geom: " >> $out
    SYNTHRAY_F64_TILE=1 SYNTHRAY_TILE=$geom python3 bench.py $q --rays $rays | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  kernel_ms %.3f  fallback %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['fallback_rays']))" >> $out 2>&1
  done
done
cat $out
