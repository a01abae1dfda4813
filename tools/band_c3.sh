#!/bin/bash
# usage (GPU box): bash tools/band_c3.sh -> gpurun_out/r04_band_c3.txt : rows per band / planes per segment on the headline (C3)
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/r04_band_c3.txt; : > $out
q="--steps 8 --warmup 2 --cpu-sample 0 --other-steps 0 --api-flow-reps 0"
for geom in 8,8,2,2,171 8,8,2,3,171 8,8,2,4,171 8,8,2,3,128 8,8,2,2,171; do
  echo -n "C3 SYNTHRAY_TILE=$geom: " >> $out
  SYNTHRAY_TILE=$geom python3 bench.py $q | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  kernel_ms %.3f  fallback %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['fallback_rays']))" >> $out 2>&1
done
cat $out
