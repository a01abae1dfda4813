#!/bin/bash
# usage (GPU box): bash tools/ab_rotate.sh   -> gpurun_out/r04_ab_rotate.txt : C3 with the producers' wavefront rotating / fixed, twice each
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/r04_ab_rotate.txt; : > $out
q="--steps 10 --warmup 2 --cpu-sample 0 --other-steps 0 --api-flow-reps 0"
for rep in 1 2; do
  for rot in 1 0; do
    echo "SYNTHRAY_TILE_ROTATE=$rot pass $rep" >> $out
    SYNTHRAY_TILE_ROTATE=$rot python3 bench.py $q | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ms_per_step %.2f  kernel_ms %.2f  value %.4g  fallback %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['fallback_rays']))" >> $out 2>&1
  done
done
cat $out
