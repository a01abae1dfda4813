#!/bin/bash
# usage (GPU box): bash tools/density_crossover.sh -> gpurun_out/r04_density_crossover.txt : tile path against per-ray kernel on 512^3 at falling ray densities
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/r04_density_crossover.txt; : > $out
q="--steps 5 --warmup 1 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --precision f64"
for rays in 8e6 4e6 2e6 1e6 5e5; do
  for tile in 1 0; do
    echo -n "rays $rays (per lateral cell of the beam: $(python3 -c "import math; print(round($rays / (math.pi * (4e-3 / (1e-2 / 511)) ** 2), 1))"))  SYNTHRAY_F64_TILE=$tile: " >> $out
    SYNTHRAY_F64_TILE=$tile python3 bench.py $q --rays $rays | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  kernel_ms %.3f  ray-steps/s %.4g  fallback %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['fallback_rays']))" >> $out 2>&1
  done
done
cat $out
