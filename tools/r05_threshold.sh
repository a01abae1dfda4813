#!/bin/bash
# usage (GPU box): bash tools/r05_threshold.sh -> the tile path (records kernel, the library's geometry) against the per-ray kernel between 7 and 15 rays per
# lateral cell of the beam's box on 512^3: where sr_tile_min_density belongs
out=gpurun_out/r05_threshold.txt; : > $out
q="--steps 10 --warmup 2 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --precision f64"
for rays in 1.25e6 1.5e6 1.75e6 2.0e6 2.25e6 2.5e6; do
  for tile in 1 0; do
    echo -n "rays $rays (per cell of the beam's box: $(python3 -c "print(round($rays / (8e-3 / (1e-2 / 511) + 1) ** 2, 2))"))  SYNTHRAY_F64_TILE=$tile: " >> $out
    SYNTHRAY_F64_TILE=$tile timeout -k 10 120 python3 bench.py $q --rays $rays | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  kernel_ms %.3f  ray-steps/s %.4g  stragglers %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['fallback_rays']))" >> $out 2>&1
  done
done
cat $out
