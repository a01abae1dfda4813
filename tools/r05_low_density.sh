#!/bin/bash
# usage (GPU box): bash tools/r05_low_density.sh -> the records kernel's geometry at 5e6 / 2.5e6 / 1.25e6 rays on 512^3 (30 / 15 / 7.4 rays per cell
# of the beam's box): rows per band, planes per segment; against the producers' kernel and the per-ray kernel
T=r05k
out=gpurun_out/${T}_low_density.txt; : > $out
q="--steps 10 --warmup 2 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --precision f64"
run() { # rays label env...
  rays=$1; label=$2; shift 2
  echo -n "rays $rays  $label: " >> $out
  env "$@" timeout -k 10 120 python3 bench.py $q --rays $rays | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f  kernel_ms %.3f  ray-steps/s %.4g  stragglers %d  %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['config']['fallback_rays'], d['roofline']['kernel']))" >> $out 2>&1
}
for rays in 5e6 2.5e6; do
  run $rays "library's choice" X=1
  run $rays "producers' kernel" SYNTHRAY_TILE_RECORDS=0
  run $rays "per-ray kernel" SYNTHRAY_F64_TILE=0
  for g in "8,7,2,3,128" "8,7,2,4,128" "8,7,2,5,128" "8,7,2,6,128" "8,7,2,4,103" "8,7,2,5,103" "8,7,2,6,103" "8,7,2,5,86" "8,8,2,5,128"; do
    run $rays "records $g" SYNTHRAY_F64_TILE=1 SYNTHRAY_TILE=$g
  done
done
for rays in 1.25e6; do
  run $rays "library's choice" X=1
  run $rays "producers' kernel forced" SYNTHRAY_TILE_RECORDS=0 SYNTHRAY_F64_TILE=1
  for g in "8,7,2,6,128" "8,7,2,6,103" "8,7,2,6,86" "8,7,2,8,103" "8,8,2,6,103"; do
    run $rays "records $g" SYNTHRAY_F64_TILE=1 SYNTHRAY_TILE=$g
  done
done
cat $out
