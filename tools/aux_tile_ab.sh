#!/bin/bash
# usage (GPU box): bash tools/aux_tile_ab.sh -> gpurun_out/r04_aux_tile_ab.txt : geometry of the tile path with the optional terms (1e6 rays x 256^3)
R=$GRAFT_REPO_ROOT; cd $R
out=gpurun_out/r04_aux_tile_ab.txt; : > $out
for geom in default 6,7,2,2,128 6,7,2,3,128 6,7,2,4,128 6,7,2,3,255 6,7,2,3,86 7,7,2,3,128 6,8,2,3,128 8,8,2,3,128 5,6,2,3,128; do
  echo "SYNTHRAY_TILE=$geom" >> $out
  if [ $geom = default ]; then python3 tools/aux_rate.py 2>&1 | grep "^both\|^kappa" >> $out; else SYNTHRAY_TILE=$geom python3 tools/aux_rate.py 2>&1 | grep "^both" >> $out; fi
done
cat $out
