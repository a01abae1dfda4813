#!/bin/bash
# usage (GPU box): bash tools/make_profiles_r04.sh <set>     set = a: C3 in its three builds;  b: C2, C4 per GPU;  c: C5 (one dense chunk, 8 slabs)
# kernel-trace stats + PMC passes (tools/profile_r04.sh) -> gpurun_out/prof_r04_<tag>/; then, in the build container:
# bash tools/collect_profiles_r04.sh
R=$GRAFT_REPO_ROOT; cd $R
Q="--api-flow-reps 0"
case "$1" in
  a)
    bash tools/profile_r04.sh c3_f64 --precision f64 $Q > gpurun_out/mp4_c3_f64.log 2>&1                              # the tile path (the library's choice)
    SYNTHRAY_F64_TILE=0 bash tools/profile_r04.sh c3_f64_per_ray --precision f64 $Q > gpurun_out/mp4_c3_f64_per_ray.log 2>&1
    bash tools/profile_r04.sh c3_mixed --precision mixed $Q > gpurun_out/mp4_c3_mixed.log 2>&1
    ;;
  b)
    STATS_STEPS=40 STATS_WARMUP=5 bash tools/profile_r04.sh c2 --workload c2 $Q > gpurun_out/mp4_c2.log 2>&1
    bash tools/profile_r04.sh c4 --workload c4 $Q > gpurun_out/mp4_c4.log 2>&1
    ;;
  c)
    STATS_STEPS=2 STATS_WARMUP=1 bash tools/profile_r04.sh c5 --workload c5 > gpurun_out/mp4_c5.log 2>&1
    ;;
esac
echo collected $1
