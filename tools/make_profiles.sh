#!/bin/bash
# usage (GPU box): [RND=r05] bash tools/make_profiles.sh <set>
#   a: C3 -- the library's choice (the records kernel), the producers' kernel, the per-ray kernel
#   b: C3 in the mixed build; C2; C4 per GPU       c: C5 (one dense chunk, 8 slabs)
# kernel-trace stats + PMC passes (tools/profile_pass.sh) -> gpurun_out/prof_<RND>_<tag>/; then, in the build container:
# bash tools/collect_profiles.sh
R=$GRAFT_REPO_ROOT; cd $R
export RND=${RND:-r05}
Q="--api-flow-reps 0"
case "$1" in
  a)
    bash tools/profile_pass.sh c3_f64 --precision f64 $Q > gpurun_out/mp_c3_f64.log 2>&1                              # the tile path (the library's choice)
    SYNTHRAY_TILE_RECORDS=0 bash tools/profile_pass.sh c3_f64_producers --precision f64 $Q > gpurun_out/mp_c3_f64_producers.log 2>&1
    SYNTHRAY_F64_TILE=0 bash tools/profile_pass.sh c3_f64_per_ray --precision f64 $Q > gpurun_out/mp_c3_f64_per_ray.log 2>&1
    ;;
  b)
    bash tools/profile_pass.sh c3_mixed --precision mixed $Q > gpurun_out/mp_c3_mixed.log 2>&1
    STATS_STEPS=40 STATS_WARMUP=5 bash tools/profile_pass.sh c2 --workload c2 $Q > gpurun_out/mp_c2.log 2>&1
    bash tools/profile_pass.sh c4 --workload c4 $Q > gpurun_out/mp_c4.log 2>&1
    ;;
  c)
    STATS_STEPS=2 STATS_WARMUP=1 bash tools/profile_pass.sh c5 --workload c5 > gpurun_out/mp_c5.log 2>&1
    ;;
esac
echo collected $1
