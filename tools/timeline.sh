#!/bin/bash
# usage (GPU box): [FIRST='k_keys('] bash tools/timeline.sh <tag> [bench args...] -> gpurun_out/<tag>_timeline.txt : one step of bench.py as a timeline
# (tools/timeline.py; FIRST = the kernel a step starts with: k_keys_band( on the tile path, k_keys( otherwise)
T=${1:-tl}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && out=$R/gpurun_out/${T}_tl && rm -rf $out && mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 "$@" > $out.log 2>&1
f=$(ls $out/*/*kernel_trace.csv $out/*kernel_trace.csv 2>/dev/null | head -1); python3 $R/tools/timeline.py $f "${FIRST:-k_keys_band(}" > $R/gpurun_out/${T}_timeline.txt 2>&1; tail -24 $R/gpurun_out/${T}_timeline.txt; rm -rf $out
