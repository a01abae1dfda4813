# usage (GPU box): bash tools/profile_stats.sh <tag> <bench args...> -> gpurun_out/stats_<tag>.csv (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/st_$TAG; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out -o s --output-format csv -- python3 $R/bench.py --cpu-sample 0 --other-steps 0 "$@" > $out.log 2>&1 || { echo "failed: $TAG"; tail -3 $out.log; }
cp $(ls $out/*/*kernel_stats.csv $out/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/stats_$TAG.csv && head -5 $R/gpurun_out/stats_$TAG.csv | cut -c1-130
