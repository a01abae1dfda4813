#!/bin/bash
# usage: tools/kstats.sh <tag> <bench args...>   (env passes through: SYNTHRAY_F64_TILE, SYNTHRAY_TILE, ...)
# rocprofv3 --kernel-trace --stats of one bench run -> gpurun_out/kstats_<tag>.csv (+ the first rows printed)
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/kstats_$TAG; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out -o s --output-format csv -- python3 $R/bench.py --steps ${STEPS:-3} --warmup 1 --cpu-sample 0 --other-steps 0 "$@" > $out.log 2>&1
cp $(ls $out/*/*kernel_stats.csv $out/*kernel_stats.csv 2>/dev/null | head -1) $R/gpurun_out/kstats_$TAG.csv
python3 - $R/gpurun_out/kstats_$TAG.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"].replace("void (anonymous namespace)::", "")[:60].ljust(60), r["Calls"].rjust(5), "avg %10.1f us" % (float(r["AverageNs"]) / 1e3), "%5s %%" % r["Percentage"])
PY
