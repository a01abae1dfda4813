# usage: bash tools/profile_c3.sh [mixed|f64] [kernel name part]
# kernel-trace stats + HBM traffic PMC of the default bench (C3) -> gpurun_out/prof_c3_<precision>/
set -e
cd /tmp && export TMPDIR=/tmp
PREC=${1:-mixed}; KERN=${2:-k_trace_mixed}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_c3_$PREC; rm -rf $out; mkdir -p $out
timeout -k 10 280 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-sample 0 --precision $PREC > $out/stats.log 2>&1
for c in FETCH_SIZE "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  d=$out/$(echo $c | tr ' ' '_')
  timeout -k 10 280 rocprofv3 --pmc $c -d $d -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --precision $PREC > $d.log 2>&1 || echo "pass failed: $c"
done
python3 - $out $KERN <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
cp $(ls $out/stats/*/*kernel_stats.csv $out/stats/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv
head -4 $out/kernel_stats.csv | cut -c1-160
