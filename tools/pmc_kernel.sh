# usage: bash tools/pmc_kernel.sh <kernel name part> <tag> "<counters group 1>" ["<group 2>" ...]  -> prints sums for that kernel
set -e
kern=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmck_$tag; rm -rf $out; mkdir -p $out
n=0
for grp in "$@"; do
  n=$((n+1)); d=$out/g$n
  timeout -k 10 280 rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $d.log 2>&1 || { echo "pass failed: $grp"; tail -3 $d.log; }
done
python3 - $out "$kern" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
PY
