#!/bin/bash
# usage (GPU box): tools/pmc_tile.sh <tag> <kernel name part> <bench args...>   (env passes through)
# two --pmc passes of ONE bench launch each: issue / wait / LDS counters of the named kernel, per wavefront-step
cd /tmp && export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pt_$TAG; rm -rf $out; mkdir -p $out
n=0
for grp in \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/g$n -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 "$@" > $out/g$n.log 2>&1 || { echo "pass failed: $grp"; tail -3 $out/g$n.log; }
done
python3 - $out "$KERN" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k:28s} {tot[k]:.6g}  ({cnt[k]} dispatches)")
PY
