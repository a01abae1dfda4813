# usage: bash tools/pmc.sh <tag> [SYNTHRAY_LIB path]   -> gpurun_out/pmc_<tag>.txt (k_trace_mixed rows)
set -e
tag=$1; lib=${2:-}
cd /tmp && export TMPDIR=/tmp
[ -n "$lib" ] && export SYNTHRAY_LIB=$GRAFT_REPO_ROOT/$lib
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  d=$out/$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 280 rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $d.log 2>&1 || { echo "pmc pass failed: $grp"; tail -3 $d.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_mixed" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
w = tot.get("SQ_WAVES", 0)
if w: print("VALU per wave-step", tot["SQ_INSTS_VALU"] / w / 511, "VMEM_RD per wave-step", tot["SQ_INSTS_VMEM_RD"] / w / 511, "SALU", tot["SQ_INSTS_SALU"] / w / 511, "LDS", tot["SQ_INSTS_LDS"] / w / 511)
PY
