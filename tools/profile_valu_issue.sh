# usage: bash tools/profile_valu_issue.sh   -> profiles-ready: gpurun_out/valu_issue.json (cycles per instruction) and
# gpurun_out/valu_classes.csv (which SQ_INSTS_VALU_* class counts each instruction: the calibration of the kernel model)
set -e
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $R/tools/valu_issue.hip -o /tmp/valu_issue
/tmp/valu_issue > $R/gpurun_out/valu_issue.json
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/prof_valu_issue; rm -rf $out; mkdir -p $out
n=0
for grp in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  n=$((n+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $out/g$n -o p --output-format csv -- /tmp/valu_issue > $out/g$n.log 2>&1 || echo "pass failed: $grp"
done
python3 - $out $R/gpurun_out/valu_classes.csv <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"])
kernels = sorted({k for k, _ in tot}); counters = sorted({c for _, c in tot})
with open(sys.argv[2], "w") as o:
    o.write("kernel," + ",".join(counters) + "\n")
    for k in kernels:
        base = tot.get((k, "SQ_INSTS_VALU"), 0) or 1
        o.write(k + "," + ",".join(f"{tot.get((k, c), 0) / base:.4f}" if c.startswith("SQ_INSTS_VALU") and c != "SQ_INSTS_VALU" else f"{tot.get((k, c), 0):.6g}" for c in counters) + "\n")
print(open(sys.argv[2]).read())
PY
