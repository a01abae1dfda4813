set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_f64; rm -rf $out; mkdir -p $out
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  d=$out/$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 280 rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --precision f64 > $d.log 2>&1 || echo "pass failed"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_planes" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(tot): print(k, tot[k])
w = tot.get("SQ_WAVES", 0)
if w: print("VALU per wave-step", tot["SQ_INSTS_VALU"] / w / 511, "VMEM_RD", tot["SQ_INSTS_VMEM_RD"] / w / 511, "SALU", tot["SQ_INSTS_SALU"] / w / 511, "LDS", tot["SQ_INSTS_LDS"] / w / 511)
PY
