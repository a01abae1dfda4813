# usage (GPU box): bash tools/profile_waits.sh   -> what the wavefronts of the headline kernel wait for: extra PMC passes (LDS, barrier-like and
# issue counters), one group per rocprofv3 run (program directly after `--`), C3 f64 tile path, one step; raw CSVs under gpurun_out/prof_waits/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_waits; rm -rf $out; mkdir -p $out
rocprofv3 --list-avail > $out/avail.txt 2>&1 || true
n=0
for grp in \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_EXP_GDS SQ_ACTIVE_INST_FLAT SQ_WAVE_CYCLES" \
  "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_INSTS_WAVE32 SQ_WAVE_CYCLES" \
  "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" \
  "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES" \
  "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES"; do
  n=$((n+1)); d=$out/g$n
  timeout -k 10 300 rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --precision f64 > $d.log 2>&1 || { echo "pass $n failed: $grp"; tail -2 $d.log; }
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_trace_tile" in row.get("Kernel_Name", ""):
            tot[(f.split("/prof_waits/")[1].split("/")[0], row["Counter_Name"])] += float(row["Counter_Value"])
for (g, c), v in sorted(tot.items()):
    print(f"{g} {c} {v:.6g}")
PY
rm -rf $out/g*/*/*agent_info.csv
