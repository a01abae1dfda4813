# usage (GPU box): bash tools/pmc_quick.sh <tag> <kernel name part> <bench args...>   (env passes through)
# two --pmc passes of ONE bench launch each: VALU/wait counters and the vector-memory pipe; prints per-wave-step figures
cd /tmp && export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pq_$TAG; rm -rf $out; mkdir -p $out
n=0
for grp in \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum TCP_GATE_EN2_sum GRBM_GUI_ACTIVE"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $out/g$n -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 "$@" > $out/g$n.log 2>&1 || { echo "pass failed: $grp"; tail -3 $out/g$n.log; }
done
python3 $R/tools/summarise_pmc.py $out $KERN | awk -v t=$TAG '{print t": "$0}'
