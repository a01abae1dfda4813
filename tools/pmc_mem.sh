# usage (GPU box): bash tools/pmc_mem.sh <tag> <kernel name part> <bench args...>  -> SQ wait/busy + L2/HBM counters of one launch
cd /tmp && export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pm_$TAG; rm -rf $out; mkdir -p $out
n=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  n=$((n+1))
  timeout -k 10 120 rocprofv3 --pmc $grp -d $out/g$n -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 "$@" > $out/g$n.log 2>&1 || { echo "pass failed: $grp"; tail -2 $out/g$n.log; }
done
python3 $R/tools/summarise_pmc.py $out $KERN | sed "s/(.*//" | awk -v t=$TAG '{print t": "$0}'
