# usage (GPU box): bash tools/pmc_quick.sh <tag> <kernel name part> <bench args...>   (env passes through)
# one --pmc pass of ONE bench launch: VALU / wait counters of the named kernel
cd /tmp && export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pq_$TAG; rm -rf $out; mkdir -p $out
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $out/g1 -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 "$@" > $out/g1.log 2>&1 || { echo "pass failed"; tail -3 $out/g1.log; }
python3 $R/tools/summarise_pmc.py $out $KERN | sed "s/(.*//" | awk -v t=$TAG '{print t": "$0}'
