# usage (GPU box): bash tools/pmc_icache.sh <tag> <kernel name part> <bench args...> -> instruction-cache counters of one launch
cd /tmp && export TMPDIR=/tmp
TAG=$1; KERN=$2; shift 2
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pi_$TAG; rm -rf $out; mkdir -p $out
timeout -k 10 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/g1 -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 "$@" > $out/g1.log 2>&1 || { echo "pass failed"; tail -2 $out/g1.log; }
python3 $R/tools/summarise_pmc.py $out $KERN | sed "s/(.*//" | awk -v t=$TAG '{print t": "$0}'
