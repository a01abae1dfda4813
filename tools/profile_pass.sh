# usage (GPU box): [RND=r05] bash tools/profile_pass.sh <tag> <bench args...>        e.g.  bash tools/profile_pass.sh c3_f64 --precision f64
# rocprofv3 kernel-trace stats + PMC passes of ONE bench launch each (program directly after `--`, counters in their own
# runs, never together with a trace domain) -> gpurun_out/prof_<RND>_<tag>/{kernel_stats.csv,pmc.csv,g*/}
cd /tmp && export TMPDIR=/tmp
RND=${RND:-r05}
TAG=$1; shift
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_${RND}_$TAG; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $R/bench.py --steps ${STATS_STEPS:-4} --warmup ${STATS_WARMUP:-1} --cpu-sample 0 --other-steps 0 "$@" > $out/stats.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv $out/stats/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv
n=0
for grp in \
  "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64" \
  "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
  "SQ_THREAD_CYCLES_VALU" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  n=$((n+1)); d=$out/g$n
  timeout -k 10 400 rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 "$@" > $d.log 2>&1 || { echo "pass failed: $grp"; tail -3 $d.log; }
done
python3 $R/tools/summarise_pmc.py $out > /dev/null
rm -rf $out/stats $out/g*/*agent_info.csv
head -3 $out/kernel_stats.csv | cut -c1-160
