#!/bin/bash
# usage (GPU box): bash tools/profile_aux.sh -> gpurun_out/prof_r04_aux/{kernel_stats.csv,pmc.csv}: tools/aux_rate.py (1e6 rays x 256^3, kappa / Faraday / both)
# under rocprofv3: kernel-trace stats, then two PMC passes (the program directly after `--`, counters in runs of their own)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/prof_r04_aux; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/stats -o s --output-format csv -- python3 $R/tools/aux_rate.py > $out/stats.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv $out/stats/*kernel_stats.csv 2>/dev/null | head -1) $out/kernel_stats.csv
n=0
for grp in \
  "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  n=$((n+1)); d=$out/g$n
  timeout -k 10 300 rocprofv3 --pmc $grp -d $d -o p --output-format csv -- python3 $R/tools/aux_rate.py > $d.log 2>&1 || { echo "pass failed: $grp"; tail -3 $d.log; }
done
python3 $R/tools/summarise_pmc.py $out > /dev/null
rm -rf $out/stats $out/g*/*agent_info.csv
head -8 $out/kernel_stats.csv | cut -c1-170
grep "k_trace_tile<true, true>" $out/pmc.csv
