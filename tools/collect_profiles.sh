# usage (build container, after tools/make_profiles.sh ran on the GPU box): bash tools/collect_profiles.sh
# gpurun_out/ -> profiles/r02_* and profiles/kernel_model.json, stamped with the id of the library in the tree
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
BID=$(python3 -c "from synthpy_amd import _ffi; print(_ffi.lib.sr_version().decode().split('src:')[-1])")
cp gpurun_out/valu_issue.json profiles/r02_valu_issue.json
cp gpurun_out/valu_classes.csv profiles/r02_valu_classes.csv
rm -f profiles/kernel_model.json
for prec in f64 mixed; do
  kern=k_trace_f64; [ $prec = mixed ] && kern=k_trace_mx
  python3 tools/summarise_pmc.py gpurun_out/prof_r02_c3_$prec $kern --model 512_10000000_phase 5110000000 $BID > /dev/null
  cp gpurun_out/prof_r02_c3_$prec/kernel_stats.csv profiles/r02_kernel_stats_c3_$prec.csv
  (echo "# build $BID; one launch per pass: rocprofv3 --pmc <group> -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 --precision $prec (tools/profile_r02.sh)"; cat gpurun_out/prof_r02_c3_$prec/pmc.csv) > profiles/r02_pmc_c3_$prec.csv
done
python3 -c "
import json; m=json.load(open('profiles/kernel_model.json')); print('build', m['build_id'])
for k,v in m['kernels'].items():
    for w,e in v.items(): print(k, w, 'VALU/wave-step', e['valu_instructions_per_wave_step'], 'busy', round(e['valu_busy'],3), 'clock', round(e['clock_ghz'],3), 'ms', round(e['kernel_ms_profiled'],2), 'HBM GB', round(e['hbm_bytes_per_launch']/1e9,2))
"
