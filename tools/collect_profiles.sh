#!/bin/bash
# usage (build container, after tools/make_profiles.sh a, b, c ran on the GPU box): [RND=r05] bash tools/collect_profiles.sh
# gpurun_out/prof_<RND>_* -> profiles/<RND>_{kernel_stats,pmc}_<tag>.csv and profiles/kernel_model.json, stamped with the id of
# the library in the tree (bench.py prints a roofline fraction only for that build).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
RND=${RND:-r05}
BID=$(python3 -c "from synthpy_amd import _ffi; print(_ffi.lib.sr_version().decode().split('src:')[-1])")
rm -f profiles/kernel_model.json
#     tag            kernel        workload key            ray-steps per trace   launches per trace
while read tag kern wkey steps per; do
  d=gpurun_out/prof_${RND}_$tag
  [ -d $d ] || { echo "missing $d"; continue; }
  python3 tools/summarise_pmc.py $d $kern --model $wkey $steps $BID --per-trace $per > /dev/null
  cp $d/kernel_stats.csv profiles/${RND}_kernel_stats_$tag.csv
  (echo "# build $BID; one trace per pass: rocprofv3 --pmc <group> -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --other-steps 0 ... (tools/profile_pass.sh $tag)"; cat $d/pmc.csv) > profiles/${RND}_pmc_$tag.csv
done <<'TAB'
c3_f64            k_trace_tile  512_10000000_phase   5110000000   4
c3_f64_producers  k_trace_tile  512_10000000_phase   5110000000   3
c3_f64_per_ray    k_trace_f64   512_10000000_phase   5110000000   1
c3_mixed          k_trace_mx    512_10000000_phase   5110000000   1
c2                k_trace_mx    256_1000000_nophase  255000000    1
c4                k_trace_tile  512_12500000_phase   6387500000   4
c5                k_trace_tile  c5_1024_20000000     20460000000  8
TAB
python3 -c "
import json; m=json.load(open('profiles/kernel_model.json')); print('build', m['build_id'])
for k,v in m['kernels'].items():
    for w,e in v.items(): print(k, w, 'VALU/wave-step', e['valu_instructions_per_wave_step'], 'busy', round(e['valu_busy'],3), 'lanes', round(e['lane_utilisation'] or 0,3), 'wait', round(e['wait_any_frac_of_wave_cycles'] or 0,3), 'ms', round(e['kernel_ms_profiled'],2), 'HBM GB', round((e['hbm_bytes_per_launch'] or 0)/1e9,2))
"
