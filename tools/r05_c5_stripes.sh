#!/bin/bash
# usage (GPU box): bash tools/r05_c5_stripes.sh -> C5 on one GPU (2e7 rays x 1024^3 in 8 slabs) in the chunks of a slab pipeline: cut by
# position (--stripe-chunks: every chunk a stripe of the beam at the job's density) at 1.25e6 / 2.5e6 / 5e6 rays per chunk, against the
# index-range chunks the plan makes today (5.37e6 rays: the smallest at the tile path's threshold density), and the job in ONE chunk
out=gpurun_out/r05_c5_stripes.txt; : > $out
run() {
  local name=$1; shift
  timeout -k 10 500 python bench.py --workload c5 --steps 2 --warmup 1 --cpu-sample 0 --api-flow-reps 0 "$@" > gpurun_out/r05_c5s_$name.json 2> gpurun_out/r05_c5s_$name.err || { echo "$name failed" >> $out; tail -3 gpurun_out/r05_c5s_$name.err >> $out; return; }
  python - $name >> $out <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/r05_c5s_{sys.argv[1]}.json").read().strip().splitlines()[-1]); c = d["config"]
print(f"{sys.argv[1]:22s} {c['pipeline']['chunks']:3d} chunks of {c['chunk']:8d} rays  {d['value']:.3e} ray-steps/s  {d['ms_per_step']:7.1f} ms/step  trace kernels {d['roofline']['kernel_ms']:6.1f} ms  {c['kernel'][:48]}")
PY
}
run one_chunk --rays 2e7
run index_5.37e6 --rays 21495808 --chunk 5373952
run stripes_5e6 --rays 2e7 --chunk 5000000 --stripe-chunks
run stripes_2.5e6 --rays 2e7 --chunk 2500000 --stripe-chunks
run stripes_1.25e6 --rays 2e7 --chunk 1250000 --stripe-chunks
cat $out
