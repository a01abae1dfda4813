# usage (ONE-GPU box): bash tools/rehearse_multi.sh  -> the N > 1 job of bench.py with every rank on device 0 (--rehearse-shared-gpu):
# c3 strong at 2 ranks with the oracle check, weak at 3 and at 6 ranks (the box allows at most 6 processes on its GPU: an
# 8-rank rehearsal cannot run there), c4 (all three diagnostics) at 4 ranks, c5 (slab pipeline) at 2 and 3 ranks, and at 2 with its chunks cut by position; small
# grids / ray counts where the full ones would not fit the time; prints check.multi_gpu of each
ONLY=${ONLY:-}   # ONLY="strong2 c5_2_stripes": just these legs
run() {
  name=$1; shift
  if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
  timeout -k 10 500 python bench.py --rehearse-shared-gpu "$@" > gpurun_out/reh_$name.json 2> gpurun_out/reh_$name.err
  echo "$name rc=$?"; tail -2 gpurun_out/reh_$name.err
  python - gpurun_out/reh_$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d["config"]
print(d["n_gpus"], d["scaling"], "rays this rank", c.get("rays_this_gpu"), "all", c.get("rays_all_gpus"), "ranks_seen", c.get("ranks_seen"), "ms/step %.1f" % d["ms_per_step"],
      "| multi_gpu:", (d.get("check") or {}).get("multi_gpu"))
PY
}
run strong2 --gpus 2 --steps 2 --warmup 1 --scaling strong --cpu-sample 20000 --other-steps 1
run weak2_full --gpus 2 --steps 2 --warmup 1 --scaling weak --cpu-sample 0 --other-steps 0
run weak3 --gpus 3 --steps 2 --warmup 1 --scaling weak --rays 2e6 --cpu-sample 0 --other-steps 0
run weak6 --gpus 6 --steps 1 --warmup 1 --scaling weak --grid 128 --rays 2e5 --cpu-sample 0 --other-steps 0
run c4_4 --gpus 4 --workload c4 --steps 1 --warmup 1 --grid 256 --rays 5e5 --cpu-sample 0 --other-steps 0
run c5_2 --gpus 2 --workload c5 --grid 256 --rays 2e6 --chunk 5e5 --steps 1 --warmup 1
run c5_3 --gpus 3 --workload c5 --grid 192 --rays 1e6 --chunk 2.5e5 --steps 1 --warmup 0
run c5_2_stripes --gpus 2 --workload c5 --grid 256 --rays 2e6 --stripe-chunks --steps 1 --warmup 1   # chunks cut by position (plan_chunks(cut='stripe'))
