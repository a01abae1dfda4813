#!/bin/bash
# The multi-GPU legs on ONE 8-GPU node (the driver runs them; a 1-GPU box can only do the N = 1 lines):
#   bash tools/scale_check.sh [max_gpus=8] [steps=5]
# c3 weak + strong at N = 1, 2, 4, 8; c4 (1e8 rays over 8 GPUs, all three diagnostics, RCCL reduce) at N = 8;
# c5 (1024^3 domain_fft volume in slabs, RCCL hand-off) at N = 8.  Each line is bench.py's one JSON line; N > 1 lines carry
# check.multi_gpu (reduced image == sum of the ranks' deposits == one GPU doing every rank's rays).
# `python bench.py --gpus N` spawns its own ranks (no torchrun needed); 127.0.0.1 rendezvous.
set -u
MAXG=${1:-8}; STEPS=${2:-5}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
out=gpurun_out/scale; mkdir -p $out
run() {  # name, args...
  local name=$1; shift
  echo "== $name: python bench.py $*"
  timeout -k 10 1500 python bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name FAILED (rc $?)"; tail -5 $out/$name.err; return; }
  python - $out/$name.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("check") or {}
print("   n_gpus", d["n_gpus"], d["scaling"], "value %.4g %s" % (d["value"], d["unit"]), "rays/s %.4g" % d.get("rays_per_s", 0), "ms/step %.2f" % d["ms_per_step"])
t = d.get("timing") or {}
print("   ranks_seen", (c.get("multi_gpu") or {}).get("ranks_seen", (d.get("config") or {}).get("ranks_seen", 1)),
      "| per-rank ms for the steps", t.get("per_rank_ms_for_the_steps"), "| per-rank ms in the image reduce", t.get("per_rank_ms_in_the_image_reduce"))
print("   multi_gpu:", c.get("multi_gpu"))
PY
}
for n in 1 2 4 8; do
  [ $n -le $MAXG ] || continue
  run c3_weak_$n --gpus $n --steps $STEPS --warmup 1 --workload c3 --scaling weak
  run c3_strong_$n --gpus $n --steps $STEPS --warmup 1 --workload c3 --scaling strong
done
if [ $MAXG -ge 8 ]; then
  run c4_8 --gpus 8 --steps $STEPS --warmup 1 --workload c4
  run c5_8 --gpus 8 --steps 2 --warmup 1 --workload c5
fi
python - <<'PY'
import glob, json, os
rows, runs = {}, []
for f in sorted(glob.glob("gpurun_out/scale/*.json")):
    name = os.path.basename(f)[:-5]
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        runs.append({"name": name, "rc": 1, "error": str(e)})
        continue
    runs.append({"name": name, "n": d["n_gpus"], "scaling": d["scaling"], "cmd": "python bench.py --gpus %d ..." % d["n_gpus"], "rc": 0, "parsed": d})
    if name.startswith("c3_") and d.get("value"):
        rows[(d["scaling"], d["n_gpus"])] = d["value"]
eff = {}
for sc in ("weak", "strong"):
    base = rows.get((sc, 1))
    for n in (1, 2, 4, 8):
        if (sc, n) in rows and base:
            eff[f"{sc}_{n}"] = rows[(sc, n)] / (n * base)
            print(f"c3 {sc:6s} N={n}: {rows[(sc, n)]:.4g} ray-steps/s, efficiency {eff[f'{sc}_{n}']:.3f}")
# what ONE GPU said a rank's share costs (tools/share_curve.sh -> profiles/*_share_curve.json): the strong-scaling line to hold against
pred = None
for f in sorted(glob.glob("profiles/*_share_curve.json"))[-1:]:
    pred = json.load(open(f))["predicted_strong_scaling_efficiency"]
    for n in ("2", "4", "8"):
        m = eff.get(f"strong_{n}")
        print(f"c3 strong N={n}: predicted from the one-GPU share curve {pred[n]:.3f}" + (f", measured {m:.3f}" if m else ", not measured here"))
# the same schema as the driver's per-N records: one entry per run with bench.py's parsed line
json.dump({"where": "tools/scale_check.sh on this host", "runs": runs, "efficiency": eff, "predicted_strong_scaling_efficiency_from_one_gpu": pred},
          open("profiles/SCALE_local.json", "w"), indent=1)
print("wrote profiles/SCALE_local.json")
PY
