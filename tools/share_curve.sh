#!/bin/bash
# usage (ONE-GPU box): bash tools/share_curve.sh [steps=20] -> gpurun_out/share_curve.{json,txt}
# BASELINE's metric is "1e7 rays x 512^3 on 1/2/4/8 GPUs": at N GPUs each rank traces 1e7 / N rays.  What one GPU needs for such a
# share -- ms per step, of which in the trace kernels, the kernel that ran -- and from it the strong-scaling efficiency to hold the
# first real N > 1 line against:  t(1e7) / (N * t(1e7 / N))  (the image reduce, ~35 MB of counts or 284 MB of field sums over xGMI
# once per job, is not in it).  Copied to profiles/<round>_share_curve.* by hand; tools/scale_check.sh prints it beside what it measures.
STEPS=${1:-20}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out/share
for n in 10000000 5000000 2500000 1250000; do
  timeout -k 10 300 python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps $STEPS --warmup 3 --rays $n > gpurun_out/share/$n.json 2> gpurun_out/share/$n.err || { echo "rays $n failed"; tail -3 gpurun_out/share/$n.err; exit 1; }
done
python - <<'PY'
import json
rows, t = [], {}
for n in (10000000, 5000000, 2500000, 1250000):
    d = json.loads(open(f"gpurun_out/share/{n}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    t[n] = d["ms_per_step"]
    rows.append({"rays": n, "gpus_this_share_stands_for": 10000000 // n, "ms_per_step": d["ms_per_step"], "trace_kernels_ms": r["kernel_ms"],
                 "outside_the_trace_kernels_ms": d["ms_per_step"] - r["kernel_ms"], "ray_steps_per_s": d["value"], "kernel": r["kernel"],
                 "rate_over_the_full_bundles": d["value"] / None if False else None, "stragglers": d["config"].get("fallback_rays"),
                 "library": d["config"].get("library")})
for row in rows:
    row["rate_over_the_full_bundles"] = row["ray_steps_per_s"] / rows[0]["ray_steps_per_s"]
pred = {str(N): t[10000000] / (N * t[10000000 // N]) for N in (2, 4, 8)}
out = {"workload": "C3: 512^3 turbulent n_e, phase integral + reference beam + two-lens interferogram, float64; one MI355X",
       "rows": rows, "predicted_strong_scaling_efficiency": pred,
       "formula": "t(1e7) / (N * t(1e7 / N)) from the ms_per_step column; the one image reduce per job is not in it"}
json.dump(out, open("gpurun_out/share_curve.json", "w"), indent=1)
with open("gpurun_out/share_curve.txt", "w") as f:
    f.write("# tools/share_curve.sh: the per-rank share of BASELINE's 1e7 rays x 512^3 on N GPUs, traced by ONE MI355X\n")
    f.write("# rays       stands for N   ms/step   in the trace kernels   outside them   ray-steps/s   rate / full bundle   kernel\n")
    for r in rows:
        f.write(f"{r['rays']:9d}   {r['gpus_this_share_stands_for']:6d}        {r['ms_per_step']:7.3f}   {r['trace_kernels_ms']:10.3f}             {r['outside_the_trace_kernels_ms']:6.3f}         "
                f"{r['ray_steps_per_s']:.4g}     {r['rate_over_the_full_bundles']:.3f}                {r['kernel']}\n")
    f.write("# predicted strong-scaling efficiency t(1e7) / (N t(1e7 / N)):  " + "  ".join(f"N={k}: {v:.3f}" for k, v in pred.items()) + "\n")
print(open("gpurun_out/share_curve.txt").read())
PY
