#!/bin/bash
# usage (ONE-GPU box): bash tools/share_curve.sh [steps=20] -> gpurun_out/share_curve.{json,txt}
# BASELINE's metric is "1e7 rays x 512^3 on 1/2/4/8 GPUs": at N GPUs each rank traces 1e7 / N rays of ONE seeded bundle (bench.py --scaling
# strong).  What one GPU needs for such a share -- ms per step, of which in the trace kernels, the kernel that ran -- under the two cuts
# bench.py knows (--shard): "stripe" (the default: equal-count stripes of the beam along x; the edge stripe, rank 0, and a middle one,
# rank N/2) and "index" (contiguous index ranges, the reference's cut: the whole beam at 1/N of the density), and from it the
# strong-scaling efficiency to hold the first real N > 1 line against:  t(1e7) / (N * the slowest share's t)  (the image reduce, ~35 MB of
# counts or 284 MB of field sums over xGMI once per job, is not in it).  Copied to profiles/<round>_share_curve.* by hand;
# tools/scale_check.sh prints it beside what it measures.
STEPS=${1:-20}
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
mkdir -p gpurun_out/share; rm -f gpurun_out/share/*.json gpurun_out/share/*.err
run() {  # name, bench args
  local name=$1; shift
  timeout -k 10 300 python bench.py --cpu-sample 0 --other-steps 0 --api-flow-reps 0 --steps $STEPS --warmup 3 --rays 1e7 "$@" > gpurun_out/share/$name.json 2> gpurun_out/share/$name.err || { echo "$name failed"; tail -3 gpurun_out/share/$name.err; exit 1; }
}
run full
for n in 2 4 8; do
  run stripe_${n}_edge --share-of $n --share-rank 0 --shard stripe
  run stripe_${n}_mid --share-of $n --share-rank $((n / 2)) --shard stripe
  run index_${n} --share-of $n --share-rank 0 --shard index
done
python - <<'PY'
import json
def line(name):
    d = json.loads(open(f"gpurun_out/share/{name}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    return {"share": name, "rays": d["config"]["rays_this_gpu"], "ms_per_step": d["ms_per_step"], "trace_kernels_ms": r["kernel_ms"],
            "outside_the_trace_kernels_ms": d["ms_per_step"] - r["kernel_ms"], "ray_steps_per_s": d["value"], "kernel": r["kernel"],
            "stragglers": d["config"].get("fallback_rays"), "library": d["config"].get("library")}
full = line("full")
rows = [dict(full, gpus_this_share_stands_for=1, cut="-")]
pred = {"stripe": {}, "index": {}}
for n in (2, 4, 8):
    for cut, names in (("stripe", (f"stripe_{n}_edge", f"stripe_{n}_mid")), ("index", (f"index_{n}",))):
        ls = [dict(line(nm), gpus_this_share_stands_for=n, cut=cut) for nm in names]
        rows += ls
        pred[cut][str(n)] = full["ms_per_step"] / (n * max(l["ms_per_step"] for l in ls))
for r in rows:
    r["rate_over_the_full_bundles"] = r["ray_steps_per_s"] / full["ray_steps_per_s"]
out = {"workload": "C3: 512^3 turbulent n_e, phase integral + reference beam + two-lens interferogram, float64; one MI355X",
       "rows": rows, "predicted_strong_scaling_efficiency": pred["stripe"], "predicted_strong_scaling_efficiency_index_cut": pred["index"],
       "formula": "t(1e7) / (N * the slowest share's t) from the ms_per_step column; the one image reduce per job is not in it"}
json.dump(out, open("gpurun_out/share_curve.json", "w"), indent=1)
with open("gpurun_out/share_curve.txt", "w") as f:
    f.write("# tools/share_curve.sh: the per-rank share of BASELINE's 1e7 rays x 512^3 on N GPUs (bench.py --scaling strong), traced by ONE MI355X\n")
    f.write("# share            rays       stands for N   ms/step   in the trace kernels   outside them   ray-steps/s   rate / full bundle   kernel\n")
    for r in rows:
        f.write(f"{r['share']:16s} {r['rays']:9d}   {r['gpus_this_share_stands_for']:6d}        {r['ms_per_step']:7.3f}   {r['trace_kernels_ms']:10.3f}             {r['outside_the_trace_kernels_ms']:6.3f}         "
                f"{r['ray_steps_per_s']:.4g}     {r['rate_over_the_full_bundles']:.3f}                {r['kernel']}\n")
    for cut in ("stripe", "index"):
        f.write(f"# predicted strong-scaling efficiency, shares cut by {cut}:  " + "  ".join(f"N={k}: {v:.3f}" for k, v in pred[cut].items()) + "\n")
print(open("gpurun_out/share_curve.txt").read())
PY
