# usage (GPU box): bash tools/ab_block.sh   -> kernel ms of the two headline kernels by workgroup size (SYNTHRAY_BLOCK_F64 / _MX)
for b in 64 128 256; do
  SYNTHRAY_BLOCK_F64=$b timeout -k 10 300 python bench.py --precision f64 --steps 5 --warmup 1 --cpu-sample 0 --other-steps 0 > gpurun_out/blk_f64_$b.json 2> gpurun_out/blk_f64_$b.err || echo "f64 $b failed"
  SYNTHRAY_BLOCK_MX=$b timeout -k 10 300 python bench.py --precision mixed --steps 5 --warmup 1 --cpu-sample 0 --other-steps 0 > gpurun_out/blk_mx_$b.json 2> gpurun_out/blk_mx_$b.err || echo "mx $b failed"
  python - $b <<'PY'
import json, sys
b = sys.argv[1]
for k in ("f64", "mx"):
    try:
        d = json.load(open(f"gpurun_out/blk_{k}_{b}.json"))
        print(k, b, "ms/step", round(d["ms_per_step"], 2), "kernel", round(d["roofline"]["kernel_ms"], 2))
    except Exception as e:
        print(k, b, "ERR", e)
PY
done
