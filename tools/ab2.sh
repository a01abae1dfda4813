# usage (GPU box): bash tools/ab2.sh <precision> <name>...   ("cur" = the in-tree library) -> ms per step / kernel ms / check
prec=$1; shift
for n in "$@"; do
  lib=ab/libsynthray_$n.so; [ "$n" = cur ] && lib=synthpy_amd/libsynthray.so
  SYNTHRAY_LIB=$lib timeout -k 10 300 python bench.py --precision $prec --steps 5 --warmup 1 --cpu-sample 20000 --other-steps 0 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n failed"; tail -5 gpurun_out/ab_$n.err; }
  python - $n <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{n}.json"))
    c = d["check"]
    print(n, "ms/step", round(d["ms_per_step"], 2), "kernel", round(d["roofline"]["kernel_ms"], 2), "dx", c["max_dx_m"], "dth", c["max_dtheta_rad"],
          "interf", c.get("interferogram_from_s0_max_dH_over_max_H"))
except Exception as e:
    print(n, "ERR", e)
PY
done
