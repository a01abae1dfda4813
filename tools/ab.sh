set -e
for n in "$@"; do
  SYNTHRAY_LIB=ab/libsynthray_$n.so timeout -k 10 300 python bench.py --steps 5 --warmup 1 --cpu-sample 20000 > gpurun_out/ab_$n.json 2> gpurun_out/ab_$n.err || { echo "$n failed"; tail -5 gpurun_out/ab_$n.err; }
done
python - "$@" <<'PY'
import json, sys
for n in sys.argv[1:]:
    try:
        d = json.load(open(f"gpurun_out/ab_{n}.json"))
        print(n, round(d["ms_per_step"], 2), round(d["roofline"]["kernel_ms"], 2), d["check"]["max_dx_m"], d["check"]["max_dtheta_rad"])
    except Exception as e:
        print(n, "ERR", e)
PY
