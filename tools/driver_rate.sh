# usage (GPU box): bash tools/driver_rate.sh  -> the job script with NumPy's seeded rays from the host (the reference's own
# sample), 1e7 rays x 512^3 in the reference's chunks of 5e5: chunks drawn one after the other against drawn ahead by workers
for w in 0 8; do
  timeout -k 10 500 python -m synthpy_amd.run_trace -d 512 -r 1e7 --chunk 5e5 --diagnostics shadow,schlieren,interf --ray-workers $w -o /tmp/drv_$w.npz 2> gpurun_out/driver_rate_$w.err | sed "s/^/ray-workers $w: /"
done
python - <<'PY'
import numpy as np
a, b = np.load("/tmp/drv_0.npz"), np.load("/tmp/drv_8.npz")
print("shadow equal:", bool(np.array_equal(a["shadow"], b["shadow"])), "schlieren equal:", bool(np.array_equal(a["schlieren"], b["schlieren"])),
      "interferogram max diff / max:", float(np.max(np.abs(a["interf"] - b["interf"])) / np.max(np.abs(a["interf"]))))
PY
