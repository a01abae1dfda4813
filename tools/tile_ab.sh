#!/bin/bash
# A/B of the float64 tile path (trace_tile.inc) on BASELINE config 3, all on ONE box (devices differ by several per cent):
# the per-ray kernel, then tile geometries "tb,tc,halo,band,planes per segment".  One line per variant in gpurun_out/tile_ab.txt.
# LIBS="name ..." adds library variants ab/libsynthray_<name>.so at the default geometry.
out=gpurun_out/tile_ab.txt
mkdir -p gpurun_out
: > $out
run() {
  python bench.py --steps ${STEPS:-4} --warmup 1 --other-steps 0 --cpu-sample 0 "${@:2}" > gpurun_out/tile_ab_last.json 2> gpurun_out/tile_ab_last.err || { echo "$1: FAILED" >> $out; tail -3 gpurun_out/tile_ab_last.err >> $out; return; }
  python - "$1" >> $out <<'PY'
import json, sys
d = json.load(open("gpurun_out/tile_ab_last.json"))
print(sys.argv[1], "ms/step %.2f" % d["ms_per_step"], "kernel_ms %.2f" % d["roofline"].get("kernel_ms", float("nan")), "fallback", d["config"]["fallback_rays"], "value %.3e" % d["value"])
PY
}
SYNTHRAY_F64_TILE=0 run "per-ray" "$@"
for l in $LIBS; do
  SYNTHRAY_LIB=$GRAFT_REPO_ROOT/ab/libsynthray_$l.so SYNTHRAY_F64_TILE=1 run "lib $l" "$@"
done
for lg in $LIBGEOMS; do  # "name:geometry": a library variant at a geometry of its own
  SYNTHRAY_LIB=$GRAFT_REPO_ROOT/ab/libsynthray_${lg%%:*}.so SYNTHRAY_F64_TILE=1 SYNTHRAY_TILE=${lg#*:} run "lib $lg" "$@"
done
for g in ${GEOMS:-"12,16,4,4,128" "12,16,4,4,171" "12,16,4,4,256"}; do
  SYNTHRAY_F64_TILE=1 SYNTHRAY_TILE=$g run "tile $g" "$@"
done
cat $out
