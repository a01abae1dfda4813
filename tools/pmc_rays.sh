# usage: bash tools/pmc_rays.sh RAYS   -- HBM bytes fetched by the trace kernel for a bundle of RAYS rays (C3 volume)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_rays_$1; rm -rf $out; mkdir -p $out
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE -d $out -o p --output-format csv -- python3 $R/bench.py --rays $1 --steps 1 --warmup 0 --cpu-sample 0 > $out/run.log 2>&1
python3 - $out $1 <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace_mixed" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            tot += float(r["Counter_Value"]); n += 1
# FETCH_SIZE is in KB; gfx950 reports half the bytes of 16-byte-per-lane loads (MI355X_MICROARCH.md, HBM section): doubled
print(f"rays {sys.argv[2]}: {n} launch(es), FETCH_SIZE {tot:.0f} KB -> {2 * tot * 1024 / max(n, 1) / 1e9:.2f} GB per launch (doubled)")
PY
