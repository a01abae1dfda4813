# usage: bash tools/driver_gaps.sh  -- kernel time vs wall of the chunked driver (device beam), from a kernel trace
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/driver_trace; rm -rf $out; mkdir -p $out
cd $R
timeout -k 10 280 rocprofv3 --kernel-trace -d $out -o d --output-format csv -- python3 -m synthpy_amd.run_trace -d 512 -r 5e7 --device-beam --diagnostics shadow,schlieren,interf -o $out/o.npz > $out/run.log 2>&1
tail -2 $out/run.log
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the chunk loop: from the first k_beam to the end
i0 = next(i for i, r in enumerate(rows) if "k_beam" in r[2])
rows = rows[i0:]
busy = sum(e - s for s, e, _ in rows)
wall = rows[-1][1] - rows[0][0]
by = collections.Counter()
import re
short = lambda n: (re.search(r"(k_\w+|__amd\w+)", n) or re.search(r"(\w+)", n)).group(1)
for s, e, n in rows: by[short(n)] += e - s
print(f"kernels {busy/1e6:.1f} ms of {wall/1e6:.1f} ms wall ({100*busy/wall:.0f}% busy), {len(rows)} launches")
for n, t in by.most_common(8): print(f"  {t/1e6:8.1f} ms  {n}")
gaps = sorted(((rows[i+1][0] - rows[i][1]) / 1e3, short(rows[i][2]), short(rows[i+1][2])) for i in range(len(rows) - 1))
big = collections.Counter()
for g, a, b in gaps:
    if g > 20: big[(a, b)] += g
print("largest idle gaps by (kernel before, kernel after), ms total:")
for (a, b), t in big.most_common(8): print(f"  {t/1e3:8.1f}  {a} -> {b}")
PY
