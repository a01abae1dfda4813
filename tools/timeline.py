"""One step of a bench run as a timeline: rocprofv3 --kernel-trace CSV -> per kernel launch its start (ms after the step's first
kernel), duration, queue, the idle gap before it on its queue and how much of it ran beside a kernel of another queue.
    python tools/timeline.py <..._kernel_trace.csv> [first kernel of a step = k_keys_band] [which step, default: the last]"""
import csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
first = sys.argv[2] if len(sys.argv) > 2 else "k_keys_band("
which = int(sys.argv[3]) if len(sys.argv) > 3 else -1
name = lambda r: r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if name(r).startswith(first)]
if not starts:
    sys.exit("no kernel named " + first)
lo = starts[which]
hi = starts[which + 1] if which != -1 and which + 1 < len(starts) else len(rows)
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
last_end, busy = {}, []
tot = {}
for r in step:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    beside = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, q2 in busy if q2 != q)
    busy.append((s, e, q))
    n = name(r).split("(")[0]
    tot[n] = tot.get(n, 0.0) + (e - s) / 1e6
    print(f"{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:10.1f} us  q{q:<3} gap {gap:8.1f} us  beside {beside / 1e3:9.1f} us  {name(r)[:70]}")
end = max(int(r["End_Timestamp"]) for r in step)
print(f"step: {(end - t0) / 1e6:.3f} ms from the first kernel's start to the last one's end; kernels by name (ms):")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"   {v:9.3f}  {n}")
# union of busy intervals = time some kernel was running
iv = sorted((s, e) for s, e, _ in busy)
covered, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        covered += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
covered += cur_e - cur_s
print(f"some kernel running: {covered / 1e6:.3f} ms; nothing running: {(end - t0 - covered) / 1e6:.3f} ms")
