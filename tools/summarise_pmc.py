#!/usr/bin/env python3
"""Sum the rocprofv3 --pmc passes of tools/profile_r02.sh per kernel and counter.

    python tools/summarise_pmc.py gpurun_out/prof_r02_<tag> [kernel-name-part]            -> table on stdout + <dir>/pmc.csv
    python tools/summarise_pmc.py gpurun_out/prof_r02_<tag> <kernel> --model <workload-key> <ray-steps-per-launch> <build-id>
        also merges the kernel's VALU mix and HBM bytes into profiles/kernel_model.json (what bench.py's roofline reads).
"""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).strip()


def load(d):
    tot, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(d + "/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def main():
    d = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else None
    tot, cnt = load(d)
    with open(d + "/pmc.csv", "w") as o:
        o.write("kernel,counter,value_summed_over_dispatches,dispatches\n")
        for k in sorted(tot):
            o.write(f'"{k[0]}",{k[1]},{tot[k]:.8g},{cnt[k]}\n')
    if not want:
        return
    ks = sorted({k for k, _ in tot if want in k})
    for kern in ks:
        c = {cn: tot[(kn, cn)] / max(1, cnt[(kn, cn)]) * (1 if cn != "SQ_INSTS_VALU" and cn != "GRBM_GUI_ACTIVE" else 1)
             for (kn, cn) in tot if kn == kern}
        print(kern)
        for cn in sorted(c):
            print(f"   {cn:28s} {c[cn]:.6g}   ({cnt[(kern, cn)]} dispatches averaged)")
    if "--model" in sys.argv:
        i = sys.argv.index("--model")
        wkey, steps, build = sys.argv[i + 1], float(sys.argv[i + 2]), sys.argv[i + 3]
        # --per-trace D: the kernel runs D times per trace (the tile path's segments); the entry describes the D launches together
        per_trace = int(sys.argv[sys.argv.index("--per-trace") + 1]) if "--per-trace" in sys.argv else 1
        kern = ks[0]
        c = {cn: tot[(kn, cn)] / max(1, cnt[(kn, cn)]) * per_trace for (kn, cn) in tot if kn == kern}
        classes = {"FMA_F64": "SQ_INSTS_VALU_FMA_F64", "ADD_F64": "SQ_INSTS_VALU_ADD_F64", "MUL_F64": "SQ_INSTS_VALU_MUL_F64",
                   "TRANS_F64": "SQ_INSTS_VALU_TRANS_F64", "FMA_F32": "SQ_INSTS_VALU_FMA_F32", "ADD_F32": "SQ_INSTS_VALU_ADD_F32",
                   "MUL_F32": "SQ_INSTS_VALU_MUL_F32", "TRANS_F32": "SQ_INSTS_VALU_TRANS_F32", "CVT": "SQ_INSTS_VALU_CVT",
                   "INT32": "SQ_INSTS_VALU_INT32", "INT64": "SQ_INSTS_VALU_INT64"}
        mix = {k: c.get(v, 0.0) for k, v in classes.items()}
        mix["OTHER"] = c["SQ_INSTS_VALU"] - sum(mix.values())  # moves, selects, compares, min/max: no class counter
        wave_steps = steps / 64.0
        # the time of the profiled launch itself, from the stats pass of the same script
        kms = None
        for r in csv.DictReader(open(d + "/kernel_stats.csv")):
            if short(r["Name"]) == kern:
                kms = float(r["AverageNs"]) * 1e-6 * per_trace
        ent = {"valu_per_launch": mix, "valu_per_wave_step": {k: round(v / wave_steps, 2) for k, v in mix.items()},
               "valu_instructions_per_wave_step": round(c["SQ_INSTS_VALU"] / wave_steps, 1),
               "salu_per_wave_step": round(c.get("SQ_INSTS_SALU", 0) / wave_steps, 1),
               "vmem_rd_per_wave_step": round(c.get("SQ_INSTS_VMEM_RD", 0) / wave_steps, 2),
               "ray_steps_per_launch": steps, "kernel_ms_profiled": kms, "launches_per_trace": per_trace,
               "valu_busy": (c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)) if "GRBM_GUI_ACTIVE" in c else None,
               "clock_ghz": (c["GRBM_GUI_ACTIVE"] / 8 / (kms * 1e-3) / 1e9) if kms and "GRBM_GUI_ACTIVE" in c else None,
               "wait_any_frac_of_wave_cycles": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if "SQ_WAVE_CYCLES" in c else None,
               # active lanes per issued VALU instruction: SQ_THREAD_CYCLES_VALU counts lane-cycles (4 per lane and instruction
               # slot), SQ_ACTIVE_INST_VALU the 4-cycle slots
               "lane_utilisation": (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
                                    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU") else None),
               "hw_issue_cycles_per_launch": sum(mix[k] * w for k, w in {"FMA_F64": 4, "ADD_F64": 4, "MUL_F64": 4, "TRANS_F64": 16,
                                                                           "FMA_F32": 4, "ADD_F32": 4, "MUL_F32": 4, "TRANS_F32": 8, "CVT": 4,
                                                                           "INT32": 2, "INT64": 4, "OTHER": 4}.items()),
               "hbm_bytes_per_launch": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
               "hbm_note": "FETCH_SIZE (KB) doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section), WRITE_SIZE as read; separate --pmc passes",
               "source": os.path.relpath(d, ROOT)}
        path = os.path.join(ROOT, "profiles", "kernel_model.json")
        model = json.load(open(path)) if os.path.exists(path) else {}
        if model.get("build_id") != build:
            model = {"build_id": build, "kernels": {}}
        full = next((short(r["Name"]) for r in csv.DictReader(open(d + "/kernel_stats.csv")) if short(r["Name"]) == kern), kern)
        model["kernels"].setdefault(full, {})[wkey] = ent
        json.dump(model, open(path, "w"), indent=1)
        print(json.dumps(ent, indent=1))


if __name__ == "__main__":
    main()
