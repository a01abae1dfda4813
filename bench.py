#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json: ray-steps/s and rays/s to detector,
1e7 rays through a 512^3 turbulent n_e volume, phase integral + interferogram; config C3).

    python bench.py [--gpus N --steps K --warmup W]

With N > 1 and no torchrun environment the command spawns its own N ranks (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* set, 127.0.0.1 rendezvous) BEFORE anything touches a GPU and relays rank 0's JSON line; under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` it is one of the ranks.

One "step" = one pass of the path over one batch of rays already resident in HBM:
    bin rays by entry cell -> plane-stepping RK4 trace (+ time-stepping fallback) -> reference beam +
    two-lens optics + complex detector deposit (accumulating into the job's image)
and the timed job = K steps + ONE RCCL sum of the per-GPU images when N > 1 (the reference's drivers sum their
chunks' images locally and reduce once, examples/jobs/run_scripts/pvti_trace_mpi.py:144-170).

--scaling weak (default): every rank traces its own seeded bundle of --rays rays (as the reference's MPI drivers
give each rank its own bundle); --scaling strong: BASELINE's 1e7 rays are ONE seeded bundle cut into one share per rank, so
the job does not depend on N -- equal-count STRIPES of the beam (--shard stripe, the default: distributed.shard_stripe; a
rank's rays keep the full bundle's density on 1/N of the area, which is what the trace's rate depends on) or contiguous
index ranges as the reference cuts them (--shard index: distributed.shard_range; every rank's share covers the whole beam
at 1/N of the density).  The volume is replicated in every GPU's HBM.
No torch anywhere: the control plane (rendezvous, barrier, max over ranks, the RCCL id hand-off) is plain TCP
(synthpy_amd/_rendezvous.py); the data path is libsynthray.so + RCCL.

Precision: "auto" (engine.resolve_precision) traces a phase-integrating volume in float64 -- the only build whose
interferogram reproduces the oracle's from the same rays -- and a volume without the phase (counts diagnostics) with
the mixed build.  The other build is timed in the same run (key "other_build"), outside the timed region.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is what a copy achieves
SIMDS, CLOCK_GHZ = 1024, 2.4  # 256 CUs x 4 SIMD-32; max shader clock (MI355X_MICROARCH.md chip table)
BYTES_PER_RAY_STEP = {False: 384, True: 512}  # 4 RHS x 8 corners x 4 B x (3 gradients [+ n])
KERNEL_MODEL = os.path.join(ROOT, "profiles", "kernel_model.json")  # tools/summarise_pmc.py writes it from rocprofv3 passes
VALU_ISSUE = os.path.join(ROOT, "profiles", "r02_valu_issue.json")  # tools/valu_issue.hip


# --------------------------------------------------------------------------------------------------------------
# launcher: no GPU call, no synthpy_amd import in here
# --------------------------------------------------------------------------------------------------------------
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c3",
                    help="BASELINE.json configs[1..4]: c2 = 1e6 rays x 256^3, shadow + schlieren; c3 = 1e7 x 512^3, interferometry "
                         "(the headline, default); c4 = 1.25e7 rays per GPU (1e8 over 8) x 512^3, all three diagnostics; "
                         "c5 = 1024^3 volume (real domain_fft field) cut into slabs of node planes, one per GPU (--slabs on one GPU when N = 1), "
                         "rays handed from slab to slab (--rays = total rays, default 1e8; 2e7 at N = 1)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --rays rays per GPU (each rank its own seeded bundle); strong: --rays rays in all, one seeded "
                         "bundle cut into one share per rank (--shard)")
    ap.add_argument("--shard", choices=["stripe", "index"], default="stripe",
                    help="--scaling strong: a rank's share = an equal-count stripe of the beam along x (full density on 1/N of the "
                         "area; default) or a contiguous index range (the reference's cut: the whole beam at 1/N of the density)")
    ap.add_argument("--share-of", type=int, default=0, metavar="N",
                    help="one GPU traces the share ONE rank of N would get under --scaling strong (tools/share_curve.sh): "
                         "--share-rank of N, cut by --shard; the line's n_gpus stays 1")
    ap.add_argument("--share-rank", type=int, default=0)
    ap.add_argument("--rays", type=float, default=None, help="rays per GPU (weak) / in all (strong); overrides the workload's")
    ap.add_argument("--grid", type=int, default=None, help="nodes per axis (overrides the workload's)")
    ap.add_argument("--beam-size", type=float, default=4e-3, help="beam radius in m (kernel diagnostics: a narrow beam keeps every line in L2)")
    ap.add_argument("--substeps", type=int, default=1)
    ap.add_argument("--precision", choices=["auto", "mixed", "f64"], default="auto",
                    help="auto: float64 when the phase is integrated (interferometry), else mixed (float64 state and "
                         "accumulation, float32 stage arithmetic); see engine.resolve_precision")
    ap.add_argument("--other-steps", type=int, default=3, help="steps of the OTHER build timed after the job (0 = skip)")
    ap.add_argument("--no-sort", action="store_true")
    ap.add_argument("--no-phase", action="store_true", help="shadowgraphy + schlieren deposit instead of the interferogram")
    ap.add_argument("--api-flow-reps", type=int, default=5,
                    help="N = 1: passes of the same workload through the reference's own API (ScalarDomain.solve -> diagnostics "
                         "classes), timed outside the job and printed as `api_flow` (0 = skip)")
    ap.add_argument("--cpu-sample", type=float, default=2e5, help="rays traced by the CPU baseline / checker (0 = skip)")
    ap.add_argument("--chunk", type=float, default=None,
                    help="c5: rays per pipeline chunk.  Default: dense chunks, so that every slab is traced by the tile kernel -- N = 1: "
                         "all --rays in ONE chunk (2e7: 38 rays per lateral cell of the beam); N > 1: 1.25e7 (24 per cell; 8 chunks of 1e8)")
    ap.add_argument("--slabs", type=int, default=8, help="c5 at N = 1: slabs held by the one GPU")
    ap.add_argument("--host-rays", action="store_true", help="c5: upload a host ray bundle per chunk instead of drawing the rays on the GPU")
    ap.add_argument("--stripe-chunks", action="store_true",
                    help="c5: the job's rays are ONE host bundle cut into STRIPES of the beam (distributed.stripe_chunks: every chunk at the "
                         "whole job's density, however small; plan_chunks(cut='stripe') sizes them unless --chunk does).  One GPU: uploaded "
                         "before the timed region -- what a slab pipeline's chunks cost when they are cut by position instead of by index; "
                         "N > 1: rank 0 uploads a chunk as it enters the pipeline")
    ap.add_argument("--dry-control-plane", action="store_true",
                    help="exercise the launcher only: spawn, TCP rendezvous, barrier, max over ranks, one JSON line; no GPU, "
                         "no library (CPU test of the N > 1 command line)")
    ap.add_argument("--spawn-timeout", type=float, default=3000.0, help="seconds the parent waits for its ranks")
    ap.add_argument("--shared-gpu-rccl", action="store_true",
                    help="test of the RCCL-failure branch on a ONE-GPU box: every rank opens device 0 and asks RCCL for the image sum, "
                         "which RCCL refuses (two ranks on one device); the job must say so, take the host sum and mark its line")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="rehearsal of the N > 1 job on a ONE-GPU box: every rank opens device 0 and the image sum goes through "
                         "the host and the control plane instead of RCCL (which refuses two ranks on one device).  Exercises the sharding, the "
                         "launcher and check.multi_gpu on real kernels; its timing means nothing and the line says so")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh ranks of this very command line and
    relay rank 0's stdout.  The parent makes no GPU call (it never imports the library): a process that has
    initialised the GPU must not be re-executed, and the children are ordinary child processes, not an exec."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + args.spawn_timeout
    rc = 0
    try:
        while True:  # a rank that dies leaves the others in a barrier: end them rather than wait for the rendezvous timeout
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                break
            if any(c not in (None, 0) for c in codes) or time.time() > deadline:
                rc = 124 if time.time() > deadline else 1
                break
            time.sleep(0.2)
    finally:
        for p in procs:  # exactly the processes started here
            if p.poll() is None:
                p.kill()
    out0 = procs[0].communicate()[0] or ""  # rank 0's one line (a few KB: the pipe never fills while we poll)
    for p in procs[1:]:
        p.wait()
    sys.stdout.write(out0)
    sys.stdout.flush()
    return rc or max(abs(p.returncode or 0) for p in procs)


_RESULT = None  # the stream the ONE JSON line goes to (see own_the_result_stream); None: sys.stdout


def own_the_result_stream():
    """The job's stdout carries ONE line.  Libraries under it write there too -- RCCL prints a five-line banner ("RCCL version : ...")
    on rank 0's C stdout when the first communicator is made -- so a rank keeps a private copy of file descriptor 1 for its
    result and points descriptor 1 itself at stderr for everybody else, native code included."""
    global _RESULT
    if _RESULT is None:
        sys.stdout.flush()
        _RESULT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(out):
    line = json.dumps(out)
    if _RESULT is None:
        print(line, flush=True)
    else:
        _RESULT.write(line + "\n")
        _RESULT.flush()


def dry_control_plane(args):
    """The launcher path with nothing behind it: rendezvous over the product's control plane (synthpy_amd._rendezvous: TCP,
    no torch), a barrier on both sides of a stand-in timed region, max over ranks, one JSON line from rank 0."""
    from synthpy_amd._rendezvous import TcpGroup

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    if os.environ.get("SYNTHRAY_BENCH_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        raise SystemExit(3)
    grp = TcpGroup(rank, world, timeout_s=120)
    grp.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    grp.barrier()
    t = grp.allreduce(time.perf_counter() - t0, "max")
    seen = grp.allreduce(1.0, "sum")
    if rank == 0:
        emit({"metric": "dry control plane (no GPU work)", "value": None, "unit": "ray-steps/s", "n_gpus": args.gpus,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": t * 1e3, "higher_is_better": True,
              "scaling": args.scaling, "vs_baseline": None, "dry": True, "ranks_seen": int(seen),
              "config": {"workload": "launcher only"}})
    grp.barrier()
    grp.close()
    return 0


# --------------------------------------------------------------------------------------------------------------
# workload pieces
# --------------------------------------------------------------------------------------------------------------
def host_cores():
    """CPU cores this process may actually use: the cgroup quota if there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def make_volume(grid, seed=1234, device=False):
    """n_e = 1e25 + 9e24*noise, noise = k^(-11/3) Gaussian random field of examples/jobs/run_scripts/turb_gen.py:36-50
    (gaussian3D.domain_fft(l_max=1, l_min=0.01, extent=5 mm, res=grid/2)), box +-5 mm, `grid` nodes per axis.
    device=True: the inverse FFT on the GPU (sr_field_ifft_real: the same seeded field to 1e-16; what a 1024^3 volume uses)."""
    import numpy as np

    from synthpy_amd.field_generator.gaussian3D import gaussian3D

    np.random.seed(seed)
    noise = gaussian3D(lambda k: k ** (-11 / 3)).domain_fft(1.0, 0.01, 5, grid // 2, 1.0, device=device)
    ne = 1e25 + 9e24 * noise * float(os.environ.get("SYNTHRAY_BENCH_NOISE", "1"))  # 0: a flat volume (kernel diagnostics)
    x = np.linspace(-5e-3, 5e-3, grid)
    return ne, x


def make_rays(n, ext, seed, beam_size=4e-3):
    """Circular beam, radius 4 mm, divergence 5e-5 rad (examples/jobs/run_scripts/test_SynthRayTrace.py:60-63)."""
    import numpy as np

    from synthpy_amd.solvers_legacy.full_solver import init_beam

    np.random.seed(seed)
    return init_beam(n, beam_size, 5e-5, ext, "circular", "z")


def kernel_name(precision, phase, substeps=1, tile_segments=0, records=False):
    """The dominant kernel of a trace as rocprofv3 names it (trace.hip's launch table).  tile_segments > 0: the float64 tile
    path ran (RayBundle.tile_segments): that many launches of k_trace_tile per trace, priced together as one unit; records: the
    records kernel (RayBundle.tile_records: coefficient records ready-made in HBM, by LDS-DMA), else the producers' kernel."""
    ph = "true" if phase else "false"
    if tile_segments > 0 and precision == "f64":
        return f"k_trace_tile<{ph}, false, {'true' if records else 'false'}>"  # <PHASE, AUX, REC>
    if precision == "mixed":
        return f"k_trace_mx<{ph}>" if substeps == 1 else f"k_trace_f64<{ph}, false, true, 0>"  # no mixed kernel for sub-steps: float64
    return f"k_trace_f64<{ph}, false, {'false' if substeps == 1 else 'true'}, 0>"  # <PHASE, AUX, SUBS, SEL>


# What a wave64 VALU instruction holds its SIMD for on gfx950 (MI355X_MICROARCH.md "vector-instruction ISSUE cost", and
# profiles/r02_valu_issue.json at four wavefronts per SIMD): 4 cycles for float64, packed float32, three-operand float32,
# conversions, 64-bit moves, compares and selects; 2 for the two-operand float32 / int32 forms; 8 / 16 for the float32 /
# float64 transcendental (rcp) -- the HARDWARE cost, whatever occupancy the kernel runs at.
HW_CYCLES = {"FMA_F64": 4, "ADD_F64": 4, "MUL_F64": 4, "TRANS_F64": 16, "FMA_F32": 4, "ADD_F32": 4, "MUL_F32": 4, "TRANS_F32": 8,
             "CVT": 4, "INT32": 2, "INT64": 4, "OTHER": 4}
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X float64 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
MODEL_STALE_REL = 0.10         # live kernel time vs the profiled launch's: beyond this the model is not this run's (boxes of the pool run
                               # the same binary 4-6 % apart -- per-ray kernel 58.3 / 59.2 / 60.5 ms on three of them this round)


def compulsory_hbm_bytes(kernel, n_rays, volume_bytes, launches):
    """What the kernel cannot avoid moving through HBM per trace: the volume ONCE (every node plane is needed by some
    workgroup: the packed volume, 16 B per node, 20 B with lo(n-1) -- for the records kernel the ready-made records instead, 128 B
    per node plane and lateral cell; `volume_bytes` is whichever the kernel reads) + the rays' state in and out.  Per ray: the launch from s0 reads 7 rows of s0
    + the 4-byte permutation (60 B); the launch that finishes writes sf (72) + rf (32) + Jf (32) + the edge guard's word (4) =
    140 B; the tile path hands the state from segment to segment through the hand-off records: first segment writes 10 rows
    (80 B), a middle one reads 7 and writes 7 (112 B), the last reads 9 (72 B)."""
    if kernel.startswith("k_trace_tile") and launches and launches > 1:
        per_ray = (60 + 80) + (launches - 2) * 112 + (72 + 140)
    else:
        per_ray = 60 + 140
    return {"bytes": float(volume_bytes + n_rays * per_ray), "volume_bytes_once": float(volume_bytes), "ray_bytes": float(n_rays * per_ray),
            "per_ray_bytes": per_ray}


def roofline(kernel, workload_key, kern_ms, ray_steps_per_launch, phase, build_id, n_rays=None, volume_bytes=None, launches=None):
    """The bound of the dominant kernel is VALU issue, not HBM (DESIGN.md "Measured").  `achieved` = SIMD issue cycles the
    launch's VALU instructions need at the HARDWARE cost per instruction class (HW_CYCLES) -- the instruction counts come
    from rocprofv3 --pmc passes on THIS build of the library, committed in profiles/kernel_model.json -- per second of the
    LIVE kernel time (HIP events around the launches of this run); `peak` = 1024 SIMDs x 2.4 GHz; `frac` their ratio.
    Beside it: the same priced at the issue rate the kernel's own occupancy allows (profiles/r02_valu_issue.json:
    `frac_at_kernel_occupancy`), the useful float64 FLOP rate against the vector peak, lane utilisation and the waiting share
    from the same passes, the HBM bytes of the PMC passes and SURVEY 8(d)'s algorithmic bytes against 8 TB/s.  The model is
    tied to the run that prints it: a model of another build is not printed, and neither is one whose profiled launch took
    more than 10 % longer or shorter than the live one (`"model": "stale"`); the ratio is printed either way."""
    bps = BYTES_PER_RAY_STEP[phase]
    t = kern_ms * 1e-3
    alg = ray_steps_per_launch * bps / t / 1e9
    out = {"bound": "valu", "achieved": None, "peak": SIMDS * CLOCK_GHZ, "unit": "G SIMD-issue-cycles/s", "frac": None,
           # frac = instruction counts of a PROFILED launch of this build (profiles/kernel_model.json) priced at the hardware's issue
           # cost, over THIS run's live kernel time: a model replayed on a live time, not a counter read in this process
           "modelled": True, "traffic": None,
           "kernel": kernel, "kernel_ms": kern_ms, "ray_steps_per_launch": ray_steps_per_launch,
           "algorithmic": {"bytes_per_ray_step": bps, "GBps": alg, "ratio_to_hbm_peak_NOT_A_BOUND": alg / HBM_PEAK_GBS,
                           "note": "not a bound, may exceed 1: SURVEY 8(d) counts every gather of the reference's algorithm (4 stages x 8 "
                                   "corners x 4 fields); the kernels serve them from registers / from LDS records shared by a cell's rays "
                                   "and read ONE new node plane per step, so HBM moves far less (see hbm and compulsory_hbm)"},
           "hbm": None, "compulsory_hbm": None, "model": None}
    if n_rays is not None and volume_bytes is not None:
        out["compulsory_hbm"] = compulsory_hbm_bytes(kernel, n_rays, volume_bytes, launches)
    try:
        model = json.load(open(KERNEL_MODEL))
        issue = json.load(open(VALU_ISSUE))["instructions"]
    except (OSError, ValueError, KeyError):
        out["model"] = "profiles/kernel_model.json or r02_valu_issue.json missing"
        return out
    ent = model.get("kernels", {}).get(kernel, {}).get(workload_key)
    if model.get("build_id") != build_id or ent is None:
        out["model"] = (f"profiles/kernel_model.json was measured on build {model.get('build_id')} (this library is {build_id}) "
                        f"or lacks {kernel} / {workload_key}: not printed")
        return out
    scale = ray_steps_per_launch / ent["ray_steps_per_launch"]
    prof_ms = ent.get("kernel_ms_profiled")
    if prof_ms and abs(kern_ms / (prof_ms * scale) - 1.0) > MODEL_STALE_REL:
        out["model"] = "stale"
        out["model_detail"] = {"kernel_ms_profiled": prof_ms * scale, "kernel_ms_live": kern_ms, "file": "profiles/kernel_model.json",
                               "note": "the profiled launch and this run differ by more than 10 % in kernel time: counts not applied"}
        return out
    # wavefronts per SIMD the kernel runs at: the mixed kernel and the tile path's RECORDS kernel 4, the per-ray float64 kernel 2, the
    # tile path's producers' kernel 3 (no such row in the issue-cadence table: priced as 2, and no occupancy figure printed)
    records_kernel = kernel.startswith("k_trace_tile") and kernel.endswith(", true>")
    col = "waves4" if ("mixed" in kernel or "_mx" in kernel or records_kernel) else "waves2"
    cyc = {k: v[col]["cycles"] for k, v in issue.items()}
    # class -> cycles per wave64 instruction at the kernel's occupancy.  rocprofv3's class counters count a packed-float32
    # instruction once, in the class of its operation (profiles/r02_valu_classes.csv); in these kernels the float32 adds
    # and multiplies are packed, so those classes are priced as v_pk_*.  OTHER = moves, selects, compares, min/max (no
    # class counter): priced as a 64-bit move / compare, the dearer of its members.
    price = {"FMA_F64": cyc["v_fma_f64"], "ADD_F64": cyc["v_add_f64"], "MUL_F64": cyc["v_mul_f64"], "TRANS_F64": cyc["v_rcp_f64"],
             "FMA_F32": cyc["v_pk_fma_f32"], "ADD_F32": cyc["v_pk_add_f32"], "MUL_F32": cyc["v_pk_mul_f32"], "TRANS_F32": cyc["v_rcp_f32"],
             "CVT": cyc["v_cvt_f64_f32"], "INT32": cyc["v_add_u32"], "INT64": cyc["v_lshl_add_u64"], "OTHER": cyc["v_mov_b64"]}
    per_launch = ent["valu_per_launch"]  # wave64 instructions per class, one launch of the workload
    need_hw = sum(per_launch.get(k, 0.0) * HW_CYCLES[k] for k in HW_CYCLES) * scale  # SIMD cycles at the hardware cost
    need_occ = sum(per_launch.get(k, 0.0) * price[k] for k in price) * scale         # ... at the kernel's own occupancy
    out["achieved"] = need_hw / t / 1e9
    out["frac"] = out["achieved"] / out["peak"]
    if records_kernel or not kernel.startswith("k_trace_tile"):  # the issue-cadence table (r02_valu_issue.json) has rows for 2 and 4
        out["frac_at_kernel_occupancy"] = need_occ / t / 1e9 / out["peak"]  # wavefronts per SIMD; the producers' tile kernel runs at 3
    flops = (2 * per_launch.get("FMA_F64", 0.0) + per_launch.get("ADD_F64", 0.0) + per_launch.get("MUL_F64", 0.0)) * 64 * scale
    out["f64_flops"] = {"TFLOPs": flops / t / 1e12, "peak_TFLOPs": F64_VECTOR_PEAK_TFLOPS, "frac": flops / t / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                        "fma_only_TFLOPs": 2 * per_launch.get("FMA_F64", 0.0) * 64 * scale / t / 1e12}
    if kernel.startswith("k_trace_tile"):
        # The tile kernel issues fewer instructions per ray-step than the per-ray kernel (no per-ray conversions, plane sums or
        # re-reads): its `frac` prices less work in less time.  The per-ray kernel's figures on the same workload, from the
        # same model file (SYNTHRAY_F64_TILE=0), for comparison.
        pr = model.get("kernels", {}).get(f"k_trace_f64<{'true' if phase else 'false'}, false, false, 0>", {}).get(workload_key)
        if pr and pr.get("kernel_ms_profiled"):
            pr_ms = pr["kernel_ms_profiled"] * scale
            out["per_ray_kernel"] = {"kernel_ms_profiled": pr_ms, "valu_instructions_per_wave_step": pr.get("valu_instructions_per_wave_step"),
                                     "frac": sum(pr["valu_per_launch"].get(k, 0.0) * HW_CYCLES[k] for k in HW_CYCLES) * scale / (pr_ms * 1e-3) / 1e9 / out["peak"],
                                     "fma_only_TFLOPs": 2 * pr["valu_per_launch"].get("FMA_F64", 0.0) * 64 * scale / (pr_ms * 1e-3) / 1e12,
                                     "this_kernel_valu_instructions_per_wave_step": ent.get("valu_instructions_per_wave_step"),
                                     "time_ratio_tile_over_per_ray": kern_ms / pr_ms}
    hb = ent.get("hbm_bytes_per_launch")
    if hb:
        out["traffic"] = hb * scale
        out["hbm"] = {"bytes_per_launch": hb * scale, "GBps": hb * scale / t / 1e9, "frac": hb * scale / t / 1e9 / HBM_PEAK_GBS,
                      "source": "rocprofv3 --pmc FETCH_SIZE (x2: the gfx950 unit) + WRITE_SIZE of the profiled launch"}
        if out["compulsory_hbm"]:
            out["hbm"]["over_compulsory"] = hb * scale / out["compulsory_hbm"]["bytes"]
    out["kernel_ms_live_over_profiled"] = kern_ms / (prof_ms * scale) if prof_ms else None
    out["model"] = {"file": "profiles/kernel_model.json", "source": ent.get("source"), "build_id": build_id,
                    "kernel_ms_profiled": prof_ms * scale if prof_ms else None, "valu_per_wave_step": ent.get("valu_per_wave_step"),
                    "hw_cycles_per_class": HW_CYCLES, "cycles_per_class_at_kernel_occupancy": {k: price[k] for k in per_launch if k in price},
                    "clock_ghz_measured": ent.get("clock_ghz"), "valu_busy_measured": ent.get("valu_busy"),
                    "lane_utilisation": ent.get("lane_utilisation"), "wait_any_frac_of_wave_cycles": ent.get("wait_any_frac_of_wave_cycles"),
                    "waves_per_simd_priced": None if (kernel.startswith("k_trace_tile") and not records_kernel) else col}
    return out


def api_flow(engine, ne, x, s0, ext, lwl, wl_diag, reps):
    """The same workload as a synthPy caller writes it (examples/jobs/run_scripts/pvti_trace_mpi.py:111-131,
    src/solvers-legacy/rtm_solver.py:142-178, 205-214, 376-453): host arrays into ScalarDomain.solve, host arrays out of it,
    the diagnostics classes built from them.  The classes find the bundle solve() left in HBM (synthpy_amd/resident.py) and
    deposit from it -- the fused kernel the job above times -- instead of uploading rf again.  Wall-clock per pass, outside the
    timed job; the last pass is quoted (the first page-locks the result arrays)."""
    import numpy as np

    from synthpy_amd.solvers_legacy import full_solver as fs, rtm_solver as rtm

    phase = wl_diag != "shadow+schlieren"
    dom = fs.ScalarDomain(x, x, x, ext, phaseshift=phase)
    dom.external_ne(ne)
    dom.calc_dndr(lwl)
    passes = []
    for _ in range(reps):
        lap = {}
        t0 = time.perf_counter()
        if phase:
            rf, Jf = dom.solve(s0, return_E=True)
        else:
            rf, Jf = dom.solve(s0), None
        t1 = time.perf_counter()
        lap["solve_ms"] = (t1 - t0) * 1e3
        on_device, sums = [], {}
        if wl_diag in ("interferometry", "all"):
            it = rtm.Interferometry(rf, E=Jf)
            it.two_lens_solve(wl=lwl)
            on_device.append(it.on_device)
            it.interferogram(bin_scale=1, clear_mem=True)
        t2 = time.perf_counter()
        if wl_diag in ("shadow+schlieren", "all"):
            sh = rtm.Shadowgraphy(rf)
            sh.two_lens_solve()
            on_device.append(sh.on_device)
            sh.histogram(bin_scale=1, clear_mem=True)
            sc = rtm.Schlieren(rf)
            sc.DF_solve()
            on_device.append(sc.on_device)
            sc.histogram(bin_scale=1, clear_mem=True)
        t3 = time.perf_counter()
        if wl_diag in ("interferometry", "all"):  # (the sums are the caller's business: outside the laps)
            sums["interferogram_sum"] = float(it.H.sum())
        if wl_diag in ("shadow+schlieren", "all"):
            sums["shadowgram_counts"], sums["schlieren_counts"] = int(sh.H.sum()), int(sc.H.sum())
        lap.update(interferometry_ms=(t2 - t1) * 1e3, counts_diagnostics_ms=(t3 - t2) * 1e3, total_ms=(t3 - t0) * 1e3,
                   deposits_from_hbm=bool(all(on_device)), tile_segments=dom._rays.tile_segments,
                   trace_kernel_ms=dom.trace_stats.trace_kernel_ms, **sums)
        passes.append(lap)
        del rf, Jf
    dom.clear_memory()
    last = passes[-1]
    return {"flow": "ScalarDomain.solve(s0" + (", return_E=True)" if phase else ")") +
                    (" -> Interferometry.two_lens_solve -> interferogram" if wl_diag in ("interferometry", "all") else "") +
                    (" -> Shadowgraphy.two_lens_solve -> histogram -> Schlieren.DF_solve -> histogram" if wl_diag in ("shadow+schlieren", "all") else "") +
                    " (synthpy_amd.solvers_legacy, bin_scale 1)",
            "host_arrays": "s0 (9, N) float64 in (pageable NumPy), rf (4, N)" + (" + Jf (2, N) complex128" if phase else "") + " out, H out",
            "rays": int(s0.shape[1]), "ms": last["total_ms"], "rays_per_s": s0.shape[1] / last["total_ms"] * 1e3,
            "last_pass": last, "passes_total_ms": [round(q["total_ms"], 2) for q in passes],
            "note": "not `value`: PCIe-inclusive (s0 up, rf / Jf / H down), one pass at a time"}


def init_device(engine, grp, shared=False):
    """One GPU per rank (engine.init_rank: fails if the job has more local ranks than GPUs; ranks starting in the same
    instant are the library's business, sr_device_count)."""
    engine.init_rank(grp.local_rank, grp.local_world, shared=shared)


def rehearse_on_one_gpu(engine, grp):
    """--rehearse-shared-gpu: RCCL's part is played by the host.  reduce_image = download, a sum through the control plane, and the sum kept
    beside the image for whoever downloads it next; everything else is the job as it runs on N GPUs."""
    sums = {}
    plain_download, plain_zero = engine.DetectorImage.download, engine.DetectorImage.zero

    def reduce_image(img, root=0):
        total = grp.reduce_host(plain_download(img), root=root)
        if total is not None:
            sums[id(img)] = total

    def download(img):
        return sums[id(img)] if id(img) in sums else plain_download(img)

    def zero(img):
        sums.pop(id(img), None)
        plain_zero(img)

    grp.reduce_image = reduce_image
    grp.comm_ranks = lambda: (grp.rank, grp.world)
    engine.DetectorImage.download, engine.DetectorImage.zero = download, zero


def build_id_of(version: str) -> str:
    return version.split("src:")[-1].strip() if "src:" in version else "unknown"


# --------------------------------------------------------------------------------------------------------------
# C5: the slab pipeline
# --------------------------------------------------------------------------------------------------------------
def bench_c5(args):
    """BASELINE configs[4]: a 1024^3 volume (--grid) cut into slabs of node planes along the probing axis, one per GPU, chunks
    of rays handed from GPU to GPU on the shared planes (RCCL send/recv), the last GPU deposits.  The volume is the real
    thing: gaussian3D.domain_fft(res = grid/2) with the k^-11/3 spectrum, its inverse FFT on the GPU.  At N = 1 the one GPU
    holds --slabs slabs and hands over in place: that measures what the cut costs (beside the slabs the WHOLE volume is
    resident too -- 21.5 GB of the 288 -- for the check).  A "step" = all --rays rays through the whole volume; the rays
    are drawn on the first slab's GPU (sr_rays_generate: init_beam's distributions, Philox stream), so no host upload
    sits in the pipeline (--host-rays uploads a host bundle per chunk instead, as the reference's drivers would).
    check (rank 0, N = 1): the chain of slabs against the WHOLE volume, bit for bit, and against the oracle FROM s0 on a
    1e5-ray sample (exit rays, shadowgram counts, interferogram); cpu_baseline: the oracle's trace of that sample."""
    import numpy as np

    from synthpy_amd import _ffi, engine
    from synthpy_amd.distributed import RayShardGroup, SlabPipeline

    grp = RayShardGroup()
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {grp.world}")
    rehearse = args.rehearse_shared_gpu and grp.world > 1  # ranks share device 0, hand-off through the host and the control plane
    init_device(engine, grp, shared=rehearse)
    if rehearse:
        grp.comm_ranks = lambda: (grp.rank, grp.world)
    n, ext, lwl = int(args.grid or 1024), 5e-3, 1064e-9
    n_rays = int(args.rays if args.rays is not None else (1e8 if grp.world > 1 else 2e7))
    n_slabs = grp.world if grp.world > 1 else args.slabs
    cuts = engine.slab_cuts(n, n_slabs)
    mine = [cuts[grp.rank]] if grp.world > 1 else cuts

    # ONE rank synthesises the field (tens of GB of host temporaries at 1024^3); the others map the file it leaves behind
    t0 = time.time()
    shared = os.path.join(os.environ.get("SYNTHRAY_TMP", "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"),
                          f"synthray_c5_{n}_{os.environ.get('MASTER_PORT', os.getpid())}.npy")
    if grp.rank == 0:
        ne, x = make_volume(n, device=True)
        if grp.world > 1:
            np.save(shared, ne)
    grp.barrier()
    if grp.rank != 0:
        ne, x = np.load(shared, mmap_mode="r"), np.linspace(-ext, ext, n)
    t_vol = time.time() - t0

    def slab_volume(lo, hi):
        return engine.Volume.from_ne_slab(engine.slab_source(ne, 2, lo, hi), x, x, x, lwl, "z", lo, hi, phaseshift=True)

    def flags(q, count):
        return (engine.HANDOFF_ENTER if q else 0) | (engine.HANDOFF_EXIT if q + 1 < count else 0)

    vols = [slab_volume(lo, hi) for lo, hi in mine]
    grp.barrier()
    if grp.rank == 0 and grp.world > 1:
        os.remove(shared)
    precision = engine.resolve_precision(args.precision, vols[0], handoff=1)
    # N > 1: the pipeline's plan (distributed.plan_chunks): the smallest chunk every rank still traces with the tile kernel, so that
    # the job is many chunks -- rank g idles g steps at the start and world - 1 - g at the end.  N = 1: one dense chunk.
    from synthpy_amd.distributed import beam_cells_of, plan_chunks

    box_cells = beam_cells_of([-4e-3, -4e-3, -ext, 4e-3, 4e-3, -ext], x, x, x, 2)  # the 4 mm beam's bounding box (sr_rays_generate's)
    plan = plan_chunks(n_rays, grp.world, box_cells, chunk=int(args.chunk) if args.chunk else (n_rays if grp.world == 1 else None),
                       cut="stripe" if args.stripe_chunks else "index")
    chunk, sizes = plan["chunk"], plan["sizes"]
    t_end = engine.default_t_end(ext)
    ns = int(min(args.cpu_sample, 100000, sizes[0]))
    # one host bundle: --host-rays re-uploads it per chunk; otherwise only the check's sample is drawn on the host
    s0_chunk = make_rays(max(sizes) if args.host_rays else max(ns, 1), ext, seed=0)
    stripes = None
    if args.stripe_chunks and grp.rank == 0:  # the job's rays: ONE host bundle, cut by position (rank 0 is where chunks enter)
        from synthpy_amd.distributed import stripe_chunks

        whole = make_rays(n_rays, ext, seed=0)
        stripes = [np.ascontiguousarray(whole[:, idx]) for idx in stripe_chunks(whole, sizes)]
        del whole
    beam_cells = np.pi * (4e-3 / (2 * ext / (n - 1))) ** 2  # lateral cells under the 4 mm beam
    img = engine.DetectorImage.complex_field(bin_scale=1)
    dep = [(img, engine.chain_shadow_two(), dict(kwave=2 * np.pi / lwl, ref_beam=(10, 10)))]
    pipe = SlabPipeline(grp, transport="host" if rehearse else "rccl")
    beam = dict(beam_size=4e-3, divergence=5e-5, ne_extent=ext, beam_type="circular", probing_direction="z", seed=0)
    kern_ms, tile_segs, tile_recs = [], [], []
    c5_bundles = {}  # N = 1: the bundles live as long as the job (their buffers are allocated once)

    def one_pass():
        img.zero()
        if grp.world > 1:
            return pipe.trace_chunks(vols[0], ext, sizes, (lambda m, ci: stripes[ci]) if stripes is not None else (lambda m, ci: s0_chunk[:, :m]),
                                     precision=precision, substeps=args.substeps, deposits=dep,
                                     device_beam=None if (args.host_rays or args.stripe_chunks) else beam)[0]
        steps, first = 0, 0
        dbg = os.environ.get("SYNTHRAY_BENCH_DEBUG")
        for ci, m in enumerate(sizes):
            key = (ci, m) if stripes is not None else m
            fresh = key not in c5_bundles
            r = c5_bundles.get(key) or c5_bundles.setdefault(key, engine.RayBundle(m))
            t_a = time.perf_counter()
            if stripes is not None:
                if fresh:  # once, in the first warm-up pass: the chunks are resident in HBM when the timed region starts
                    r.upload(stripes[ci])
            elif args.host_rays:
                r.upload(s0_chunk[:, :m])
            else:
                r.generate(first_ray=first, **beam)
            if dbg:
                engine.synchronize()
            t_b = time.perf_counter()
            first += m
            per = []
            for q, v in enumerate(vols):
                t_c = time.perf_counter()
                st = r.trace(v, t_end, ext, precision=precision, substeps=args.substeps, handoff=flags(q, len(vols)))
                per.append((time.perf_counter() - t_c) * 1e3)
                steps += st.ray_steps
                kern_ms.append(st.trace_kernel_ms)
                tile_segs.append(r.tile_segments)
                tile_recs.append(r.tile_records)
            t_d = time.perf_counter()
            for im, chain, kw in dep:
                r.deposit(im, chain, want_stats=False, **kw)
            if dbg:
                engine.synchronize()
                print(f"c5 pass: draw {1e3 * (t_b - t_a):.1f} ms, traces {[round(x, 1) for x in per]} ms, deposit {1e3 * (time.perf_counter() - t_d):.1f} ms",
                      file=sys.stderr)
        engine.synchronize()
        return steps

    for _ in range(args.warmup):
        one_pass()
    engine.synchronize()
    grp.barrier()
    kern_ms.clear()
    tile_segs.clear()
    tile_recs.clear()
    t_start = time.perf_counter()
    steps_total = 0
    for _ in range(args.steps):
        steps_total += one_pass()
    engine.synchronize()
    grp.barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t_start)
    all_steps = grp.sum_over_ranks(float(steps_total))
    ranks_seen = grp.comm_ranks()[1] if grp.world > 1 else 1
    if grp.world > 1 and ranks_seen != args.gpus:
        raise SystemExit(f"the data-path communicator reports {ranks_seen} ranks, the job was started with --gpus {args.gpus}")
    check, cpu = None, None
    if grp.rank == 0 and grp.world == 1 and ns > 0:
        # (a) the cut changes nothing: the sample through the chain of slabs == through the WHOLE volume, bit for bit
        r1 = engine.RayBundle(ns).upload(s0_chunk[:, :ns])
        for q, v in enumerate(vols):
            r1.trace(v, t_end, ext, precision=precision, substeps=args.substeps, handoff=flags(q, len(vols)))
        sf_chain, rf_chain, Jf_chain = r1.download()
        whole = engine.Volume.from_ne(ne, x, x, x, lwl, "z", phaseshift=True)
        r2 = engine.RayBundle(ns).upload(s0_chunk[:, :ns])
        st_w = r2.trace(whole, t_end, ext, precision=precision, substeps=args.substeps)
        sf_w = r2.download()[0]
        check = {"rays": ns, f"chain_of_{len(vols)}_slabs_equals_whole_volume_bitwise": bool(np.array_equal(sf_chain, sf_w, equal_nan=True)),
                 "max_dx_m_chain_vs_whole": float(np.nanmax(np.abs(sf_chain[:3] - sf_w[:3]))),
                 "max_dphase_rad_chain_vs_whole": float(np.nanmax(np.abs(sf_chain[7] - sf_w[7]))),
                 "nan_rays": int(np.isnan(sf_chain[0]).sum()), "whole_volume_kernel_ms_for_the_sample": st_w.trace_kernel_ms}
        # (b) against the oracle from the same s0, and the CPU baseline
        from oracle import oracle as orc  # the checker / reported CPU baseline, never the product

        orc.build()
        orc.set_num_threads(host_cores())
        tb = time.perf_counter()
        dom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=True)
        t_dom = time.perf_counter() - tb
        tc = time.perf_counter()
        sf_o, steps_o = orc.trace_rk4(dom, s0_chunk[:, :ns], (x[1] - x[0]) / orc.c, t_end, "z", "planes", args.substeps)
        rf_o, Jf_o = orc.ray_to_jones(sf_o, ext, "z")
        tc = time.perf_counter() - tc
        r_mm_o = orc.optics(rf_o, [(orc.SCALE, 1e3)])[0]
        Ho = orc.histogram(orc.optics(r_mm_o, orc.chain_shadow_two())[0], bin_scale=1)
        E_o = orc.interfere_ref_beam(rf_o, Jf_o, 10, 10)
        r_o, E_o = orc.optics(r_mm_o, orc.chain_shadow_two(), E_o, 2 * np.pi / lwl)
        Io = orc.interferogram(r_o, E_o, bin_scale=1)
        hc, hi_ = engine.DetectorImage.counts(bin_scale=1), engine.DetectorImage.complex_field(bin_scale=1)
        r1.deposit(hc, engine.chain_shadow_two())
        r1.deposit(hi_, engine.chain_shadow_two(), kwave=2 * np.pi / lwl, ref_beam=(10, 10))
        check.update({"vs": "oracle (CPU restatement) from the same s0, rays through the chain of slabs",
                      "max_dx_m": float(np.max(np.abs(rf_chain[0::2] - rf_o[0::2]))), "max_dtheta_rad": float(np.max(np.abs(rf_chain[1::2] - rf_o[1::2]))),
                      "max_dphase_rad": float(np.max(np.abs(sf_chain[7] - sf_o[7]))), "ray_steps_equal": bool(st_w.ray_steps == steps_o),
                      "H_counts_equal_from_s0": bool(np.array_equal(hc.download().astype(np.int64), Ho.astype(np.int64))),
                      "interferogram_from_s0_max_dH_over_max_H": float(np.max(np.abs(hi_.amplitude() - Io)) / np.max(Io))})
        cpu = {"value": steps_o / tc, "unit": "ray-steps/s", "cores": orc.num_threads(), "kind": "port", "rays_per_s": ns / tc,
               "sample": f"first {ns} rays of the first chunk's host bundle through the same {n}^3 volume, trace + back-projection, "
                         f"oracle/synthray_oracle.c with OpenMP over rays, {tc:.1f} s (its calc_dndr of the volume: {t_dom:.1f} s, not counted)"}
        whole.close()
    if grp.rank == 0:
        per_step_ms = sum(kern_ms) / args.steps if kern_ms else None
        out = {
            "metric": "ray-steps/sec (+ rays/sec to detector), slab-decomposed volume with ray hand-off",
            "value": all_steps / elapsed, "unit": "ray-steps/s", "rays_per_s": n_rays * args.steps / elapsed,
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64" if precision == "f64" else "f64 state+accumulation / f32 stage arithmetic", "data": "synthetic",
            "config": {"workload": f"C5: {n_rays:.3g} rays in chunks of {chunk:.3g} x {n}^3 k^-11/3 turbulent n_e (gaussian3D.domain_fft, res {n // 2}), "
                                   f"{n_slabs} slabs of node planes ({'one per GPU, RCCL hand-off' if grp.world > 1 else 'all on one GPU, hand-off in place'}), "
                                   "phase integral + interferogram on the last slab's GPU",
                       "grid": n, "slabs": n_slabs, "chunk": chunk, "precision": precision, "volume_setup_s": round(t_vol, 1),
                       "rays_per_lateral_cell_of_the_beam": chunk / beam_cells, "rays_per_lateral_cell_of_the_beams_box": plan["rays_per_beam_cell"],
                       "chunks_cut": ("stripes of the beam: ONE host bundle cut by position (distributed.stripe_chunks)" + (", resident before the timed region" if grp.world == 1 else ", uploaded by rank 0 chunk by chunk")
                                      if args.stripe_chunks else "host bundle uploaded per chunk" if args.host_rays else "index ranges of the device beam's Philox stream"),
                       "pipeline": {"chunks": plan["chunks"], "ranks": plan["ranks"], "fill_fraction": plan["fill_fraction"],
                                    "schedule": getattr(pipe, "schedule", "one GPU: slabs one after the other, hand-off in place")},
                       "kernel": ((f"k_trace_tile<true> on every slab (dense chunks; the records kernel on {sum(tile_recs)} of {len(tile_recs)} slab traces: "
                                   "a slab's records are kept while they fit in a third of the free HBM), k_trace_f64 for the rays a tile loses") if tile_segs and all(tile_segs)
                                  else ("k_trace_f64<true, false, false> (per-ray kernel)" if not any(tile_segs) else "mixed: " + str(sorted(set(tile_segs))))
                                  ) if grp.world == 1 else "every rank chooses by its chunk's density (sr_rays_tile_segments)",
                       "ranks_seen": ranks_seen, "library": _ffi.lib.sr_version().decode(),
                       "volume_hbm_bytes_this_rank": int(sum(v.nbytes for v in vols))},
            "roofline": (roofline(kernel_name(precision, True, args.substeps, 1 if tile_segs and all(tile_segs) else 0, bool(tile_recs) and all(tile_recs)), f"c5_{n}_{chunk}", per_step_ms,
                                  steps_total / args.steps, True, build_id_of(_ffi.lib.sr_version().decode()), n_rays=n_rays,
                                  volume_bytes=int(sum(v.nbytes for v in vols)), launches=len(vols) * max(1, max(tile_segs or [1]))) if per_step_ms else None),
            "cpu_baseline": cpu, "check": check,
        }
        if rehearse:
            out["rehearsal"] = "every rank on device 0, hand-off through the host and the control plane: value and ms_per_step are not a measurement"
            out["value"], out["rays_per_s"] = None, None
        emit(out)
    grp.barrier()
    grp.close()
    return 0


# --------------------------------------------------------------------------------------------------------------
# C2 / C3 / C4: ray-sharded
# --------------------------------------------------------------------------------------------------------------
def bench_rays(args):
    import numpy as np

    wl_rays, wl_grid, wl_diag = {"c2": (1e6, 256, "shadow+schlieren"), "c3": (1e7, 512, "interferometry"),
                                 "c4": (1.25e7, 512, "all")}[args.workload]
    args.rays = wl_rays if args.rays is None else args.rays
    args.grid = wl_grid if args.grid is None else args.grid
    if args.no_phase and wl_diag == "interferometry":
        wl_diag = "shadow+schlieren"

    from synthpy_amd import _ffi, engine
    from synthpy_amd.distributed import RayShardGroup, shard_range, shard_stripe

    grp = RayShardGroup()
    if grp.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {grp.world}")
    init_device(engine, grp, shared=args.rehearse_shared_gpu or args.shared_gpu_rccl)
    if args.rehearse_shared_gpu and grp.world > 1:
        rehearse_on_one_gpu(engine, grp)
    build_id = build_id_of(_ffi.lib.sr_version().decode())

    grid, ext, lwl = args.grid, 5e-3, 1064e-9
    phase = wl_diag != "shadow+schlieren"
    t0 = time.time()
    ne, x = make_volume(grid)
    t_vol = time.time() - t0
    vol = engine.Volume.from_ne(ne, x, x, x, lwl, "z", phaseshift=phase)
    precision = engine.resolve_precision(args.precision, vol)
    other = {"f64": "mixed", "mixed": "f64"}[precision]

    # the rays: weak = this rank's own seeded bundle; strong = this rank's share of ONE seeded bundle (a stripe of the beam, or an index range)
    share_world = args.share_of if args.share_of > 0 else grp.world  # --share-of N: this ONE GPU traces a rank's share of N

    def bundle_of(rank):
        if args.scaling == "weak" and not args.share_of:
            return make_rays(int(args.rays), ext, seed=rank, beam_size=args.beam_size)
        whole = make_rays(int(args.rays), ext, seed=0, beam_size=args.beam_size)
        if args.shard == "stripe":  # rows 0..2 are x, y, z and the beam probes along z: stripes along x
            return np.ascontiguousarray(whole[:, shard_stripe(whole[0], rank, share_world)])
        lo, hi = shard_range(int(args.rays), rank, share_world)
        return np.ascontiguousarray(whole[:, lo:hi])

    s0 = bundle_of(args.share_rank if args.share_of else grp.rank)
    n_rays = s0.shape[1]
    total_rays = n_rays if args.share_of else int(args.rays) * (grp.world if args.scaling == "weak" else 1)
    rays = engine.RayBundle(n_rays).upload(s0)  # inputs resident in HBM before the timed region
    t_end = engine.default_t_end(ext)

    def make_images():
        ims = []
        if wl_diag in ("interferometry", "all"):
            ims.append((engine.DetectorImage.complex_field(bin_scale=1), engine.chain_shadow_two(),
                        dict(kwave=2 * np.pi / lwl, ref_beam=(10, 10))))
        if wl_diag in ("shadow+schlieren", "all"):
            ims += [(engine.DetectorImage.counts(bin_scale=1), engine.chain_shadow_two(), {}),
                    (engine.DetectorImage.counts(bin_scale=1), engine.chain_schlieren(), {})]
        return ims

    images = make_images()

    def one_step(bundle=rays, ims=images, prec=precision):
        st = bundle.trace(vol, t_end, ext, substeps=args.substeps, sort_rays=not args.no_sort, precision=prec)
        dep_ms, hit, again = 0.0, 0, 0
        counts = [(img, chain) for img, chain, _ in ims if img.kind == engine.IMG_COUNTS]
        refined = len(counts) > 1
        if refined:  # ONE float64 re-trace of the rays near a bin or mask edge of any counts diagnostic (sr_rays_refine)
            again += bundle.refine(counts)
        for img, chain, kw in ims:  # the image ACCUMULATES over the steps of a job, as the reference's drivers sum
            ms, h = bundle.deposit(img, chain, exact_counts=not refined, **kw)  # their chunks' images (pvti_trace_mpi.py:144-163)
            dep_ms += ms
            hit += h
            again += bundle.retraced  # counts images of a mixed-precision trace: rays the edge guard traced again in float64
        one_step.retraced = again
        return st, dep_ms, hit

    def reduce_images(ims):  # ONE sum over the ranks per job (pvti_trace_mpi.py:169-170), inside the timed region
        for img, _, _ in ims:
            grp.reduce_image(img, root=0)

    for img, _, _ in images:
        img.zero()
    collective = "none (one GPU)" if grp.world == 1 else "host and control plane (rehearsal)" if args.rehearse_shared_gpu else "rccl"
    rccl_error = ""
    try:
        reduce_images(images)  # untimed: creates the RCCL communicator and its rings whatever --warmup is
    except RuntimeError as e:
        if grp.world == 1 or args.rehearse_shared_gpu:
            raise
        rccl_error = str(e)
    if collective == "rccl" and grp.sum_over_ranks(1.0 if rccl_error else 0.0) > 0:
        # The first N > 1 run of a build may meet a host on which RCCL cannot start.  A line that says so, with the job's steps
        # measured and the image sum taken through the host, tells more than a traceback; it is marked and is NOT the RCCL job.
        print(f"[bench rank {grp.rank}] RCCL COULD NOT SUM THE IMAGES ({rccl_error or 'another rank failed'}): "
              "this job sums them through the host and the control plane and marks its line", file=sys.stderr, flush=True)
        collective = "HOST FALLBACK, NOT RCCL: " + (rccl_error or "another rank's RCCL call failed")
        rehearse_on_one_gpu(engine, grp)
        reduce_images(images)
    for _ in range(args.warmup):
        one_step()
    for img, _, _ in images:
        img.zero()
    engine.synchronize()
    grp.barrier()
    t_start = time.perf_counter()
    k_ms, d_ms, steps_total, hits, guard_again = [], [], 0, 0, 0
    for _ in range(args.steps):
        st, dep_ms, hit = one_step()
        guard_again = one_step.retraced
        k_ms.append(st.trace_kernel_ms)
        d_ms.append(dep_ms)
        steps_total += st.ray_steps
        hits = hit
        fallback = st.fallback_rays
    tile_segs = rays.tile_segments  # the float64 tile path carried the headline's traces (library's choice: dense bundles)
    tile_recs = rays.tile_records   # ... with the records kernel (the records fit in HBM beside everything else)
    engine.synchronize()
    t_local = time.perf_counter() - t_start  # this rank's K steps, before the one collective of the job
    reduce_images(images)
    engine.synchronize()
    t_reduce = time.perf_counter() - t_start - t_local  # the RCCL sum as this rank saw it (waiting for the slowest rank included)
    grp.barrier()
    elapsed = grp.max_over_ranks(time.perf_counter() - t_start)
    # for the first real N > 1 run: every rank's own time for its steps, and the reduce, side by side (outside the timed region)
    per_rank_ms = [grp.sum_over_ranks(t_local * 1e3 if grp.rank == q else 0.0) for q in range(grp.world)] if grp.world > 1 else [t_local * 1e3]
    reduce_ms = [grp.sum_over_ranks(t_reduce * 1e3 if grp.rank == q else 0.0) for q in range(grp.world)] if grp.world > 1 else [t_reduce * 1e3]
    all_steps = grp.sum_over_ranks(float(steps_total))
    all_rays = grp.sum_over_ranks(float(n_rays * args.steps))

    # ---- the other build on the same rays, same volume, same deposits (outside the timed job) ----
    other_out = None
    if args.other_steps > 0:
        oims = make_images()
        one_step(ims=oims, prec=other)  # warm-up
        for img, _, _ in oims:
            img.zero()
        engine.synchronize()
        grp.barrier()
        t1 = time.perf_counter()
        o_k, o_steps = [], 0
        for _ in range(args.other_steps):
            st, _, _ = one_step(ims=oims, prec=other)
            o_k.append(st.trace_kernel_ms)
            o_steps += st.ray_steps
        reduce_images(oims)
        engine.synchronize()
        grp.barrier()
        o_elapsed = grp.max_over_ranks(time.perf_counter() - t1)
        o_all = grp.sum_over_ranks(float(o_steps))
        other_out = {"precision": other, "steps": args.other_steps, "value": o_all / o_elapsed, "unit": "ray-steps/s",
                     "rays_per_s": grp.sum_over_ranks(float(n_rays * args.other_steps)) / o_elapsed,
                     "ms_per_step": o_elapsed / args.other_steps * 1e3, "kernel": kernel_name(other, phase, args.substeps, rays.tile_segments, rays.tile_records),
                     "kernel_ms": float(np.mean(o_k))}
        for img, _, _ in oims:
            img.close()

    # ---- N > 1: the reduced image against the ranks' own totals and against ONE GPU doing every rank's rays ----
    multi = None
    if grp.world > 1:
        cims = [(engine.DetectorImage.counts(bin_scale=1), engine.chain_shadow_two(), {})]
        if phase:
            cims.append((engine.DetectorImage.complex_field(bin_scale=1), engine.chain_shadow_two(),
                         dict(kwave=2 * np.pi / lwl, ref_beam=(10, 10))))
        _, _, _ = one_step(ims=cims)
        dep_counts = grp.sum_over_ranks(float(rays.deposit(engine.DetectorImage.counts(bin_scale=1), engine.chain_shadow_two())[1]))
        reduce_images(cims)
        engine.synchronize()
        comm_rank, comm_size = grp.comm_ranks()
        if comm_size != args.gpus:  # the line must not be printed for a job that ran on fewer ranks than it was asked for
            raise SystemExit(f"the data-path communicator reports {comm_size} ranks, the job was started with --gpus {args.gpus}")
        ranks_ok = grp.sum_over_ranks(1.0 if comm_rank == grp.rank else 0.0)
        if grp.rank == 0:
            H_red = cims[0][0].download()
            A_red = cims[1][0].download() if phase else None
            solo = [(engine.DetectorImage.counts(bin_scale=1), engine.chain_shadow_two(), {})]
            if phase:
                solo.append((engine.DetectorImage.complex_field(bin_scale=1), engine.chain_shadow_two(),
                             dict(kwave=2 * np.pi / lwl, ref_beam=(10, 10))))
            for r in range(grp.world):  # rank 0 alone traces every rank's bundle
                sr0 = s0 if r == 0 else bundle_of(r)
                b = rays if r == 0 else engine.RayBundle(sr0.shape[1]).upload(sr0)
                one_step(bundle=b, ims=solo)
                if r:
                    b.close()
            H_solo = solo[0][0].download()
            multi = {"ranks_seen": int(comm_size), "comm_ranks_consistent": bool(ranks_ok == grp.world),
                     "counts_sum_equals_sum_of_deposited": bool(int(H_red.sum()) == int(dep_counts)),
                     "counts_image_equals_single_gpu_image": bool(np.array_equal(H_red, H_solo)),
                     "deposited_rays_all_ranks": int(dep_counts)}
            if phase:
                A_solo = solo[1][0].download()
                multi["interferogram_sums_max_diff_over_max"] = float(np.max(np.abs(A_red - A_solo)) / np.max(np.abs(A_solo)))
        grp.barrier()

    # ---- correctness next to the timing + the CPU baseline (rank 0, N = 1 only for the baseline) ----
    check, cpu = None, None
    if grp.rank == 0:
        ns = int(min(args.cpu_sample, n_rays))
        if ns > 0:
            from oracle import oracle as orc  # the checker / reported CPU baseline, never the product

            orc.build()
            orc.set_num_threads(host_cores())
            dom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=phase)
            tc = time.perf_counter()
            sf_o, steps_o = orc.trace_rk4(dom, s0[:, :ns], (x[1] - x[0]) / orc.c, t_end, "z", "planes", args.substeps)
            rf_o, Jf_o = orc.ray_to_jones(sf_o, ext, "z")
            tc = time.perf_counter() - tc
            # the oracle's own images FROM s0 (trace -> reference beam -> optics -> detector), the comparison the
            # reference's flow defines (rtm_solver.py:376-453)
            r_mm_o = orc.optics(rf_o, [(orc.SCALE, 1e3)])[0]
            Ho = orc.histogram(orc.optics(r_mm_o, orc.chain_shadow_two())[0], bin_scale=1)
            Io = None
            if phase:
                E_o = orc.interfere_ref_beam(rf_o, Jf_o, 10, 10)
                r_o, E_o = orc.optics(r_mm_o, orc.chain_shadow_two(), E_o, 2 * np.pi / lwl)
                Io = orc.interferogram(r_o, E_o, bin_scale=1)

            def gpu_vs_oracle(prec):
                rs = engine.RayBundle(ns).upload(s0[:, :ns])
                rs.trace(vol, t_end, ext, substeps=args.substeps, precision=prec)
                sf_g, rf_g, Jf_g = rs.download()
                c = {"precision": prec, "rays": ns, "max_dx_m": float(np.max(np.abs(rf_g[0::2] - rf_o[0::2]))),
                     "max_dtheta_rad": float(np.max(np.abs(rf_g[1::2] - rf_o[1::2]))),
                     "max_dphase_rad": float(np.max(np.abs(sf_g[7] - sf_o[7]))), "vs": "oracle (CPU restatement) from the same s0"}
                hc = engine.DetectorImage.counts(bin_scale=1)
                rs.deposit(hc, engine.chain_shadow_two())
                Hg = hc.download().astype(np.int64)
                c["H_counts_equal_from_s0"] = bool(np.array_equal(Hg, Ho.astype(np.int64)))
                c["H_counts_L1_diff_from_s0"] = int(np.abs(Hg - Ho.astype(np.int64)).sum())
                # the fused deposit against the oracle's optics + binning fed the GPU's own exit rays: exact by construction
                Hs = orc.histogram(orc.optics(orc.optics(rf_g, [(orc.SCALE, 1e3)])[0], orc.chain_shadow_two())[0], bin_scale=1)
                c["H_counts_equal_on_gpu_rays"] = bool(np.array_equal(Hg, Hs.astype(np.int64)))
                if phase:
                    hi_ = engine.DetectorImage.complex_field(bin_scale=1)
                    rs.deposit(hi_, engine.chain_shadow_two(), kwave=2 * np.pi / lwl, ref_beam=(10, 10))
                    Ig = hi_.amplitude()
                    c["interferogram_from_s0_max_dH_over_max_H"] = float(np.max(np.abs(Ig - Io)) / np.max(Io))
                    E_s = orc.interfere_ref_beam(rf_g, Jf_g, 10, 10)
                    r_s, E_s = orc.optics(orc.optics(rf_g, [(orc.SCALE, 1e3)])[0], orc.chain_shadow_two(), E_s, 2 * np.pi / lwl)
                    c["interferogram_on_gpu_rays_max_dH_over_max_H"] = float(np.max(np.abs(Ig - orc.interferogram(r_s, E_s, bin_scale=1))) / np.max(Io))
                    hi_.close()
                hc.close()
                rs.close()
                return c

            check = gpu_vs_oracle(precision)
            if other_out is not None:
                other_out["check"] = gpu_vs_oracle(other)
            # ... and the HEADLINE launch itself: the sample above is a sparse bundle of its own (the per-ray kernel), the timed job
            # traced all n_rays at once (dense: the tile path for precision f64).  The same rays, taken out of the full bundle's arrays.
            st_h = rays.trace(vol, t_end, ext, substeps=args.substeps, sort_rays=not args.no_sort, precision=precision)
            sf_h, rf_h, _ = rays.download(Jf=False)
            check["headline_launch"] = {
                "kernel": kernel_name(precision, phase, args.substeps, rays.tile_segments, rays.tile_records), "rays_in_launch": n_rays, "rays_compared": ns,
                "max_dx_m": float(np.max(np.abs(rf_h[0::2, :ns] - rf_o[0::2]))), "max_dtheta_rad": float(np.max(np.abs(rf_h[1::2, :ns] - rf_o[1::2]))),
                "max_dphase_rad": float(np.max(np.abs(sf_h[7, :ns] - sf_o[7]))), "ray_steps_equal_n_minus_1_times_rays": bool(st_h.ray_steps == (grid - 1) * args.substeps * n_rays),
                "vs": "oracle (CPU restatement) from the same s0: the first rays of the full bundle, out of the full launch's arrays"}
            del sf_h, rf_h
            if args.gpus == 1:
                cpu = {"value": steps_o / tc, "unit": "ray-steps/s", "cores": orc.num_threads(), "kind": "port",
                       "rays_per_s": ns / tc,
                       "sample": f"first {ns} rays of the same bundle through the same {grid}^3 volume, trace + back-projection, "
                                 f"oracle/synthray_oracle.c with OpenMP over rays, {tc:.1f} s"}
        if multi is not None:
            check = dict(check or {}, multi_gpu=multi)

    flow = None
    if grp.rank == 0 and grp.world == 1 and args.api_flow_reps > 0 and args.substeps == 1 and not args.no_sort:
        flow = api_flow(engine, ne, x, s0, ext, lwl, wl_diag, args.api_flow_reps)
        flow["engine_path_ms_per_step"] = elapsed / args.steps * 1e3
        flow["ms_over_engine_path"] = flow["ms"] / flow["engine_path_ms_per_step"]

    if grp.rank == 0:
        kern_ms = float(np.mean(k_ms))
        steps_per_launch = steps_total / args.steps
        wkey = f"{grid}_{n_rays}_{'phase' if phase else 'nophase'}"
        rec_bytes = grid * (grid - 1) ** 2 * 128  # the records kernel reads these, not the packed volume
        rl = roofline(kernel_name(precision, phase, args.substeps, tile_segs, tile_recs), wkey, kern_ms, steps_per_launch, phase, build_id,
                      n_rays=n_rays, volume_bytes=rec_bytes if (tile_segs and tile_recs) else vol.nbytes - (rec_bytes if tile_recs else 0),
                      launches=tile_segs or 1)
        rl["deposit_kernel_ms"] = float(np.mean(d_ms))
        if tile_segs:
            rl["launches_per_trace"] = tile_segs
            rl["note"] = (f"the tile path: {tile_segs} launches of k_trace_tile per trace (segments of node planes, rays binned again in between), "
                          "kernel_ms = their sum; rays a tile loses (fallback_rays) are carried to the end of the volume by k_trace_f64 on a side "
                          "stream, beside the remaining segments" + ("; the records kernel: 128-byte coefficient records per (node plane, lateral "
                          "cell) ready-made in HBM, brought into the tiles by LDS-DMA, four workgroups per CU" if tile_recs else ""))
        if other_out is not None:
            orl = roofline(other_out["kernel"], wkey, other_out["kernel_ms"], steps_per_launch, phase, build_id)
            other_out["roofline"] = {k: orl[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm")}
        out = {
            "metric": "ray-steps/sec (+ rays/sec to detector), 1e7 rays x 512^3 volume",
            "value": all_steps / elapsed,
            "unit": "ray-steps/s",
            "rays_per_s": all_rays / elapsed,
            "n_gpus": args.gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64" if precision == "f64" else "f64 state+accumulation / f32 stage arithmetic",
            "data": "synthetic",
            "config": {
                "workload": (args.workload.upper() + ": " if (int(args.rays), grid) == (int(wl_rays), wl_grid) else "") +
                            (f"rank {args.share_rank}'s share of {args.share_of} ({'a stripe of the beam' if args.shard == 'stripe' else 'an index range'}) of {int(args.rays):.3g} rays" if args.share_of else
                             f"{int(args.rays):.3g} rays/GPU" if args.scaling == "weak" else
                             f"{int(args.rays):.3g} rays in all ({'equal-count stripes' if args.shard == 'stripe' else 'contiguous index ranges'} of one seeded bundle)") +
                            f" x {grid}^3 k^-11/3 turbulent n_e (1e25 + 9e24*noise), RK4 {args.substeps} step/cell, " +
                            {"interferometry": "phase integral + reference beam + two-lens interferogram",
                             "shadow+schlieren": "two-lens shadowgraphy + dark-field schlieren",
                             "all": "phase integral; interferogram + two-lens shadowgraphy + dark-field schlieren"}[wl_diag] +
                            ", detector 3448x2574 (bin_scale 1)",
                "precision": precision, "rays_this_gpu": n_rays, "rays_all_gpus": total_rays, "grid": grid, "substeps": args.substeps,
                "sort_rays": not args.no_sort, "fallback_rays": int(fallback), "deposited_rays": int(hits),
                "edge_guard_retraced_rays_per_step": int(guard_again),
                "volume_setup_s": round(t_vol, 1), "volume_hbm_bytes": vol.nbytes, "library": _ffi.lib.sr_version().decode(),
            },
            "roofline": rl,
            "cpu_baseline": cpu,
            "check": check,
            "other_build": other_out,
            "api_flow": flow,
            "timing": {"per_rank_ms_for_the_steps": [round(v, 3) for v in per_rank_ms], "per_rank_ms_in_the_image_reduce": [round(v, 3) for v in reduce_ms],
                       "note": "timed job = K steps on every rank + ONE sum of the images over the ranks (RCCL); a rank that finishes its steps early "
                               "waits in the reduce, so its reduce time holds the slowest rank's lag"},
            "collective": collective,
        }
        if collective.startswith("HOST FALLBACK"):
            out["rccl_failed"] = ("RCCL could not sum the images on this host; they went through the host and the control plane: the job's time "
                                  "holds a host sum where it has an xGMI one, so it is not this metric's value (value: null; ms_per_step_with_the_host_sum "
                                  "and value_with_the_host_sum say what was timed, per_rank_ms_for_the_steps what the GPUs did)")
            out["value_with_the_host_sum"], out["ms_per_step_with_the_host_sum"] = out["value"], out["ms_per_step"]
            out["value"], out["rays_per_s"] = None, None
        if args.rehearse_shared_gpu:
            out["rehearsal"] = "every rank on device 0, image sum through the host and the control plane: value and ms_per_step are not a measurement"
            out["value"], out["rays_per_s"] = None, None
        emit(out)
    grp.barrier()  # the other ranks wait for rank 0's check before the group goes away
    grp.close()
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, argv)
    own_the_result_stream()
    if args.dry_control_plane:
        return dry_control_plane(args)
    if args.workload == "c5":
        return bench_c5(args)
    return bench_rays(args)


if __name__ == "__main__":
    sys.exit(main())
