"""CPU-only checks: the C-ABI library loads and exports exactly what include/synthray.h declares, fails
loudly without a GPU (no fallback), and the host-side inputs (beam, profiles, field generator) reproduce
the reference's seeded outputs."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build()
    from synthpy_amd import _ffi

    return _ffi


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "synthray.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built):
    declared = _declared_symbols()
    assert len(declared) >= 30
    lib = ctypes.CDLL(built.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, f"declared in synthray.h but not exported: {missing}"
    assert sorted(built.SYMBOLS) == declared, "ctypes table and header out of step"
    nm = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (sr_[a-z0-9_]+)$", nm, flags=re.M)))
    assert exported == declared, "library exports symbols the header does not declare (or the reverse)"


def test_device_code_is_gfx950(built):
    out = subprocess.run(["strings", "-a", built.LIB_PATH], capture_output=True, text=True).stdout
    assert "gfx950" in out and "gfx90a" not in out and "sm_" not in out


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under synthpy_amd/ may import or load it."""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "synthpy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                if re.search(r"^\s*(from|import)\s+oracle\b|liboracle|/root/reference", txt, flags=re.M):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


@pytest.mark.skipif(os.path.exists("/dev/kfd") and os.access("/dev/kfd", os.R_OK), reason="a GPU is visible")
def test_fails_loudly_without_gpu(built):
    """No CPU fallback: every compute entry raises with the HIP error text when no device is visible."""
    from synthpy_amd import engine

    assert engine.device_count() == 0
    x = np.linspace(-1, 1, 4)
    with pytest.raises(built.SynthrayError, match="no HIP device"):
        engine.Volume.from_ne(np.ones((4, 4, 4)), x, x, x, 1e-6)
    with pytest.raises(built.SynthrayError, match="no HIP device"):
        engine.hist2d(np.zeros(3), np.zeros(3), 4, 4, -1, 1, -1, 1)
    with pytest.raises(built.SynthrayError, match="no HIP device"):
        engine.optics(np.zeros((4, 3)), engine.chain_shadow_two())
    with pytest.raises(built.SynthrayError):
        engine.RayBundle(8)


def test_argument_validation_before_device(built):
    from synthpy_amd import engine

    with pytest.raises(ValueError):
        engine.axis_index("w")
    with pytest.raises(ValueError):
        engine.Volume.from_ne(np.ones((4, 4, 5)), np.arange(4), np.arange(4), np.arange(4), 1e-6)
    with pytest.raises(built.SynthrayError, match="ascending"):
        engine.Volume.from_ne(np.ones((4, 4, 4)), [0, 1, 1, 2], np.arange(4), np.arange(4), 1e-6)
    with pytest.raises(built.SynthrayError, match="chain length"):
        engine.optics(np.zeros((4, 1)), [(0, 1.0)] * 40)
    with pytest.raises(built.SynthrayError, match="unknown optic"):
        engine.optics(np.zeros((4, 1)), [(17, 1.0)])


# ---------------------------------------------------------------- host inputs vs the reference
def test_init_beam_reproduces_reference(built):
    from synthpy_amd.solvers_legacy import full_solver as fs

    g = golden("g0_beams")
    for bt, size in (("circular", 3e-3), ("square", 2e-3), ("rectangular", (1e-3, 2e-3)), ("linear", 4e-3)):
        for pd in "xyz":
            np.random.seed(int(g["seed"]))
            s0 = fs.init_beam(int(g["Np"]), size, float(g["divergence"]), float(g["ne_extent"]), bt, probing_direction=pd)
            assert np.array_equal(s0, g[f"{bt}_{pd}"]), (bt, pd)
    with pytest.raises(ValueError):
        fs.init_beam(10, 1e-3, 0, 5e-3, "even")
    with pytest.raises(ValueError):
        fs.init_beam(10, 1e-3, 0, 5e-3, "circular", probing_direction="q")


def test_jax_generation_beam(built):
    from synthpy_amd.simulator.beam import Beam

    b = Beam(1000, 4e-3, 5e-5, 5e-3, probing_direction="z", seeded=True)
    assert b.s0.shape == (9, 1000) and np.all(b.s0[2] == -5e-3) and np.all(b.s0[6] == 1)
    assert np.allclose(np.linalg.norm(b.s0[3:6], axis=0), 299792458.0, rtol=1e-15)
    assert np.max(np.hypot(b.s0[0], b.s0[1])) <= 4e-3
    # seeded: re-seeding with 0 before each draw (utils.py:8-24) -> the radial draw is np.random.power(2, N) from seed 0
    np.random.seed(0)
    u = np.random.power(2, 1000)
    assert np.allclose(np.hypot(b.s0[0], b.s0[1]), 4e-3 * u, rtol=1e-12)
    b2 = Beam(1000, 4e-3, 5e-5, 5e-3, probing_direction="x", beam_type="square", seeded=True)
    assert np.all(b2.s0[0] == -5e-3)


def test_profiles_reproduce_reference(built):
    """The integratedPy.npy recipe of the reference (evaluation/sergio_testing): test_linear_cos, bit-exact."""
    from synthpy_amd.solvers_legacy import full_solver as fs

    g = golden("g0_profiles")
    d = fs.ScalarDomain(g["x"], g["y"], g["z"], float(g["extent"]))
    d.test_linear_cos(s1=-1, s2=1, n_e0=1e26, Ly=5e-3)
    assert np.array_equal(d.ne, g["linear_cos"])
    assert np.array_equal(d.ne.sum(axis=2), g["linear_cos_sum"])
    d.test_exponential_cos()
    assert np.array_equal(d.ne, g["exponential_cos"])
    d.test_null()
    assert d.ne.shape == (20, 50, 10) and not d.ne.any()


def test_domain_fft_reproduces_reference(built):
    from synthpy_amd.field_generator.gaussian3D import gaussian3D

    g = golden("g0_domain_fft")
    np.random.seed(int(g["seed"]))
    f = gaussian3D(lambda k: k ** (-11 / 3)).domain_fft(float(g["l_max"]), float(g["l_min"]), int(g["extent"]), int(g["res"]),
                                                      float(g["factor"]))
    assert np.array_equal(f, g["field"])


def test_simulator_domain_surface(built):
    from synthpy_amd.simulator.domain import ScalarDomain

    d = ScalarDomain([10e-3, 10e-3, 20e-3], [8, 9, 10], ne_type="test_slab")
    assert d.ne.shape == (8, 9, 10) and d.x.dtype == np.float32 and d.region_count == 1
    assert np.allclose(d.z[[0, -1]], [-10e-3, 10e-3]) and d.lengths.tolist() == [10e-3, 10e-3, 20e-3]
    with pytest.raises(ValueError):
        d.external_ne(np.zeros((3, 3, 3)))


def test_driver_merges_consecutive_chunks():
    """run_trace.merge_groups: consecutive chunks of a rank into bundles of at least merge_rays rays; every chunk in exactly one
    group, in order; 0 = the reference's one chunk at a time (pvti_trace_mpi.py:144-163)."""
    from synthpy_amd import run_trace as rt

    chunks = rt.rank_chunks(10 ** 7 + 123, 5 * 10 ** 5, 0, 1)
    assert len(chunks) == 21 and chunks[0][1] == 123
    for merge in (0, 1, 5 * 10 ** 5, rt.MERGE_RAYS, 10 ** 9):
        groups = rt.merge_groups(chunks, merge)
        assert [q for grp in groups for q in grp] == list(range(len(chunks)))
        sizes = [sum(chunks[q][1] for q in grp) for grp in groups]
        if merge <= 0:
            assert len(groups) == len(chunks)
        elif merge >= 10 ** 9:
            assert len(groups) == 1
        else:
            assert all(s >= min(merge, sum(sizes)) // 2 for s in sizes) and all(s >= merge for s in sizes[:-1])
    assert len(rt.merge_groups(chunks, rt.MERGE_RAYS)) == 2  # 1e7 rays in the reference's chunks: two dense bundles
    assert rt.merge_groups([], 100) == []


def test_driver_chunk_split():
    """Remainder first, then full 5e5-ray chunks (pvti_trace_mpi.py:144-163); the split needs no GPU."""
    import ast
    import pathlib

    src = pathlib.Path(__file__).resolve().parents[1] / "synthpy_amd" / "run_trace.py"
    ns = {}
    tree = ast.parse(src.read_text())
    keep = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name == "chunk_sizes")
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "NP_RAY_SPLIT")]
    exec(compile(ast.Module(keep, []), str(src), "exec"), ns)
    cs = ns["chunk_sizes"]
    assert cs(10 ** 7) == [500000] * 20 and cs(1200000) == [200000, 500000, 500000]
    assert cs(3, 5) == [3] and cs(0) == [] and cs(10, 5) == [5, 5]


def test_small_host_mirrors():
    """annular_stop returns the mask and leaves r alone (rtm_solver.py:100-108, diagnostics.py:201-210); test_B is
    B_z = Bmax x / x_length (domain.py:493-503)."""
    from synthpy_amd.simulator import diagnostics as diag, domain as d
    from synthpy_amd.solvers_legacy import rtm_solver as rtm

    r = np.array([[0.5, 1.5, 2.5, np.nan], [0.0] * 4, [0.0, 0.0, 0.0, 0.0], [0.0] * 4])
    keep = r.copy()
    for fn in (rtm.annular_stop, diag.annular_stop):
        assert fn(r, 1.0, 2.0).tolist() == [False, True, False, False]
        assert np.array_equal(r, keep, equal_nan=True)
    dom = d.ScalarDomain(1e-2, (6, 5, 4), ne_type="test_null")
    dom.test_B(2.0)
    assert dom.B.shape == (6, 5, 4, 3) and not dom.B[..., :2].any()
    assert np.allclose(dom.B[:, 0, 0, 2], 2.0 * np.float32(np.linspace(-5e-3, 5e-3, 6)) / 1e-2)


def test_back_propogate_as_written():
    """propagator.py:300-349: straight-line projection onto the plane at ne_extent; the 'y' case swaps rows 0 and 2."""
    from synthpy_amd.simulator import propagator as p

    rng = np.random.default_rng(0)
    s = rng.standard_normal((9, 7))
    s[3:6] = 3e8 * (0.1 * rng.standard_normal((3, 7)) + 1.0)
    for pd, a in (("x", 0), ("y", 1), ("z", 2)):
        out = p.back_propogate(s, 5e-3, pd)
        t = (s[a] - 5e-3) / s[3 + a]
        want = s[:3] - s[3:6] * t
        assert np.all(out[a] == 5e-3) and np.array_equal(out[3:], s[3:])
        if pd == "y":
            assert np.allclose(out[0], want[2]) and np.allclose(out[2], want[0])
        else:
            b, c = [k for k in range(3) if k != a]
            assert np.allclose(out[b], want[b]) and np.allclose(out[c], want[c])


def test_driver_chunks_do_not_depend_on_the_number_of_gpus():
    """run_trace.rank_chunks: with the host ray source the ranks take whole chunks of the job's global chunk list (chunk c
    is the same seeded draw at any world size); with the device beam they take contiguous ray shards."""
    from synthpy_amd.run_trace import chunk_sizes, rank_chunks

    n, chunk = 2_300_001, 500_000
    whole = rank_chunks(n, chunk, 0, 1)
    assert [c[1] for c in whole] == chunk_sizes(n, chunk) and whole[0][2] == 0 and sum(c[1] for c in whole) == n
    for world in (2, 3, 8):
        parts = [rank_chunks(n, chunk, r, world) for r in range(world)]
        assert [c for p in parts for c in p] == whole  # the same chunks, each exactly once, in order
        dev = [rank_chunks(n, chunk, r, world, per_ray_stream=True) for r in range(world)]
        spans = sorted((c[2], c[2] + c[1]) for p in dev for c in p)
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_bench_roofline_is_recomputable_from_profiles(built):
    """bench.py's roofline object: the VALU-issue fraction is computed from committed files only (the per-class instruction
    counts of profiles/kernel_model.json, measured on THIS build of the library, priced at the hardware's issue cost) plus
    the live kernel time; it is <= 1; a model measured on another build, or whose profiled launch is more than 10 % away from
    the live kernel time, is refused, not printed."""
    import json

    import bench

    model = json.load(open(os.path.join(ROOT, "profiles", "kernel_model.json")))
    issue = json.load(open(os.path.join(ROOT, "profiles", "r02_valu_issue.json")))["instructions"]
    bid = bench.build_id_of(built.lib.sr_version().decode())
    assert len(bid) == 12
    if model["build_id"] != bid:  # the library's sources changed since the profiles were taken: bench.py prints no fraction either
        stale = bench.roofline(bench.kernel_name("f64", True), "512_10000000_phase", 60.0, 5.11e9, True, bid)
        assert stale["frac"] is None and "not printed" in stale["model"]
        pytest.skip(f"profiles/kernel_model.json is of build {model['build_id']}, the library is {bid}: re-run tools/make_profiles.sh a / b / c + collect_profiles.sh")
    # the headline's kernel: the tile path -- since round 5 its RECORDS kernel, four launches per trace priced together -- with the
    # per-ray kernel's figures on the same workload beside it (fewer instructions per ray-step in less time), and the producers'
    # kernel (three launches, what runs where the records do not fit) profiled on the same workload
    tile = bench.kernel_name("f64", True, tile_segments=4, records=True)
    prod = bench.kernel_name("f64", True, tile_segments=3)
    assert tile == "k_trace_tile<true, false, true>" and prod == "k_trace_tile<true, false, false>" and tile in model["kernels"] and prod in model["kernels"]
    rt = bench.roofline(tile, "512_10000000_phase", model["kernels"][tile]["512_10000000_phase"]["kernel_ms_profiled"], 5.11e9, True, bid)
    pr = rt["per_ray_kernel"]
    assert pr["this_kernel_valu_instructions_per_wave_step"] < 0.8 * pr["valu_instructions_per_wave_step"]
    assert pr["time_ratio_tile_over_per_ray"] < 0.8 and rt["f64_flops"]["fma_only_TFLOPs"] > 1.25 * pr["fma_only_TFLOPs"]
    assert model["kernels"][tile]["512_10000000_phase"]["launches_per_trace"] == 4 and model["kernels"][prod]["512_10000000_phase"]["launches_per_trace"] == 3
    assert model["kernels"][tile]["512_10000000_phase"]["kernel_ms_profiled"] < 0.93 * model["kernels"][prod]["512_10000000_phase"]["kernel_ms_profiled"]
    rec_bytes = 512 * 511 ** 2 * 128
    for prec, kern in (("f64", tile), ("f64", prod), ("f64", bench.kernel_name("f64", True)), ("mixed", bench.kernel_name("mixed", True))):
        ent = model["kernels"][kern]["512_10000000_phase"]
        launches = 4 if kern == tile else (3 if kern == prod else 1)
        vol_bytes = rec_bytes if kern == tile else 20 * 512 ** 3
        r = bench.roofline(kern, "512_10000000_phase", ent["kernel_ms_profiled"], 5.11e9, True, bid, n_rays=10 ** 7, volume_bytes=vol_bytes, launches=launches)
        assert r["bound"] == "valu" and r["modelled"] is True and 0.3 < r["frac"] <= 1.0
        if kern == prod:  # three wavefronts per SIMD: the issue-cadence table has no such row, so no such figure
            assert "frac_at_kernel_occupancy" not in r and r["model"]["waves_per_simd_priced"] is None
            continue_occ = False
        else:
            assert r["frac"] <= r["frac_at_kernel_occupancy"] <= 1.0
            continue_occ = True
        # HBM: counter bytes beside the bytes the kernel cannot avoid (volume once + ray state in and out [+ hand-off records])
        comp = r["compulsory_hbm"]
        assert comp["volume_bytes_once"] == vol_bytes and comp["per_ray_bytes"] == {4: 576, 3: 464, 1: 200}[launches]
        assert 1.0 <= r["hbm"]["over_compulsory"] < 6.0
        # by hand, `frac`: sum over classes of instructions x HARDWARE cycles, over SIMD-cycles available at the peak clock
        need = sum(ent["valu_per_launch"][k] * c for k, c in bench.HW_CYCLES.items())
        assert abs(need - ent["hw_issue_cycles_per_launch"]) <= 1e-6 * need
        assert abs(r["frac"] - need / (1024 * 2.4e9 * ent["kernel_ms_profiled"] * 1e-3)) < 1e-9
        # `frac_at_kernel_occupancy`: the same priced with the measured issue cadence at the kernel's waves per SIMD
        col = "waves4" if (prec == "mixed" or kern == tile) else "waves2"
        cyc = {k: v[col]["cycles"] for k, v in issue.items()}
        price = {"FMA_F64": "v_fma_f64", "ADD_F64": "v_add_f64", "MUL_F64": "v_mul_f64", "TRANS_F64": "v_rcp_f64", "FMA_F32": "v_pk_fma_f32",
                 "ADD_F32": "v_pk_add_f32", "MUL_F32": "v_pk_mul_f32", "TRANS_F32": "v_rcp_f32", "CVT": "v_cvt_f64_f32", "INT32": "v_add_u32",
                 "INT64": "v_lshl_add_u64", "OTHER": "v_mov_b64"}
        need_occ = sum(ent["valu_per_launch"][k] * cyc[price[k]] for k in price)
        if continue_occ:
            assert abs(r["frac_at_kernel_occupancy"] - need_occ / (1024 * 2.4e9 * ent["kernel_ms_profiled"] * 1e-3)) < 1e-9
        assert r["hbm"]["frac"] < 0.2 and r["algorithmic"]["ratio_to_hbm_peak_NOT_A_BOUND"] > 1.0  # HBM is not the bound; SURVEY's bytes are not HBM's
        # the hardware's own busy figure (4-cycle slots over the measured clock) against the hardware-cost fraction (2.4 GHz)
        # (the records kernel keeps more of the chip busy and runs at a lower clock: 2.22 GHz against the producers' 2.30)
        at_clock = r["frac"] * 2.4 / ent["clock_ghz"]
        assert 0.9 * at_clock < ent["valu_busy"] < 1.15 * at_clock
        assert 0.5 < r["model"]["lane_utilisation"] <= 1.0 and 0.0 < r["model"]["wait_any_frac_of_wave_cycles"] < 0.7
        assert 0.0 < r["f64_flops"]["frac"] < 0.6
        # tied to the run: a live kernel time 15 % off the profiled launch's prints no fraction
        off = bench.roofline(kern, "512_10000000_phase", 1.15 * ent["kernel_ms_profiled"], 5.11e9, True, bid)
        assert off["model"] == "stale" and off["frac"] is None and off["achieved"] is None
    for wl in ("256_1000000_nophase", "512_12500000_phase"):  # C2 and C4 per GPU have their own counts
        assert any(wl in v for v in model["kernels"].values()), wl
    stale = bench.roofline(bench.kernel_name("f64", True), "512_10000000_phase", 60.0, 5.11e9, True, "0" * 12)
    assert stale["frac"] is None and stale["traffic"] is None and "not printed" in stale["model"]


def test_every_tool_is_in_the_index():
    """tools/README.md says which script reproduces which file of profiles/ (the review of round 4: 54 scripts, no index)."""
    index = open(os.path.join(ROOT, "tools", "README.md")).read()
    missing = [f for f in sorted(os.listdir(os.path.join(ROOT, "tools"))) if f != "README.md" and not f.startswith("__") and f not in index]
    assert not missing, f"tools/README.md does not mention {missing}"


def test_product_does_not_import_torch():
    """north_star: "no PyTorch".  Importing the whole package -- the engine, the API mirrors, the multi-GPU layer and the job
    driver -- and bench.py leaves torch out of the process; gloo comes in only when a caller asks for control="gloo" (the
    CPU tests of the N > 1 path do)."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r); import bench; import synthpy_amd; "
            "from synthpy_amd import engine, distributed, run_trace, _rendezvous; "
            "from synthpy_amd.simulator import propagator, diagnostics, domain, beam; "
            "from synthpy_amd.solvers_legacy import full_solver, rtm_solver; "
            "g = distributed.RayShardGroup(rank=0, world=1); g.barrier(); g.close(); "
            "assert 'torch' not in sys.modules, sorted(m for m in sys.modules if m.startswith('torch'))[:5]" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]


def test_auto_batching_rule():
    """ScalarDomain.regions_for_memory: the reference's ceil(estimate * leeway / free) (domain.py:166-199) with this engine's
    bytes per node; never more regions than cell layers; 1 when everything fits."""
    from synthpy_amd import engine
    from synthpy_amd.simulator import domain as d

    D = d.ScalarDomain(0.01, 64, phaseshift=True, B_on=True)
    per_node = engine.volume_bytes_estimate(1, True, False, True)
    assert per_node == 16 + 4 + 32 + 8 + 4 + 32
    need = 64 ** 3 * per_node * D.leeway_factor
    assert D.auto_batching and D.region_count == 1
    assert D.regions_for_memory(free_bytes=2 ** 40) == 1
    assert D.regions_for_memory(free_bytes=int(need / 2.5)) == 3
    assert D.regions_for_memory(free_bytes=1000) == 63


def test_default_chunking_spreads_whole_chunks_evenly():
    """run_trace.default_chunk: with the default arguments (-r 1e7, no --chunk) and for 1e8 rays the busiest rank of
    2..8 holds at most 1.25 times the rays of the idlest one; the rule does not look at the number of GPUs; the device
    beam (rays sharded one by one) keeps the dense 1e7-ray chunks."""
    from synthpy_amd.run_trace import DEFAULT_CHUNK, default_chunk, rank_chunks

    for n_rays in (int(1e7), int(1e8), 12_345_678):
        chunk = default_chunk(n_rays)
        for world in range(2, 9):
            per_rank = [sum(n for _, n, _ in rank_chunks(n_rays, chunk, r, world)) for r in range(world)]
            assert sum(per_rank) == n_rays and min(per_rank) > 0
            assert max(per_rank) / min(per_rank) <= 1.26, (n_rays, world, per_rank)
    assert default_chunk(1e7, per_ray_stream=True) == DEFAULT_CHUNK and default_chunk(3000) == 94


def test_ray_farm_draws_the_same_chunks_in_any_number_of_workers():
    """run_trace.RayFarm: the host ray chunks drawn ahead by forked workers are the chunks ray_source gives one after the
    other (each chunk is its own seeded draw), whatever the number of workers; ragged sizes; /dev/shm blocks gone after
    close()."""
    import glob

    from synthpy_amd import run_trace as rt
    from synthpy_amd.solvers_legacy.full_solver import init_beam

    def ray_source(n, ci):
        np.random.seed(100 + ci)
        return init_beam(n, 4e-3, 5e-5, 5e-3, "circular", "z")

    chunks = rt.rank_chunks(2300, 500, 0, 1)  # remainder first: 300, then 4 x 500
    assert [n for _, n, _ in chunks] == [300, 500, 500, 500, 500]
    serial = [ray_source(n, ci) for ci, n, _ in chunks]
    for workers in (1, 3):
        before = set(glob.glob("/dev/shm/*"))
        farm = rt.RayFarm(ray_source, chunks, workers)
        try:
            for q in range(len(chunks)):
                s0, slot = farm.get(q)
                assert s0.shape == (9, chunks[q][1]) and np.array_equal(s0, serial[q])
                farm.release(slot)
        finally:
            farm.close()
        assert set(glob.glob("/dev/shm/*")) == before
    # a rank's share of a 2-rank job: the global chunk indices, not local ones
    share = rt.rank_chunks(2300, 500, 1, 2)
    farm = rt.RayFarm(ray_source, share, 2)
    try:
        for q, (ci, n, _) in enumerate(share):
            s0, slot = farm.get(q)
            assert np.array_equal(s0, ray_source(n, ci))
            farm.release(slot)
    finally:
        farm.close()


def test_driver_accepts_the_reference_command_line():
    """test_SynthRayTrace.py's flags (-d -r -f -m -c, :9-14) parse; -c is the host-core limit (cores - 1 ray workers),
    -f takes the reference's device names."""
    from synthpy_amd import run_trace as rt

    a = rt.build_parser().parse_args(["-d", "256", "-r", "1000000", "-f", "gpu", "-m", "32G", "-c", "8"])
    assert (a.domain, int(a.rays), a.force_device, a.memory, a.cores) == (256, 1000000, "gpu", "32G", 8)
    assert rt.device_choice("gpu") is None and rt.device_choice(None) is None and rt.device_choice("5") == 5  # None: this rank's own GPU
    for bad in ("cpu", "-1"):
        with pytest.raises(SystemExit):
            rt.device_choice(bad)


def test_resident_guard_logic():
    """synthpy_amd/resident.py without a GPU: which arrays give a diagnostic its bundle back.  Identity, the bundle's generation,
    write tracking (a write through Python to the array solve() handed out, or to any view of it, marks it for good), the
    sampled values for writers Python cannot see (every row is sampled: a rescaled / shifted / masked array is seen), E must be
    the Jf of the same solve; an entry goes when its array is collected and does not keep its bundle alive."""
    import gc
    import weakref

    from synthpy_amd import resident

    class Bundle:  # what attach() looks at
        def __init__(self):
            self.alive, self.generation, self.holders = True, 7, weakref.WeakSet()

    class Owner:
        pass

    def raw():
        rf, Jf = np.random.default_rng(0).normal(size=(4, 100000)), np.random.default_rng(1).normal(size=(2, 100000)) + 0j
        rf[:, 5] = np.nan  # a NaN column compares equal to itself (bit for bit)
        return rf, Jf

    b, o = Bundle(), Owner()
    rf0, Jf0 = raw()
    rf, Jf = resident.register(b, rf0, Jf0)
    assert isinstance(rf, np.ndarray) and np.shares_memory(rf, rf0) and np.shares_memory(Jf, Jf0)  # the same memory, no copy
    assert resident.attach(o, rf) is b and o in b.holders
    assert resident.attach(o, rf, Jf) is b
    assert resident.attach(o, rf.copy()) is None and resident.attach(o, rf, Jf.copy()) is None  # equal values, other objects
    assert resident.attach(o, rf0) is None  # the buffer solve() downloaded into, not the array it handed out
    assert resident.attach(o, [[0.0]]) is None
    b.generation += 1  # the bundle was traced / uploaded again
    assert resident.attach(o, rf) is None
    b.generation -= 1
    # writers the tracking cannot see (a base-class view): the probes, every row sampled
    rf, Jf = resident.register(b, rf0, Jf0)
    for change in (lambda a: a.__setitem__((slice(0, 4, 2), slice(None)), a[0:4:2] * 1e3), lambda a: a.__setitem__(1, a[1] + 1e-12),
                   lambda a: a.__setitem__((slice(None), slice(0, 2000)), np.nan)):
        keep = rf0.copy()
        change(np.asarray(rf))
        assert not resident.dirty(rf) and resident.attach(o, rf) is None
        rf0[...] = keep
        assert resident.attach(o, rf) is b
    # writes through Python: seen whatever their size, and for good (the array is not what solve() returned any more, even if
    # somebody puts the old values back: the reference would bin what the array holds, and so does the host path)
    writes = [lambda a: a.__setitem__((slice(None), 7), np.nan), lambda a: a.__setitem__((0, np.arange(a.shape[1]) == 99), 0.0),
              lambda a: a[0].__setitem__(3, 1.0), lambda a: np.copyto(a, 0.0), lambda a: a[0:4:2, :].__imul__(1e3),
              lambda a: np.multiply(a, 1.0, out=a), lambda a: a.T.__setitem__((0, 0), 1.0), lambda a: np.putmask(a, a > 9, 9.0),
              lambda a: a.reshape(-1).__setitem__(17, 0.0), lambda a: a.fill(0.0), lambda a: np.add.at(a, (0, 0), 1.0)]
    for w in writes:
        rf0, Jf0 = raw()
        rf, Jf = resident.register(b, rf0, Jf0)
        assert resident.attach(o, rf) is b
        w(rf)
        assert resident.dirty(rf) and resident.attach(o, rf) is None
    rf0, Jf0 = raw()
    rf, Jf = resident.register(b, rf0, Jf0)
    Jf[1, 12345] = 0.0  # one element of the field
    assert resident.attach(o, rf) is b and resident.attach(o, rf, Jf) is None
    # reading marks nothing, and what readers get back are plain arrays
    rf, Jf = resident.register(b, rf0, Jf0)
    got = [rf * 2, rf + rf, np.sqrt(np.abs(rf)), rf.copy(), rf[:, [1, 2]], rf[0][~np.isnan(rf[0])], np.where(rf > 0, rf, 0), rf.sum(0)]
    got[3][0, 0] = 5.0
    got[4][...] = 1.0
    assert not resident.dirty(rf) and resident.attach(o, rf) is b
    assert all(type(g) is np.ndarray for g in got[:3]) and all(getattr(g, "_sr_flag", None) is None for g in got)
    import pickle
    assert type(pickle.loads(pickle.dumps(rf))) is np.ndarray
    b.alive = False
    assert resident.attach(o, rf) is None
    # no Jf registered: a diagnostic that brings a field of its own takes the host path
    b2 = Bundle()
    rf2, _ = resident.register(b2, np.zeros((4, 10)))
    assert resident.attach(o, rf2) is b2 and resident.attach(o, rf2, np.zeros((2, 10), complex)) is None
    key = id(rf2)
    # an rf array somebody still holds does not keep the bundle (its HBM) alive
    wb = weakref.ref(b2)
    del b2
    gc.collect()
    assert wb() is None and resident.attach(o, rf2) is None
    del rf2
    gc.collect()
    assert key not in resident._entries
    assert resident.bundle_bytes(10 ** 7) > 3e9
