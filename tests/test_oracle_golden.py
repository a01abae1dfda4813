"""Pin the oracle (oracle/synthray_oracle.c) against the reference's own outputs.

Fixtures in tests/golden/ were produced by oracle/make_golden.py, which RUNS the
reference's legacy path (full_solver.py / rtm_solver.py) in the build container.
Everything here is CPU-only.
"""
import glob
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

FIELDS = ["g1_fields_a", "g1_fields_b", "g1_fields_c", "g1_fields_u"]
TRACES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g2_trace_*.npz")))


def _domain(orc, g, phaseshift=True):
    return orc.Domain(g["x"], g["y"], g["z"], g["dndx"], g["dndy"], g["dndz"], float(g["omega"]),
                      g["nref"] if phaseshift else None)


@pytest.mark.parametrize("name", FIELDS)
def test_calc_dndr_bit_exact(orc, name):
    """A1: omega and the three float32 gradient volumes equal the reference's bit for bit."""
    g = golden(name)
    om, gx, gy, gz = orc.calc_dndr(g["ne"], g["x"], g["y"], g["z"], float(g["lwl"]))
    assert om == float(g["omega"])
    for mine, ref in ((gx, g["dndx"]), (gy, g["dndy"]), (gz, g["dndz"])):
        assert ref.dtype == np.float32
        assert np.array_equal(mine, ref)


@pytest.mark.parametrize("name", FIELDS)
def test_n_refrac_bit_exact(orc, name):
    g = golden(name)
    assert np.array_equal(orc.n_refrac(g["ne"], float(g["omega"])), g["nref"])


@pytest.mark.parametrize("name", FIELDS)
def test_interp_bit_exact(orc, name):
    """A4: trilinear gather incl. on-node, on-face, out-of-bounds (fill) and NaN points."""
    g = golden(name)
    for k, fld in enumerate(("dndx", "dndy", "dndz")):
        mine = orc.interp(np.float32(g["x"]), np.float32(g["y"]), np.float32(g["z"]), g[fld], g["pts"], 0.0)
        assert np.array_equal(mine, g["grad"][k], equal_nan=True)


@pytest.mark.parametrize("name", FIELDS)
def test_dsdt_bit_exact(orc, name):
    """A3/A5: the full 9-component RHS (phaseshift on) equals the reference's dsdt bit for bit."""
    g = golden(name)
    mine = orc.dsdt(_domain(orc, g), g["s"])
    assert np.array_equal(mine, g["dsdt"], equal_nan=True)


def test_trilinear_of_linear_is_exact(orc):
    """Known answer of evaluation/interpolator_testing/int_val_check.ipynb cell 1: 3x3x3 grid, values=x, (0.5,0,0) -> 0.5."""
    x = np.float32(np.linspace(0, 2, 3))
    vals = np.float32(np.broadcast_to(x[:, None, None], (3, 3, 3)).copy())
    assert orc.interp(x, x, x, vals, np.array([[0.5, 0.0, 0.0]]), 0.0)[0] == 0.5


# ---------------------------------------------------------------- trace
def _setup_trace(orc, g):
    x = g["x"]
    lwl = float(g["lwl"])
    dom = orc.Domain.from_ne(g["ne"], x, x, x, lwl, phaseshift=bool(g["phaseshift"]))
    ext = float(g["extent"])
    dx = float(x[1] - x[0])
    return dom, ext, dx


def _errors(orc, g, sf):
    ext, pdir = float(g["extent"]), str(g["pdir"])
    rf, Jf = orc.ray_to_jones(sf, ext, pdir, "legacy")
    rt, st = g["rf_tight"], g["sf_tight"]
    return (np.max(np.abs(rf[0::2] - rt[0::2])), np.max(np.abs(rf[1::2] - rt[1::2])), np.max(np.abs(sf[7] - st[7])),
            np.max(np.abs(sf[:3] - st[:3])), np.max(np.abs(Jf - g["Jf_tight"])))


@pytest.mark.parametrize("name", TRACES)
def test_rk4_planes_vs_reference_tight(orc, name):
    """A2+A6, the production integrator (plane-to-plane RK4, one step per cell) against the reference
    RHS integrated by solve_ivp at rtol=1e-10 (the integrator-independent answer).
    Tolerances: exit position <=1e-8 m, exit angle <=1e-6 rad; final state at t_end <=2e-8 m;
    phase <=1e-5 of its magnitude (the 32^3 turbulence fixture is noise at the grid scale: 2e-3 of 320 rad)."""
    g = golden(name)
    dom, ext, dx = _setup_trace(orc, g)
    sf, steps = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), str(g["pdir"]), "planes", 1)
    pos, ang, ph, sfpos, jf = _errors(orc, g, sf)
    assert pos <= 1e-8 and ang <= 1e-6 and sfpos <= 2e-8
    phmax = max(1.0, np.max(np.abs(g["sf_tight"][7])))
    assert ph <= 1e-5 * phmax
    assert jf <= 1e-5 * phmax
    assert steps == (len(g["x"]) - 1) * g["s0"].shape[1]


@pytest.mark.parametrize("name", [t for t in TRACES if "turb" in t or "blob32" in t])
def test_rk4_planes_converges(orc, name):
    """Two sub-steps per cell cut the error >=3x (4th order until a ray crosses a lateral cell face mid-step)."""
    g = golden(name)
    dom, ext, dx = _setup_trace(orc, g)
    e = []
    for sub in (1, 2):
        sf, _ = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), str(g["pdir"]), "planes", sub)
        e.append(_errors(orc, g, sf))
    assert e[1][0] <= e[0][0] / 1.9 and e[1][2] <= e[0][2] / 3


@pytest.mark.parametrize("name", TRACES)
def test_rk4_time_stepping_fallback(orc, name):
    """The time-stepping form with located faces (used for rays the plane form cannot take):
    first order in h because of the field's kinks, still <=1e-8 m / 2e-5 rad at one cell per step
    (the state at t_end carries the angle error over the vacuum leg: <=1e-7 m)."""
    g = golden(name)
    dom, ext, dx = _setup_trace(orc, g)
    sf, _ = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), str(g["pdir"]), "time")
    pos, ang, ph, sfpos, _ = _errors(orc, g, sf)
    assert pos <= 1e-8 and ang <= 2e-5 and sfpos <= 1e-7
    assert ph <= 2e-4 * max(1.0, np.max(np.abs(g["sf_tight"][7])))


@pytest.mark.parametrize("name", TRACES)
def test_within_reference_default_tolerance_error(orc, name):
    """Against the reference AS SHIPPED (RK45 rtol=1e-3): the difference is the reference's own integration
    error, bounded by what the reference shows against its own tight run (+ ours)."""
    g = golden(name)
    dom, ext, dx = _setup_trace(orc, g)
    sf, _ = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), str(g["pdir"]), "planes", 1)
    rf, _ = orc.ray_to_jones(sf, ext, str(g["pdir"]), "legacy")
    rd, rt = g["rf_default"], g["rf_tight"]
    own_pos = np.max(np.abs(rd[0::2] - rt[0::2]))
    own_ang = np.max(np.abs(rd[1::2] - rt[1::2]))
    assert np.max(np.abs(rf[0::2] - rd[0::2])) <= own_pos + 1e-8
    assert np.max(np.abs(rf[1::2] - rd[1::2])) <= own_ang + 1e-6


def test_ray_to_jones_bit_exact_positions(orc):
    """A6 on the reference's own final state: positions bit-exact; angles/Jones within 2 ulp (libm)."""
    for name in TRACES:
        g = golden(name)
        rf, Jf = orc.ray_to_jones(g["sf_default"], float(g["extent"]), str(g["pdir"]), "legacy")
        assert np.array_equal(rf[0::2], g["rf_default"][0::2])
        assert np.allclose(rf[1::2], g["rf_default"][1::2], rtol=4e-16, atol=0)
        assert np.allclose(Jf, g["Jf_default"], rtol=0, atol=4e-16)


def test_null_and_slab_known_answers(orc):
    """Docstring known answers (full_solver.py:13-82): empty cube -> straight lines;
    linear slab -> uniform acceleration dv_x/dt = -c^2/2 * s*n_e0/(n_c*extent)."""
    g = golden("g2_trace_null16_z_s0")
    dom, ext, dx = _setup_trace(orc, g)
    sf, _ = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), "z")
    rf, _ = orc.ray_to_jones(sf, ext, "z")
    s0 = g["s0"]
    assert np.array_equal(sf[3:6], s0[3:6])
    assert np.allclose(rf[1], np.arctan(s0[3] / s0[5]), rtol=1e-15, atol=0)
    assert np.allclose(rf[0], s0[0] + s0[3] * (2 * ext / s0[5]), rtol=1e-12, atol=1e-15)
    g = golden("g2_trace_slab16_z_s0")
    dom, ext, dx = _setup_trace(orc, g)
    sf, _ = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), "z")
    nc = 3.14207787e-4 * dom.omega ** 2
    acc = -0.5 * orc.c ** 2 * 2e23 / (nc * ext)
    t_in = (np.float64(np.float32(ext)) * 2) / orc.c  # collimated beam: time between the two node planes
    assert np.allclose(sf[3], acc * t_in, rtol=1e-6)


# ---------------------------------------------------------------- optics / deposit
CHAINS = {
    "shadow_single": lambda o: o.chain_shadow_single(),
    "shadow_two": lambda o: o.chain_shadow_two(),
    "shadow_two_fp": lambda o: o.chain_shadow_two(L=350, R=20, focal_plane=3.0),
    "schlieren_df": lambda o: o.chain_schlieren(),
    "schlieren_lf": lambda o: o.chain_schlieren(stop_R=2, dark_field=False),
    "refracto": lambda o: o.chain_refractometry(),
}


@pytest.mark.parametrize("name", sorted(CHAINS))
def test_optics_chain_bit_exact(orc, name):
    """A7/A8: every fixed chain of the reference, ray for ray, NaN columns included."""
    g = golden("g3_optics")
    r, _ = orc.optics(orc.m_to_mm(g["rf"]), CHAINS[name](orc))
    assert np.array_equal(r, g[name + "_rf"], equal_nan=True)


@pytest.mark.parametrize("name", sorted(CHAINS))
def test_histogram_counts_exact(orc, name):
    """A9: integer counts equal np.histogram2d's (as run by the reference) at both detector sizes."""
    g = golden("g3_optics")
    rf = g[name + "_rf"]
    H = orc.histogram(rf, bin_scale=10)
    assert H.shape == (257, 344) and np.array_equal(H, g[name + "_H10"])
    H = orc.histogram(rf, bin_scale=1, pix_x=64, pix_y=48)
    assert np.array_equal(H, g[name + "_H64x48"])
    assert H.sum() == np.count_nonzero((np.abs(rf[0]) <= 9) & (np.abs(rf[2]) <= 6.75))


def test_histogram_edge_semantics(orc):
    """Values exactly on interior edges go up, on the last edge into the last bin, NaN/outside dropped."""
    g = golden("g3_optics")
    H = orc.histogram(g["edge_pts"], bin_scale=10)
    assert np.array_equal(H, g["edge_H10"])


def test_interferometry_chain_and_image(orc):
    """A8 (Interferometry.two_lens_solve) + A10: positions bit-exact; the complex field within
    1e-6 (k*|dr| ~ 3e8 rad amplifies one-ulp position differences to ~5e-8 rad); image within 1e-6 of max."""
    g = golden("g3_optics")
    k = 2 * np.pi / 532e-9
    r, E = orc.optics(orc.m_to_mm(g["rf"]), orc.chain_shadow_two(), E=g["E"], kwave=k)
    assert np.array_equal(r, g["interf_rf"], equal_nan=True)
    ok = ~np.isnan(g["interf_rE"][0])
    assert np.array_equal(ok, ~np.isnan(E[0]))
    assert np.max(np.abs(E[:, ok] - g["interf_rE"][:, ok])) <= 1e-6
    for kw, key in ((dict(bin_scale=10), "interf_H10"), (dict(bin_scale=1, pix_x=40, pix_y=30), "interf_H40x30")):
        H = orc.interferogram(g["interf_rf"], g["interf_rE"], **kw)
        assert H.shape == g[key].shape
        assert np.max(np.abs(H - g[key])) <= 1e-12 * max(1.0, g[key].max())
    H = orc.interferogram(g["interf_edge_pts"], g["E"][:, :400], bin_scale=10)
    assert np.allclose(H, g["interf_edge_H10"], rtol=0, atol=1e-12)


def test_interfere_ref_beam_transcription(orc):
    """A11 exists only in the JAX file (diagnostics.py:559-581); check the restatement against a
    NumPy transcription of that formula."""
    g = golden("g3_optics")
    rf, E = g["rf"], g["E"]
    for nf, deg in ((10, 20), (10, 10), (120, -20), (7, 60)):
        d = -abs(deg - 90) if deg >= 45 else deg
        rad = d * np.pi / 180
        yw = np.arctan(rad)
        xw = np.sqrt(1 - yw ** 2)
        want = E.copy()
        want[1] = want[1] + np.exp(2 * nf / 3 * 1.0j * (xw * rf[0] + yw * rf[2]))
        got = orc.interfere_ref_beam(rf, E, nf, deg)
        ok = ~np.isnan(rf[0])
        assert np.allclose(got[:, ok], want[:, ok], rtol=0, atol=5e-16)


# ---------------------------------------------------------------- A3's optional terms: inverse bremsstrahlung, Faraday rotation
AUX_TRACES = ["g5_trace_aux24_z", "g5_trace_aux20_x"]


def test_kappa_field_bit_exact(orc):
    """ScalarDomain.kappa() (full_solver.py:243-268) incl. external_Te's clamp at Te_min; VerdetConst (:223)."""
    g = golden("g5_fields_aux")
    assert np.array_equal(np.maximum(1.0, g["Te_in"]), g["Te"])
    assert np.array_equal(orc.kappa(g["ne"], g["Te"], g["Z"], float(g["omega"])), g["kappa"])
    assert orc.verdet(float(g["lwl"])) == float(g["verdet"])


def test_dsdt_aux_terms_bit_exact(orc):
    """All nine rows of dsdt with inv_brems, phaseshift and B_on switched on equal the reference bit for bit
    (rows 6 and 8: atten(x)*a and VerdetConst*ne*sum(B*v))."""
    g = golden("g5_fields_aux")
    dom = orc.Domain.from_ne(g["ne"], g["x"], g["y"], g["z"], float(g["lwl"]), True, g["Te"], g["Z"], g["B"])
    assert np.array_equal(dom.kappa, g["kappa"])
    out = orc.dsdt(dom, g["s"])
    ref = g["dsdt"]
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    ok = ~np.isnan(ref)
    assert np.array_equal(out[ok], ref[ok])
    assert np.count_nonzero(ref[6][ok[6]]) > 200 and np.count_nonzero(ref[8][ok[8]]) > 200
    for k, (vals, fill) in enumerate(((g["kappa"], 0.0), (g["ne"], 0.0))):
        got = orc.interp(g["x"].astype(np.float32), g["y"].astype(np.float32), g["z"].astype(np.float32), vals, g["pts"], fill)
        want = (g["kappa_at"], g["ne_at"])[k]
        m = ~np.isnan(want)
        assert np.array_equal(got[m], want[m])


def _aux_domain(orc, g):
    x = g["x"]
    return orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), True, g["Te"], g["Z"], g["B"])


@pytest.mark.parametrize("name", AUX_TRACES)
def test_aux_trace_vs_reference_tight(orc, name):
    """Amplitude and polarisation rows of the final state against solve_ivp(rtol=1e-10) over the reference's RHS:
    relative 1e-6 of the change they accumulate; positions/angles/phase as for the plain trace."""
    g = golden(name)
    dom, ext, pd = _aux_domain(orc, g), float(g["extent"]), str(g["pdir"])
    dx = g["x"][1] - g["x"][0]
    st = g["sf_tight"]
    for mode, tol in (("planes", 1e-6), ("time", 2e-4)):
        sf, _ = orc.trace_rk4(dom, g["s0"], dx / orc.c, orc.default_t_end(ext), pd, mode, 1)
        d_amp, d_pol = np.max(np.abs(st[6] - g["s0"][6])), np.max(np.abs(st[8] - g["s0"][8]))
        assert d_amp > 1e-3 and d_pol > 1e-3  # the terms are active in the fixture
        assert np.max(np.abs(sf[6] - st[6])) <= tol * d_amp, mode
        assert np.max(np.abs(sf[8] - st[8])) <= tol * d_pol, mode
        rf, Jf = orc.ray_to_jones(sf, ext, pd, "legacy")
        assert np.max(np.abs(rf[0::2] - g["rf_tight"][0::2])) <= 1e-8
        assert np.max(np.abs(Jf - g["Jf_tight"])) <= (1e-5 if mode == "planes" else 2e-4) * max(1.0, np.max(np.abs(st[7])))


@pytest.mark.parametrize("name", AUX_TRACES)
def test_aux_terms_do_not_change_the_trajectory(orc, name):
    """amp and pol are passive: rows 0-5 and 7 with the terms on equal the plain trace bit for bit."""
    g = golden(name)
    ext, pd, x = float(g["extent"]), str(g["pdir"]), g["x"]
    full = _aux_domain(orc, g)
    plain = orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), True)
    a, _ = orc.trace_rk4(full, g["s0"], (x[1] - x[0]) / orc.c, orc.default_t_end(ext), pd, "planes", 1)
    b, _ = orc.trace_rk4(plain, g["s0"], (x[1] - x[0]) / orc.c, orc.default_t_end(ext), pd, "planes", 1)
    assert np.array_equal(a[[0, 1, 2, 3, 4, 5, 7]], b[[0, 1, 2, 3, 4, 5, 7]])
    assert np.array_equal(b[6], g["s0"][6]) and np.array_equal(b[8], g["s0"][8])


# ---------------------------------------------------------------- coherent refractometer, knife edge
def test_coherent_refractometer_and_speckle(orc):
    """Refractometry.coherent_solve (rtm_solver.py:288-331): ray positions bit-exact, field to 1e-9 (k*|dr| ~ 1e9 rad
    as written, wavelength in m against mm); refractogram with np.random.seed(9) (:333-369) to the field's accuracy."""
    g = golden("g6_optics_extra")
    r0 = orc.optics(g["rf"], [(orc.SCALE, 1e3)])[0]
    assert np.array_equal(r0[0::2, :2990], g["rf"][0::2, :2990] * 1e3)
    r, E = orc.optics(r0, orc.chain_refractometry_coherent(focal_plane=2.0), g["E"], 2 * np.pi / 1064e-9)
    ok = ~np.isnan(g["coh_rf"][0])
    assert np.array_equal(np.isnan(r[0]), ~ok) and np.array_equal(r[:, ok], g["coh_rf"][:, ok])
    assert np.array_equal(np.isnan(E[0]), ~ok) and np.max(np.abs(E[:, ok] - g["coh_rE"][:, ok])) <= 1e-6
    np.random.seed(9)
    H = orc.interferogram(r, orc.speckle(r, g["coh_rE"]), bin_scale=10)  # reference field in: isolates the binning + speckle
    assert H.shape == g["coh_H10_seed9"].shape and np.max(np.abs(H - g["coh_H10_seed9"])) <= 1e-12


def test_knife_edge_bit_exact(orc):
    g = golden("g6_optics_extra")
    r0 = orc.optics(g["rf"], [(orc.SCALE, 1e3)])[0]
    for k, (off, row, dr) in enumerate(g["knife_args"]):
        r = orc.optics(r0, [(orc.KNIFE, off, dr, int(row))])[0]
        ref = g[f"knife{k}"]
        assert np.array_equal(np.isnan(r), np.isnan(ref)) and np.array_equal(r[~np.isnan(ref)], ref[~np.isnan(ref)])
        assert 100 < np.isnan(ref[0]).sum() < 2900


def test_phase_only_travel(orc):
    """SR_OP_PHASE: the JAX coherent_solve as written (diagnostics.py:505-511) = the field factor of the first travel
    with the ray left where it was."""
    g = golden("g6_optics_extra")
    r0 = orc.optics(g["rf"], [(orc.SCALE, 1e3)])[0]
    k = 2 * np.pi / 1064e-9
    r_d, E_d = orc.optics(r0, [(orc.DIST, 298.0)], g["E"], k)
    r_p, E_p = orc.optics(r0, [(orc.PHASE, 298.0)], g["E"], k)
    ok = ~np.isnan(r0[0])
    assert np.array_equal(r_p[:, ok], r0[:, ok]) and np.array_equal(E_p[:, ok], E_d[:, ok]) and not np.array_equal(r_d[:, ok], r0[:, ok])


# ---------------------------------------------------------------- A12: slab-to-slab hand-off
@pytest.mark.parametrize("name", ["g2_trace_turb32_z_s0", "g2_trace_blob24_x_s0", "g5_trace_aux24_z"])
def test_slab_chain_equals_single_pass(orc, name):
    """Handing the plane form's state over on shared node planes reproduces the single pass bit for bit, however
    the planes are cut (the semantics of the reference's region loop, propagator.py:366-452)."""
    g = golden(name)
    x, ext, pd = g["x"], float(g["extent"]), str(g["pdir"])
    if "aux" in name:
        dom = orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), True, g["Te"], g["Z"], g["B"])
    else:
        dom = orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), bool(g["phaseshift"]))
    t_end, n = orc.default_t_end(ext), len(x)
    whole, steps = orc.trace_rk4(dom, g["s0"], (x[1] - x[0]) / orc.c, t_end, pd, "planes", 1)
    for cuts in ([0, n - 1], [0, 7, n - 1], [0, 1, 2, 11, n - 2, n - 1]):
        rec, total = None, 0
        for q, (lo, hi) in enumerate(zip(cuts[:-1], cuts[1:])):
            rec, st = orc.trace_slab(dom, t_end, pd, lo, hi, s0=g["s0"] if q == 0 else None, rec=rec, last=hi == n - 1)
            total += st
        assert np.array_equal(rec, whole) and total == steps, cuts


def test_slab_chain_lost_rays_are_nan(orc):
    g = golden("g2_trace_blob32_z_s0")
    x, ext = g["x"], float(g["extent"])
    dom = orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), True)
    s0 = g["s0"].copy()
    s0[5, :7] *= -1.0   # heading away: not a plane-form ray
    s0[2, 7:9] = 0.0    # launched inside the volume
    rec, _ = orc.trace_slab(dom, orc.default_t_end(ext), "z", 0, 10, s0=s0)
    assert np.isnan(rec[:9, :9]).all() and not np.isnan(rec[:, 9:]).any() and np.array_equal(rec[9], np.arange(s0.shape[1]))
    sf, _ = orc.trace_slab(dom, orc.default_t_end(ext), "z", 10, len(x) - 1, rec=rec, last=True)
    whole, _ = orc.trace_rk4(dom, s0, (x[1] - x[0]) / orc.c, orc.default_t_end(ext), "z", "planes", 1)
    assert np.isnan(sf[:, :9]).all() and np.array_equal(sf[:, 9:], whole[:, 9:])


# ---------------------------------------------------------------- BASELINE.json configs[0], end to end
def _c1_inputs(g):
    """1e4 rays (np.random.seed(0), circular beam 4 mm, 5e-5 rad) and the 64^3 analytic Gaussian blob of config C1."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), ".."))
    from synthpy_amd import _beam  # the host-side ray draw (bit-exact against the reference: test_abi_and_host.py)

    n, ext = int(g["n"]), float(g["extent"])
    x = np.linspace(-ext, ext, n)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij", sparse=True)
    ne = 1e25 * np.exp(-(X ** 2 + Y ** 2 + Z ** 2) / (1.5e-3) ** 2)
    np.random.seed(int(g["seed"]))
    N = int(g["N"])
    t = 2 * np.pi * np.random.rand(N)
    u = np.random.rand(N) + np.random.rand(N)
    u[u > 1] = 2 - u[u > 1]
    phi = np.pi * np.random.rand(N)
    chi = float(g["divergence"]) * np.random.randn(N)
    bs = float(g["beam_size"])
    s0 = _beam.assemble(bs * u * np.cos(t), bs * u * np.sin(t), chi, phi, ext, "z")
    assert np.array_equal(s0[:, :8], g["s0_head"])
    return x, ne, s0


def test_config_c1_end_to_end_vs_reference(orc):
    """1e4 rays x 64^3 Gaussian blob, two-lens shadowgraphy (BASELINE.json configs[0]).  Against the reference at
    rtol 1e-10 on its first 2000 rays: exit positions 1e-8 m, and the histogram EXACT.  Against the reference as
    shipped (RK45, rtol 1e-3) on all 1e4 rays: the difference is the reference's own integration error (it differs
    from its own tight run by more than we do), the images have the same total and differ in few pixels."""
    g = golden("g8_config1")
    x, ne, s0 = _c1_inputs(g)
    ext, M = float(g["extent"]), int(g["M"])
    dom = orc.Domain.from_ne(ne, x, x, x, float(g["lwl"]))
    sf, steps = orc.trace_rk4(dom, s0, (x[1] - x[0]) / orc.c, orc.default_t_end(ext), "z", "planes", 1)
    assert steps == (len(x) - 1) * s0.shape[1]
    rf, _ = orc.ray_to_jones(sf, ext, "z")
    assert np.max(np.abs(rf[0::2, :M] - g["rf_tight"][0::2])) <= 1e-8 and np.max(np.abs(rf[1::2, :M] - g["rf_tight"][1::2])) <= 1e-6
    r, _ = orc.optics(orc.optics(rf, [(orc.SCALE, 1e3)])[0], orc.chain_shadow_two())
    H = orc.histogram(r, bin_scale=10)
    r_t, _ = orc.optics(orc.optics(rf[:, :M], [(orc.SCALE, 1e3)])[0], orc.chain_shadow_two())
    assert np.array_equal(orc.histogram(r_t, bin_scale=10), g["H_tight"])
    own = np.max(np.abs(g["rf_default"][0::2, :M].astype(np.float64) - g["rf_tight"][0::2]))
    ours = np.max(np.abs(rf[0::2, :M] - g["rf_tight"][0::2]))
    assert ours < own / 10
    Hd = g["H_default"].astype(np.int64)
    assert H.sum() == Hd.sum() == s0.shape[1] and np.abs(H.astype(np.int64) - Hd).sum() <= 0.02 * Hd.sum()


# ---------------------------------------------------------------- solve_at_depth: the trace stopped inside the volume
@pytest.mark.parametrize("name", ["g9_solve_at_depth_z", "g9_solve_at_depth_x"])
def test_solve_at_depth_vs_reference(orc, name):
    """full_solver.py:405-425: integrate for the time z/c only.  Every ray is still inside the volume then, so none is
    a plane-form ray: all of them take the time-stepping form with located faces.  Against the reference's RHS integrated
    tightly over the same time, and inside the error the reference's own default run shows."""
    g = golden(name)
    x, ext, pdir = g["x"], float(g["extent"]), str(g["pdir"])
    dom = orc.Domain.from_ne(g["ne"], x, x, x, float(g["lwl"]), phaseshift=True)
    sf, steps = orc.trace_rk4(dom, g["s0"], float(x[1] - x[0]) / orc.c, float(g["depth"]) / orc.c, pdir, "planes", 1)
    st = g["sf_tight"]
    a = "xyz".index(pdir)
    assert np.all(np.abs(st[a]) < ext)  # stopped inside
    assert np.max(np.abs(sf[:3] - st[:3])) <= 1e-8 and np.max(np.abs(sf[3:6] - st[3:6])) / orc.c <= 2e-5
    assert np.max(np.abs(sf[7] - st[7])) <= 2e-4 * max(1.0, np.max(np.abs(st[7])))
    rf, _ = orc.ray_to_jones(sf, ext, pdir, "legacy")
    own = np.max(np.abs(g["rf_default"][0::2] - g["rf_tight"][0::2]))
    assert np.max(np.abs(rf[0::2] - g["rf_tight"][0::2])) <= max(1e-8, own)
