#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF in this container.

Imports the reference's legacy NumPy/SciPy path read-only from
/root/reference/src/solvers-legacy (full_solver.py, rtm_solver.py) and
/root/reference/src/field_generator/gaussian3D.py, feeds it small seeded
inputs and stores inputs + the reference's outputs.  Nothing of the reference's
source is copied: a fixture is data.  The reference does not exist on the GPU
box, so the committed .npz files are what the tests there compare against.

Shims applied in this harness (never edits to the reference):
  * full_solver.omega_pe: the method is defined in the class body without
    `self` (full_solver.py:236-239) and is looked up as a module global at
    :252,:273 -> NameError as shipped when phaseshift=True.  The harness binds
    the module global to the same one-line body.
  * init_beam draws from the global np.random stream -> seeded here.
  * rtm_solver apertures mutate their argument -> copies are passed.

Run:  python oracle/make_golden.py        (writes tests/golden/)
"""
import io
import os
import sys
import contextlib

import numpy as np
import scipy
from scipy.integrate import solve_ivp

sys.dont_write_bytecode = True
REF = "/root/reference/src"
sys.path.insert(0, os.path.join(REF, "solvers-legacy"))
sys.path.insert(0, os.path.join(REF, "field_generator"))

import full_solver as fs  # noqa: E402  (the reference)
import rtm_solver as rtm  # noqa: E402  (the reference)
import gaussian3D as g3  # noqa: E402  (the reference)

fs.omega_pe = lambda ne: 5.64e4 * np.sqrt(ne)  # shim, see docstring

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
os.makedirs(OUT, exist_ok=True)
VERS = np.array([np.__version__, scipy.__version__])
LWL = 1064e-9


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, versions=VERS, **arrs)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def make_domain(n, extent, kind, phaseshift=False, seed=1234):
    """Reference ScalarDomain with one of the small test volumes."""
    if isinstance(n, int):
        n = (n, n, n)
    x = np.linspace(-extent, extent, n[0])
    y = np.linspace(-extent, extent, n[1])
    z = np.linspace(-extent, extent, n[2])
    dom = fs.ScalarDomain(x, y, z, extent, phaseshift=phaseshift)
    if kind == "null":
        dom.test_null()
    elif kind == "slab":
        dom.test_slab(s=1, n_e0=2e23)
    elif kind == "blob":
        # analytic Gaussian blob (profile family of minimal_solver.py:192-201), BASELINE config 1
        dom.external_ne(1e25 * np.exp(-(dom.XX ** 2 + dom.YY ** 2 + dom.ZZ ** 2) / (1.5e-3) ** 2))
    elif kind == "turb":
        assert n[0] == n[1] == n[2] and n[0] % 2 == 0
        np.random.seed(seed)
        field = quiet(g3.gaussian3D(lambda k: k ** (-11 / 3)).domain_fft, 1.0, 0.01 * (256 / n[0]), 5, n[0] // 2, 1.0)
        dom.external_ne(1e25 + 9e24 * field)
    else:
        raise ValueError(kind)
    dom.calc_dndr(LWL)
    return dom


# --------------------------------------------------------------------------
# G0: field generator (inputs of the benchmark volumes): domain_fft, seeded
# --------------------------------------------------------------------------
def g0_field():
    np.random.seed(1234)
    f = quiet(g3.gaussian3D(lambda k: k ** (-11 / 3)).domain_fft, 1.0, 0.16, 5, 8, 1.0)
    save("g0_domain_fft", seed=1234, l_max=1.0, l_min=0.16, extent=5, res=8, factor=1.0, field=f)
    # integratedPy.npy recipe (evaluation/sergio_testing/notebook.ipynb cells 7-8): test_linear_cos line integral
    x = np.linspace(-5e-3, 5e-3, 20)
    y = np.linspace(-5e-3, 5e-3, 50)
    z = np.linspace(-5e-3, 5e-3, 10)
    d = fs.ScalarDomain(x, y, z, 5e-3)
    d.test_linear_cos(s1=-1, s2=1, n_e0=1e26, Ly=5e-3)
    lc = d.ne.copy()
    d.test_exponential_cos()
    save("g0_profiles", x=x, y=y, z=z, extent=5e-3, linear_cos=lc, linear_cos_sum=lc.sum(axis=2), exponential_cos=d.ne)


def g0_beams():
    """init_beam for every working beam_type / probing direction, seeded (full_solver.py:547-835)."""
    out = {}
    for bt, size in (("circular", 3e-3), ("square", 2e-3), ("rectangular", (1e-3, 2e-3)), ("linear", 4e-3)):
        for pd in "xyz":
            np.random.seed(3)
            out[f"{bt}_{pd}"] = quiet(fs.init_beam, 40, size, 1e-4, 5e-3, bt, probing_direction=pd)
    save("g0_beams", seed=3, Np=40, divergence=1e-4, ne_extent=5e-3, **out)


# --------------------------------------------------------------------------
# G1: calc_dndr (A1) + interpolation (A4) + RHS (A3/A5)
# --------------------------------------------------------------------------
def g10_reference_pvti_header():
    """The one VTK file the reference tree holds: evaluation/sergio_testing/python_cube.pvti, the parallel header that
    export_pvti (src/utils/handle_filetypes.py:72-84) wrote for a 100 x 1000 x 100 cube (its piece file is not in the
    tree).  A data file, copied byte for byte: the reader is pinned on a header the repo did not write."""
    import shutil

    src = os.path.join(os.path.dirname(REF), "evaluation", "sergio_testing", "python_cube.pvti")
    shutil.copyfile(src, os.path.join(OUT, "python_cube.pvti"))
    print("python_cube.pvti", os.path.getsize(src), "bytes")


def g1_fields():
    rng = np.random.default_rng(7)
    for tag, n, ext in (("a", (12, 10, 9), 4e-3), ("b", (17, 9, 5), 1.0), ("c", (8, 8, 8), 5e-3), ("u", (5, 5, 5), 1.0)):
        x = np.linspace(-ext, ext, n[0])
        y = np.linspace(-ext, ext, n[1])
        z = np.linspace(-ext, ext, n[2])
        ne = 1e25 * (1 + 0.5 * rng.standard_normal(n)).clip(0.05)
        dom = fs.ScalarDomain(x, y, z, ext, phaseshift=True)
        dom.external_ne(ne)
        dom.calc_dndr(LWL)
        # query points: interior, exactly on nodes / faces, just outside, far outside, NaN
        P = rng.uniform(-1.2 * ext, 1.2 * ext, (400, 3))
        nodes = np.stack([np.float64(dom.x)[rng.integers(0, n[0], 60)], np.float64(dom.y)[rng.integers(0, n[1], 60)],
                          np.float64(dom.z)[rng.integers(0, n[2], 60)]], 1)
        faces = rng.uniform(-ext, ext, (40, 3))
        faces[:10, 0] = np.float64(dom.x[0]); faces[10:20, 0] = np.float64(dom.x[-1])
        faces[20:30, 2] = np.float64(dom.z[-1]); faces[30:40, 1] = np.float64(dom.y[0])
        outside = rng.uniform(-ext, ext, (6, 3)); outside[:3, 2] = -ext; outside[3:, 2] = ext  # float64 +-ext vs float32 node
        nan = np.array([[np.nan, 0, 0]])
        P = np.concatenate([P, nodes, faces, outside, nan])
        grad = dom.dndr(P.T.copy())
        nref = dom.n_refrac()
        s = np.zeros((9, len(P)))
        s[:3] = P.T
        s[3:6] = fs.c * rng.standard_normal((3, len(P)))
        s[6] = 1.0
        s[7] = rng.uniform(0, 5, len(P))
        ds = quiet(fs.dsdt, 0.0, s.flatten(), dom).reshape(9, -1)
        save(f"g1_fields_{tag}", x=x, y=y, z=z, extent=ext, lwl=LWL, ne=ne, omega=dom.omega, dndx=dom.dndx,
             dndy=dom.dndy, dndz=dom.dndz, nref=nref, pts=P, grad=grad, s=s, dsdt=ds)


# --------------------------------------------------------------------------
# G2: trace (A2) + back-projection (A6): default and tight tolerance
# --------------------------------------------------------------------------
def g2_trace():
    ext = 5e-3
    for kind, n, N, ph, pdir in (("null", 16, 64, False, "z"), ("slab", 16, 128, False, "z"), ("blob", 32, 256, True, "z"),
                                 ("turb", 32, 256, True, "z"), ("blob", 24, 96, False, "x"), ("blob", 24, 96, True, "y")):
        dom = make_domain(n, ext, kind, phaseshift=ph)
        dom.probing_direction = pdir
        for seed in ((0, 1) if kind in ("blob", "turb") and pdir == "z" else (0,)):
            np.random.seed(seed)
            s0 = fs.init_beam(N, 3.5e-3, 5e-5 if kind != "slab" else 0.0, ext, "circular", probing_direction=pdir)
            rf_d, Jf_d = quiet(dom.solve, s0.copy(), return_E=True)
            sf_d = dom.sf.copy()
            t_end = np.sqrt(8.0) * ext / fs.c
            sol = solve_ivp(lambda t, yv: fs.dsdt(t, yv, dom), [0, t_end], s0.flatten(), t_eval=[0, t_end], rtol=1e-10, atol=1e-12)
            sf_t = sol.y[:, -1].reshape(9, N)
            rf_t, Jf_t = fs.ray_to_Jonesvector(sf_t, ext, probing_direction=pdir)
            save(f"g2_trace_{kind}{n}_{pdir}_s{seed}", kind=kind, n=n, extent=ext, lwl=LWL, phaseshift=ph, pdir=pdir, seed=seed,
                 x=np.linspace(-ext, ext, n), ne=np.asarray(dom.ne, np.float64), s0=s0, sf_default=sf_d, rf_default=rf_d,
                 Jf_default=Jf_d, sf_tight=sf_t, rf_tight=rf_t, Jf_tight=Jf_t, nfev_default=0, nfev_tight=sol.nfev)


# --------------------------------------------------------------------------
# G3: optics chains (A7/A8) + histogram (A9);  G4: interferometry (A10)
# --------------------------------------------------------------------------
def g3_optics():
    rng = np.random.default_rng(11)
    N = 4000
    # synthetic exit-plane rays spanning every mask: positions +-8 mm, angles up to ~0.1 rad (some beyond the lens radius)
    rf = np.zeros((4, N))
    rf[0] = rng.uniform(-8e-3, 8e-3, N)
    rf[2] = rng.uniform(-8e-3, 8e-3, N)
    rf[1] = 4e-3 * rng.standard_normal(N)
    rf[3] = 4e-3 * rng.standard_normal(N)
    rf[1, :300] *= 25
    rf[3, :300] *= 25
    rf[:, 3990:] = np.nan
    ph = rng.uniform(0, 300, N)
    E = np.zeros((2, N), complex)
    E[1] = np.cos(ph) + 1j * np.sin(ph)
    E[0] = 0.1 * (np.cos(2 * ph) + 1j * np.sin(2 * ph))
    out = dict(rf=rf, E=E)

    def run(cls, method, **kw):
        d = cls(rf.copy(), **kw.pop("init", {}))
        getattr(d, method)(**kw)
        return d

    for name, cls, method, kw in (("shadow_single", rtm.Shadowgraphy, "single_lens_solve", {}),
                                  ("shadow_two", rtm.Shadowgraphy, "two_lens_solve", {}),
                                  ("shadow_two_fp", rtm.Shadowgraphy, "two_lens_solve", {"init": dict(focal_plane=3.0, L=350, R=20)}),
                                  ("schlieren_df", rtm.Schlieren, "DF_solve", {}),
                                  ("schlieren_lf", rtm.Schlieren, "LF_solve", {"R": 2}),
                                  ("refracto", rtm.Refractometry, "incoherent_solve", {})):
        d = run(cls, method, **kw)
        out[name + "_rf"] = d.rf.copy()
        d.histogram(bin_scale=10)
        out[name + "_H10"] = d.H.copy()
        d.histogram(bin_scale=1, pix_x=64, pix_y=48)
        out[name + "_H64x48"] = d.H.copy()
    # histogram edge semantics: values exactly on edges / the last edge / outside
    d = rtm.Shadowgraphy(rf.copy())
    xe = np.linspace(-9, 9, 345)
    ye = np.linspace(-6.75, 6.75, 258)
    pts = np.zeros((4, 700))
    pts[0, :345] = xe; pts[2, :345] = 0.1
    pts[2, 345:603] = ye; pts[0, 345:603] = -0.3
    pts[0, 603:] = rng.uniform(-9.5, 9.5, 97); pts[2, 603:] = rng.uniform(-7, 7, 97)
    pts[0, 690] = 9.0; pts[2, 690] = 6.75; pts[0, 691] = -9.0; pts[2, 691] = -6.75
    pts[0, 692] = np.nextafter(9.0, 10); pts[2, 693] = np.nextafter(-6.75, -7)
    d.rf = pts
    d.histogram(bin_scale=10)
    out["edge_pts"] = pts
    out["edge_H10"] = d.H.copy()
    out["xedges"] = d.xedges
    out["yedges"] = d.yedges

    # G4 interferometry, legacy two_lens_solve(wl) + interferogram; complex field after the chain
    it = rtm.Interferometry(rf.copy(), E=E.copy())
    it.two_lens_solve(wl=532e-9)
    out["interf_rf"] = it.rf.copy()
    out["interf_rE"] = it.rE.copy()
    it.interferogram(bin_scale=10)
    out["interf_H10"] = it.H.copy()
    it.interferogram(bin_scale=1, pix_x=40, pix_y=30)
    out["interf_H40x30"] = it.H.copy()
    # digitize edge semantics of the interferogram (edges linspace(-9, 9, 344), (-6, 6, 257))
    it2 = rtm.Interferometry(rf.copy(), E=E.copy())
    xe = np.linspace(-18 // 2, 18 // 2, 344)
    pe = np.zeros((4, 400))
    pe[0, :344] = xe
    pe[2, :344] = 0.01
    pe[0, 344:] = rng.uniform(-9.2, 9.2, 56)
    pe[2, 344:] = rng.uniform(-6.3, 6.3, 56)
    pe[2, 350] = 6.0; pe[2, 351] = -6.0
    it2.rf = pe
    it2.rE = E[:, :400].copy()
    it2.interferogram(bin_scale=10)
    out["interf_edge_pts"] = pe
    out["interf_edge_H10"] = it2.H.copy()
    save("g3_optics", **out)


# --------------------------------------------------------------------------
# G5: the optional RHS terms (A3): inverse bremsstrahlung (amp), Faraday rotation (pol)
# --------------------------------------------------------------------------
def aux_domain(x, y, z, ext, ne, Te, Z, B, pdir="z"):
    dom = fs.ScalarDomain(x, y, z, ext, B_on=True, inv_brems=True, phaseshift=True, probing_direction=pdir)
    dom.external_ne(ne)
    dom.external_Te(Te)
    dom.external_Z(Z)
    dom.external_B(B)
    dom.calc_dndr(LWL)
    dom.set_up_interps()
    return dom


def g5_aux():
    rng = np.random.default_rng(23)
    n, ext = (12, 10, 9), 4e-3
    x, y, z = (np.linspace(-ext, ext, m) for m in n)
    ne = 1e25 * (1 + 0.5 * rng.standard_normal(n)).clip(0.05)
    Te = rng.uniform(0.2, 300.0, n)  # values below Te_min = 1 are clamped by external_Te
    Z = rng.uniform(1.0, 8.0, n)
    B = rng.standard_normal(n + (3,)) * 5.0
    dom = aux_domain(x, y, z, ext, ne, Te, Z, B)
    P = np.concatenate([rng.uniform(-1.2 * ext, 1.2 * ext, (300, 3)),
                        np.stack([np.float64(dom.x)[rng.integers(0, n[0], 40)], np.float64(dom.y)[rng.integers(0, n[1], 40)],
                                  np.float64(dom.z)[rng.integers(0, n[2], 40)]], 1), np.array([[np.nan, 0, 0]])])
    s = np.zeros((9, len(P)))
    s[:3] = P.T
    s[3:6] = fs.c * rng.standard_normal((3, len(P)))
    s[6] = rng.uniform(0.5, 1.5, len(P))
    s[7] = rng.uniform(0, 5, len(P))
    s[8] = rng.uniform(-1, 1, len(P))
    ds = quiet(fs.dsdt, 0.0, s.flatten(), dom).reshape(9, -1)
    save("g5_fields_aux", x=x, y=y, z=z, extent=ext, lwl=LWL, ne=ne, Te_in=Te, Te=dom.Te, Z=Z, B=B, omega=dom.omega,
         verdet=dom.VerdetConst, kappa=dom.kappa(), pts=P, s=s, dsdt=ds, kappa_at=dom.atten(P.T.copy()),
         ne_at=dom.get_ne(P.T.copy()), B_at=dom.get_B(P.T.copy()))

    ext = 5e-3
    for n1, pdir, N in ((24, "z", 96), (20, "x", 64)):
        x = np.linspace(-ext, ext, n1)
        XX, YY, ZZ = np.meshgrid(x, x, x, indexing="ij")
        r2 = XX ** 2 + YY ** 2 + ZZ ** 2
        ne = 1e25 * np.exp(-r2 / (1.5e-3) ** 2) + 5e23
        Te = 40.0 + 160.0 * np.exp(-r2 / (2.5e-3) ** 2)
        Z = 2.0 + 3.0 * np.exp(-r2 / (2e-3) ** 2)
        B = np.stack([3.0 * YY / ext, -2.0 * XX / ext + 1.0, 8.0 * (1 + ZZ / ext) * np.exp(-r2 / (3e-3) ** 2)], -1)
        dom = aux_domain(x, x, x, ext, ne, Te, Z, B, pdir)
        np.random.seed(5)
        s0 = fs.init_beam(N, 3.5e-3, 5e-5, ext, "circular", probing_direction=pdir)
        s0[8] = 0.1  # a non-zero initial polarisation angle
        rf_d, Jf_d = quiet(dom.solve, s0.copy(), return_E=True)
        sf_d = dom.sf.copy()
        t_end = np.sqrt(8.0) * ext / fs.c
        sol = solve_ivp(lambda t, yv: fs.dsdt(t, yv, dom), [0, t_end], s0.flatten(), t_eval=[0, t_end], rtol=1e-10, atol=1e-12)
        sf_t = sol.y[:, -1].reshape(9, N)
        rf_t, Jf_t = fs.ray_to_Jonesvector(sf_t, ext, probing_direction=pdir)
        save(f"g5_trace_aux{n1}_{pdir}", n=n1, extent=ext, lwl=LWL, pdir=pdir, seed=5, x=x, ne=ne, Te=Te, Z=Z, B=B, s0=s0,
             sf_default=sf_d, rf_default=rf_d, Jf_default=Jf_d, sf_tight=sf_t, rf_tight=rf_t, Jf_tight=Jf_t)


# --------------------------------------------------------------------------
# G6: coherent refractometer (rtm_solver.py:288-369, seeded speckle) and knife-edge schlieren (:119-135)
# --------------------------------------------------------------------------
def g6_optics_extra():
    rng = np.random.default_rng(31)
    N = 3000
    rf = np.zeros((4, N))
    rf[0] = rng.uniform(-8e-3, 8e-3, N)
    rf[2] = rng.uniform(-8e-3, 8e-3, N)
    rf[1] = 4e-3 * rng.standard_normal(N)
    rf[3] = 4e-3 * rng.standard_normal(N)
    rf[1, :200] *= 25
    rf[:, 2990:] = np.nan
    ph = rng.uniform(0, 300, N)
    E = np.zeros((2, N), complex)
    E[1] = np.cos(ph) + 1j * np.sin(ph)
    E[0] = 0.2 * (np.cos(3 * ph) - 1j * np.sin(3 * ph))
    out = dict(rf=rf, E=E)
    d = rtm.Refractometry(rf.copy(), E=E.copy(), focal_plane=2.0)
    d.coherent_solve(wl=1064e-9)
    out["coh_rf"], out["coh_rE"] = d.rf.copy(), d.rE.copy()
    np.random.seed(9)
    d.refractogram(bin_scale=10)
    out["coh_H10_seed9"] = d.H.copy()
    r_mm = rtm.m_to_mm(rf.copy())
    for k, (off, ax, dr) in enumerate(((0.5, "y", 1), (-1.0, "x", -1), (0.0, "x", 1), (2.0, "y", -1))):
        out[f"knife{k}"] = rtm.knife_edge(r_mm.copy(), off, ax, dr)
    out["knife_args"] = np.array([[0.5, 2, 1], [-1.0, 0, -1], [0.0, 0, 1], [2.0, 2, -1]])
    save("g6_optics_extra", **out)


# --------------------------------------------------------------------------
# G7: analysis of the images after the path: radial_2Dspectrum (src/utils/power_spectrum.py:372-421)
# --------------------------------------------------------------------------
def g7_spectrum():
    sys.path.insert(0, os.path.join(REF, "utils"))
    import power_spectrum as ps  # noqa: E402  (the reference)

    rng = np.random.default_rng(41)
    out = {}
    for tag, n, lx, ly in (("a", 96, 18.0, 18.0), ("b", 128, 10.0, 10.0), ("c", 75, 13.5, 13.5)):
        xx = np.linspace(0, 1, n)
        img = 40 + 25 * np.sin(14 * xx)[:, None] * np.cos(9 * xx)[None, :] + rng.poisson(30, (n, n))
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                kn, kc, sp = ps.radial_2Dspectrum(img.astype(float), lx, ly)
                _, _, sp_s = ps.radial_2Dspectrum(img.astype(float), lx, ly, smooth=True)
        out.update({f"img_{tag}": img.astype(float), f"l_{tag}": np.array([lx, ly]), f"kn_{tag}": kn, f"kc_{tag}": kc,
                    f"sp_{tag}": sp, f"sps_{tag}": sp_s})
    save("g7_spectrum", **out)


# --------------------------------------------------------------------------
# G8: BASELINE.json configs[0] end to end, the reference AS SHIPPED: 1e4 rays through a 64^3 analytic Gaussian blob,
#     shadowgraphy only (solve_ivp RK45 at its default rtol 1e-3), and the same rays at rtol 1e-10
# --------------------------------------------------------------------------
def g8_config1():
    ext, n, N = 5e-3, 64, 10000
    dom = make_domain(n, ext, "blob")
    np.random.seed(0)
    s0 = fs.init_beam(N, 4e-3, 5e-5, ext, "circular", probing_direction="z")
    rf_d = quiet(dom.solve, s0.copy())
    sh = rtm.Shadowgraphy(rf_d.copy())
    quiet(sh.two_lens_solve)
    sh.histogram(bin_scale=10)
    H_d = sh.H.copy()
    t_end = np.sqrt(8.0) * ext / fs.c
    M = 2000  # the tight run on the first M rays (the RHS costs the same per ray; 1e4 at rtol 1e-10 takes minutes)
    sol = solve_ivp(lambda t, yv: fs.dsdt(t, yv, dom), [0, t_end], s0[:, :M].flatten(), t_eval=[0, t_end], rtol=1e-10, atol=1e-12)
    rf_t, _ = fs.ray_to_Jonesvector(sol.y[:, -1].reshape(9, M), ext, probing_direction="z")
    sh2 = rtm.Shadowgraphy(rf_t.copy())
    quiet(sh2.two_lens_solve)
    sh2.histogram(bin_scale=10)
    save("g8_config1", n=n, extent=ext, lwl=LWL, seed=0, N=N, M=M, beam_size=4e-3, divergence=5e-5, rf_default=rf_d.astype(np.float32),
         H_default=H_d.astype(np.uint16), rf_tight=rf_t, H_tight=sh2.H.astype(np.uint16), s0_head=s0[:, :8])


# --------------------------------------------------------------------------
# G9: solve_at_depth (full_solver.py:405-425): the trace stopped INSIDE the volume (length z of flight), as the
# reference runs it (default tolerances) and with its RHS integrated tightly over the same time
# --------------------------------------------------------------------------
def g9_solve_at_depth():
    ext, n, N = 5e-3, 32, 128
    for pdir, depth in (("z", 1.0 * ext), ("x", 1.4 * ext)):
        dom = make_domain(n, ext, "blob", phaseshift=True)
        dom.probing_direction = pdir
        np.random.seed(4)
        s0 = fs.init_beam(N, 3.5e-3, 5e-5, ext, "circular", probing_direction=pdir)
        rf_d = quiet(dom.solve_at_depth, s0.copy(), depth)
        sf_d = dom.sf.copy()
        t_end = depth / fs.c
        sol = solve_ivp(lambda t, yv: fs.dsdt(t, yv, dom), [0, t_end], s0.flatten(), t_eval=[0, t_end], rtol=1e-10, atol=1e-12)
        sf_t = sol.y[:, -1].reshape(9, N)
        rf_t, _ = fs.ray_to_Jonesvector(sf_t, ext, probing_direction=pdir)
        save(f"g9_solve_at_depth_{pdir}", n=n, extent=ext, lwl=LWL, pdir=pdir, depth=depth, x=np.linspace(-ext, ext, n),
             ne=np.asarray(dom.ne, np.float64), s0=s0, sf_default=sf_d, rf_default=rf_d, sf_tight=sf_t, rf_tight=rf_t)


# --------------------------------------------------------------------------
# G11: the set-up of examples/notebooks/test_SynthRayTracer.ipynb cells 4-15 -- the one place the reference stores end-to-end
#      numbers for the JAX-generation API (how many of 300000 rays survive four diagnostics) -- run through the reference's OWN
#      legacy solver and diagnostics: which rays survive, at the default tolerance and with the RHS integrated tightly.  The
#      JAX run behind the notebook's numbers (float32, Tsit5, PIDController(rtol=1, atol=1e-5)) cannot be repeated here (no
#      jax): this fixture says what the reference's RHS gives for the same volume and beam when it is integrated accurately.
# --------------------------------------------------------------------------
def g11_notebook():
    n, ex, ez, N, M = 128, 5e-3, 10e-3, 20000, 4000
    x, z = np.linspace(-ex, ex, n), np.linspace(-ez, ez, n)
    dom = fs.ScalarDomain(x, x, z, ez)
    dom.test_exponential_cos(n_e0=1e24, Ly=1e-3, s=2e-3)  # n_e0 * 10^(x/s) * (1 + cos(2 pi y / Ly)), full_solver.py:158-167
    dom.calc_dndr(LWL)
    np.random.seed(0)
    s0 = quiet(fs.init_beam, N, 5e-3, 5e-5, ez, "circular", probing_direction="z")
    rf_d = quiet(dom.solve, s0.copy())
    t_end = np.sqrt(8.0) * ez / fs.c
    sol = solve_ivp(lambda t, yv: fs.dsdt(t, yv, dom), [0, t_end], s0[:, :M].flatten(), t_eval=[0, t_end], rtol=1e-9, atol=1e-12)
    rf_t, _ = fs.ray_to_Jonesvector(sol.y[:, -1].reshape(9, M), ez, probing_direction="z")

    def survivors(rf):
        out = {}
        for key, cls, kw, solve in (("refractometry", rtm.Refractometry, {}, "incoherent_solve"),
                                    ("refractometry_L50", rtm.Refractometry, {"L": 50}, "incoherent_solve"),
                                    ("shadow_single", rtm.Shadowgraphy, {}, "single_lens_solve"),
                                    ("schlieren_DF", rtm.Schlieren, {}, "DF_solve")):
            o = cls(rf.copy(), **kw)
            getattr(o, solve)()
            out[key] = np.packbits(~np.isnan(o.rf[0]) & ~np.isnan(o.rf[2]))
        return out

    sd, st = survivors(rf_d), survivors(rf_t)
    save("g11_notebook", n=n, extent_x=ex, extent_z=ez, lwl=LWL, N=N, M=M, seed=0, beam_size=5e-3, divergence=5e-5,
         rf_default=rf_d, rf_tight=rf_t, **{f"kept_default_{k}": v for k, v in sd.items()}, **{f"kept_tight_{k}": v for k, v in st.items()})
    for k in sd:
        print(k, "default", int(np.unpackbits(sd[k])[:N].sum()), "of", N, "| tight", int(np.unpackbits(st[k])[:M].sum()), "of", M,
              "| default on the same", M, "rays", int(np.unpackbits(sd[k])[:M].sum()))


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    g0_field()
    g0_beams()
    g1_fields()
    g2_trace()
    g3_optics()
    g5_aux()
    g6_optics_extra()
    g7_spectrum()
    g8_config1()
    g9_solve_at_depth()
    g10_reference_pvti_header()
    g11_notebook()
