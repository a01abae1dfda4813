"""The reference's own call sequence reaches the fused device path.

A synthPy caller writes (examples/jobs/run_scripts/pvti_trace_mpi.py:111-131, src/solvers-legacy/rtm_solver.py:142-178,
205-214, 376-453; JAX generation: examples/notebooks/test_SynthRayTracer.ipynb cells 4-15)

    rf = field.solve(ss);  sh = rtm.Shadowgraphy(rf);  sh.two_lens_solve();  sh.histogram()

Through the mirror classes that sequence now deposits from the bundle solve() left in HBM (synthpy_amd/resident.py ->
sr_rays_deposit: k_deposit, the LDS-tiled kernel bench.py times).  Checked here, on BASELINE configs[0] (C1: 1e4 rays x 64^3
Gaussian blob): the images of that flow == the images of the host-array flow (the same classes fed copies of the arrays:
sr_optics + sr_hist2d / sr_interferogram) == the oracle's from the SAME s0 -- counts integer for integer, interferograms to
1e-9 of their maximum between the two GPU flows (the atomics sum in another order) and 1e-5 against the oracle; arrays a caller
has changed fall back to the host flow; .r0 / .rf / .rE read from a device-backed object are the host flow's arrays bit for bit.
"""
import pickle

import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from synthpy_amd import engine

    engine.init(0)
    return engine


@pytest.fixture(scope="module")
def c1():
    from test_oracle_golden import _c1_inputs

    g = golden("g8_config1")
    x, ne, s0 = _c1_inputs(g)
    return x, ne, s0, float(g["extent"]), float(g["lwl"])


@pytest.fixture(scope="module")
def c1_oracle(orc, c1):
    """The oracle's exit rays and Jones vectors from s0, with the phase integral."""
    x, ne, s0, ext, lwl = c1
    dom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=True)
    sf, _ = orc.trace_rk4(dom, s0, (x[1] - x[0]) / orc.c, orc.default_t_end(ext), "z", "planes", 1)
    return orc.ray_to_jones(sf, ext, "z", "legacy")


def _legacy_solve(c1, return_E=True):
    from synthpy_amd.solvers_legacy import full_solver as fs

    x, ne, s0, ext, lwl = c1
    dom = fs.ScalarDomain(x, x, x, ext, phaseshift=True)
    dom.external_ne(ne)
    dom.calc_dndr(lwl)
    return dom, dom.solve(s0, return_E=return_E)


COUNTS = [("Shadowgraphy", "two_lens_solve", "chain_shadow_two"), ("Shadowgraphy", "single_lens_solve", "chain_shadow_single"),
          ("Schlieren", "DF_solve", "chain_schlieren"), ("Refractometry", "incoherent_solve", "chain_refractometry")]


@pytest.mark.parametrize("bin_scale", [1, 10])
def test_legacy_counts_flow_deposits_from_hbm_and_equals_host_flow_and_oracle(eng, orc, c1, c1_oracle, bin_scale):
    from synthpy_amd.solvers_legacy import rtm_solver as rtm

    dom, (rf, Jf) = _legacy_solve(c1)
    rf_o, _ = c1_oracle
    for cls, solve, chain in COUNTS:
        dev = getattr(rtm, cls)(rf)
        assert dev.on_device, "the diagnostic did not find the bundle solve() left in HBM"
        getattr(dev, solve)()
        dev.histogram(bin_scale=bin_scale)
        assert dev.on_device and dev._rf is None, "histogram() brought the rays to the host"
        host = getattr(rtm, cls)(rf.copy())  # not the array solve() returned: the host-array flow
        assert not host.on_device
        getattr(host, solve)()
        host.histogram(bin_scale=bin_scale)
        assert dev.H.dtype == np.float64 and dev.H.shape == host.H.shape
        assert np.array_equal(dev.H, host.H), (cls, solve)
        r_o, _ = orc.optics(orc.m_to_mm(rf_o), getattr(orc, chain)())
        H_o = orc.histogram(r_o, bin_scale=bin_scale)
        assert dev.H.sum() == H_o.sum() > 0 and np.array_equal(dev.H, H_o), (cls, solve, "oracle from s0")
        assert np.array_equal(dev.xedges, host.xedges) and np.array_equal(dev.yedges, host.yedges)
        # the arrays a caller may read: formed on the device, the host flow's bit for bit (NaN columns included)
        assert np.array_equal(dev.rf, host.rf, equal_nan=True) and np.array_equal(dev.r0, host.r0, equal_nan=True)
        assert dev.on_device  # reading does not end the device flow
        dev.histogram(bin_scale=bin_scale, clear_mem=True)
        assert np.array_equal(dev.H, host.H) and dev.rf is None and dev.r0 is None and not dev.on_device


def test_legacy_interferometry_flow_from_hbm(eng, orc, c1, c1_oracle):
    from synthpy_amd.solvers_legacy import rtm_solver as rtm

    x, ne, s0, ext, lwl = c1
    dom, (rf, Jf) = _legacy_solve(c1)
    rf_o, Jf_o = c1_oracle
    for bs in (10, 1):
        dev = rtm.Interferometry(rf, E=Jf)
        assert dev.on_device
        dev.two_lens_solve(wl=lwl)
        dev.interferogram(bin_scale=bs)
        assert dev.on_device and dev._rf is None
        host = rtm.Interferometry(rf.copy(), E=Jf.copy())
        assert not host.on_device
        host.two_lens_solve(wl=lwl)
        host.interferogram(bin_scale=bs)
        assert dev.H.shape == host.H.shape and host.H.max() > 0
        assert np.max(np.abs(dev.H - host.H)) <= 1e-9 * host.H.max()
        r_o, E_o = orc.optics(orc.m_to_mm(rf_o), orc.chain_shadow_two(), E=Jf_o, kwave=2 * np.pi / lwl)
        H_o = orc.interferogram(r_o, E_o, bin_scale=bs)
        assert np.max(np.abs(dev.H - H_o)) <= 1e-5 * H_o.max()
    assert np.array_equal(dev.rf, host.rf, equal_nan=True)
    # one rotation by the summed argument on the device path, one per leg on sr_optics' -- the same kernel function in both:
    assert np.array_equal(dev.rE, host.rE, equal_nan=True)
    # E that is not the Jf of that solve: host flow
    assert not rtm.Interferometry(rf, E=Jf.copy()).on_device
    # solve() without return_E, then a diagnostic that asks for the field with somebody else's E
    dom2, rf2 = _legacy_solve(c1, return_E=False)
    assert rtm.Shadowgraphy(rf2).on_device and not rtm.Interferometry(rf2, E=Jf).on_device


def test_changed_arrays_fall_back_to_the_host_flow(eng, orc, c1):
    from synthpy_amd import resident
    from synthpy_amd.solvers_legacy import rtm_solver as rtm

    dom, (rf, Jf) = _legacy_solve(c1)
    assert rtm.Shadowgraphy(rf).on_device
    rf[0:4:2, :] *= 1e3  # what pvti_trace_mpi.py:120 does to the array before handing it on
    sh = rtm.Shadowgraphy(rf)
    assert not sh.on_device
    sh.two_lens_solve()
    sh.histogram(bin_scale=10)
    r_o, _ = orc.optics(orc.m_to_mm(rf), orc.chain_shadow_two())
    assert np.array_equal(sh.H, orc.histogram(r_o, bin_scale=10))
    # a mask of a few columns, a shifted row: seen as well
    for change in (lambda a: a.__setitem__((slice(None), slice(0, a.shape[1] // 50)), np.nan), lambda a: a.__setitem__(2, a[2] + 1e-9)):
        dom, (rf, Jf) = _legacy_solve(c1)
        change(rf)
        assert not rtm.Schlieren(rf).on_device
    # arrays that never came from a solve()
    assert not rtm.Shadowgraphy(np.zeros((4, 100))).on_device
    # the whole-array guard (one download instead of the probes) sees a single changed element
    dom, (rf, Jf) = _legacy_solve(c1)
    np.asarray(rf)[1, 1234] += 1e-12  # through a base-class view: a writer the tracking cannot see, between the probes
    assert not resident.dirty(rf) and rtm.Shadowgraphy(rf).on_device
    old = resident.MODE
    try:
        resident.MODE = "full"
        assert not rtm.Shadowgraphy(rf).on_device
        dom, (rf, Jf) = _legacy_solve(c1)
        assert rtm.Shadowgraphy(rf).on_device
        resident.MODE = "0"
        assert not rtm.Shadowgraphy(rf).on_device
    finally:
        resident.MODE = old


def _edits():
    """(name, edit(rf)) -- the sparse edits a caller makes to the array solve() returned before building a diagnostic: each
    changes a handful of rays, far fewer than the sampled probes would notice (round 4's hole: 8192 probes per array)."""
    def col(a):
        a[:, 5] = np.nan

    def mask(a):
        m = np.zeros(a.shape[1], bool)
        m[np.random.default_rng(3).choice(a.shape[1], 10, replace=False)] = True
        a[0, m] = np.nan

    def copyto(a):
        b = np.array(a)
        b[:, 77] = np.nan
        np.copyto(a, b)

    def through_a_view(a):
        row = a[0]
        row[4321] = np.nan

    def inplace_on_a_slice(a):
        a[2, 100:103] += 2.0e-3

    def put(a):
        np.put(a, [a.shape[1] + 9], 0.3)  # rf[1, 9]: an angle

    return [("column", col), ("mask of 10 rays", mask), ("np.copyto", copyto), ("view", through_a_view),
            ("+= on a slice", inplace_on_a_slice), ("np.put", put)]


def test_sparse_edits_are_seen_and_the_edited_array_is_what_gets_binned(eng, orc):
    """VERDICT round 4, "the resident guard can return a stale image silently": rf edited in k rays was noticed with probability
    8e-4 * k.  Now every write through Python marks the array (resident.TrackedArray): the diagnostic takes the host path and
    bins the array it was GIVEN, as the reference does (rtm_solver.py:142-178).  1e6 rays, so that 10 edited rays are 1e-5 of
    them; H == the host-array flow's == the oracle's optics + histogram of the edited array, integer for integer."""
    from synthpy_amd import resident
    from synthpy_amd.solvers_legacy import full_solver as fs, rtm_solver as rtm

    n, ext, lwl, N = 48, 5e-3, 1064e-9, 1000000
    x = np.linspace(-ext, ext, n)
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij", sparse=True)
    ne = 1e25 * np.exp(-(X ** 2 + Y ** 2 + Z ** 2) / (1.5e-3) ** 2)
    np.random.seed(11)
    s0 = fs.init_beam(N, 4e-3, 5e-5, ext, "circular", "z")
    dom = fs.ScalarDomain(x, x, x, ext)
    dom.external_ne(ne)
    dom.calc_dndr(lwl)
    rf = dom.solve(s0)
    assert isinstance(rf, np.ndarray) and not resident.dirty(rf)
    clean = rtm.Shadowgraphy(rf)
    assert clean.on_device  # the unedited flow still deposits from HBM
    clean.two_lens_solve()
    clean.histogram(bin_scale=1)
    assert clean.on_device
    H_clean = clean.H
    for name, edit in _edits():
        rf = dom.solve(s0)
        edit(rf)
        assert resident.dirty(rf), name
        sh = rtm.Shadowgraphy(rf)
        assert sh.on_device is False, name
        sh.two_lens_solve()
        sh.histogram(bin_scale=1)
        host = rtm.Shadowgraphy(np.array(rf))  # a plain copy of the edited array: the host-array flow
        host.two_lens_solve()
        host.histogram(bin_scale=1)
        r_o, _ = orc.optics(orc.m_to_mm(np.array(rf)), orc.chain_shadow_two())
        H_o = orc.histogram(r_o, bin_scale=1)
        assert np.array_equal(sh.H, host.H) and np.array_equal(sh.H, H_o), name
        assert not np.array_equal(sh.H, H_clean), name  # and the edit does change the image: the stale one would have been wrong
    # reading does not mark: arithmetic, selections, copies, a histogram of the caller's own
    rf = dom.solve(s0)
    _ = rf * 1e3, rf[0][~np.isnan(rf[0])], rf.copy(), np.histogram2d(rf[0], rf[2], bins=8), rf[:, ::7].sum(), rf.T[5]
    assert not resident.dirty(rf) and rtm.Shadowgraphy(rf).on_device


def test_writes_after_construction_count_as_the_reference_reads_them(eng, orc, c1):
    """nothing was looked at again after the constructor.  The reference reads self.E / self.r0 / self.rf when
    *_solve() / histogram() / interferogram() run: legacy Interferometry has no interfere_ref_beam, so `it.E[1] += beam` before
    two_lens_solve() is how a caller adds fringes (rtm_solver.py:372-453)."""
    from synthpy_amd.solvers_legacy import rtm_solver as rtm

    x, ne, s0, ext, lwl = c1
    dom, (rf, Jf) = _legacy_solve(c1)
    beam = np.exp(1j * 2e3 * rf[0])

    def flow(it, edit_E):
        if edit_E:
            it.E[1] += beam
        it.two_lens_solve(wl=lwl)
        it.interferogram(bin_scale=10)
        return it.H

    dev = rtm.Interferometry(rf, E=Jf)
    assert dev.on_device
    H_dev = flow(dev, True)
    assert not dev.on_device  # E was written to after construction: the host arrays from there on
    host = rtm.Interferometry(np.array(rf), E=np.array(Jf) - np.vstack([0 * beam, beam]))  # the field as it was
    H_host = flow(host, True)
    assert np.max(np.abs(H_dev - H_host)) <= 1e-9 * H_host.max()
    dom, (rf, Jf) = _legacy_solve(c1)
    plain = rtm.Interferometry(rf, E=Jf)
    H_plain = flow(plain, False)
    assert plain.on_device and np.max(np.abs(H_dev - H_plain)) > 1e-3 * H_plain.max()  # the beam does show
    # E edited AFTER two_lens_solve(): rE was formed at solve time, the interferogram does not change (reference: the same)
    dom, (rf, Jf) = _legacy_solve(c1)
    late = rtm.Interferometry(rf, E=Jf)
    late.two_lens_solve(wl=lwl)
    late.E[1] += beam
    late.interferogram(bin_scale=10)
    assert np.max(np.abs(late.H - H_plain)) <= 1e-9 * H_plain.max()
    # r0 read from a device-backed object and edited before the solve: the edited rays are the ones traced through the optics
    dom, (rf, Jf) = _legacy_solve(c1)
    sh = rtm.Shadowgraphy(rf)
    r0 = sh.r0
    assert sh.on_device
    r0[:, :2000] = np.nan
    sh.two_lens_solve()
    sh.histogram(bin_scale=10)
    assert not sh.on_device
    ref = rtm.Shadowgraphy(np.array(rf))
    ref.r0[:, :2000] = np.nan
    ref.two_lens_solve()
    ref.histogram(bin_scale=10)
    assert np.array_equal(sh.H, ref.H) and sh.H.sum() < 0.9 * s0.shape[1]
    # the chain's output read and masked before histogram(): histogram() bins what self.rf holds now
    dom, (rf, Jf) = _legacy_solve(c1)
    sh = rtm.Shadowgraphy(rf)
    sh.two_lens_solve()
    sh.rf[:, ::2] = np.nan
    sh.histogram(bin_scale=10)
    ref = rtm.Shadowgraphy(np.array(rf))
    ref.two_lens_solve()
    ref.rf[:, ::2] = np.nan
    ref.histogram(bin_scale=10)
    assert np.array_equal(sh.H, ref.H)
    # E replaced outright
    dom, (rf, Jf) = _legacy_solve(c1)
    it = rtm.Interferometry(rf, E=Jf)
    it.E = np.array(Jf) * 0.5
    assert not it.on_device
    it.two_lens_solve(wl=lwl)
    it.interferogram(bin_scale=10)
    assert np.max(np.abs(it.H - 0.5 * H_plain)) <= 1e-9 * H_plain.max()


def test_simulator_rf_assigned_after_a_field_chain_keeps_the_chains_Jf(eng, c1):
    """ADVICE round 4 (diagnostics.py rf setter): d.rf = x after two_lens_solve() must leave Jf what the chain made of it."""
    from synthpy_amd.simulator import diagnostics as diag, domain as d, propagator as p

    x, ne, s0, ext, lwl = c1
    dom = d.ScalarDomain(2 * ext, len(x), phaseshift=True)
    dom.external_ne(ne)
    rf, Jf, _ = p.solve(s0, dom, ext, return_E=True, lwl=lwl)
    dev = diag.Interferometry(lwl, rf, Jf)
    host = diag.Interferometry(lwl, np.array(rf), np.array(Jf))
    assert dev.on_device and not host.on_device
    for it in (dev, host):
        it.two_lens_solve()
        it.rf = np.array(it.rf) * 1.0
    assert np.array_equal(dev.Jf, host.Jf, equal_nan=True) and not np.array_equal(dev.Jf, Jf)


def test_simulator_classes_see_writes_too(eng, orc, c1):
    """The JAX generation's mirror (src/simulator/diagnostics.py:269-379): propagator.solve() hands out the same write-tracked
    arrays; a masked rf takes the host path and gives the host-array flow's image, and an array the object handed out itself
    (.rf after a solve) and the caller then masked is what histogram() bins."""
    from synthpy_amd import resident
    from synthpy_amd.simulator import diagnostics as diag, domain as d, propagator as p

    x, ne, s0, ext, lwl = c1
    dom = d.ScalarDomain(2 * ext, len(x), phaseshift=True)
    dom.external_ne(ne)
    rf, Jf, _ = p.solve(s0, dom, ext, return_E=True, lwl=lwl)
    assert isinstance(rf, resident.TrackedArray) and isinstance(Jf, resident.TrackedArray) and not resident.dirty(rf)
    clean = diag.Shadowgraphy(lwl, rf)
    assert clean.on_device
    clean.two_lens_solve()
    clean.histogram(bin_scale=4)
    rf[0, 17] = np.nan  # one ray of 1e4
    sh = diag.Shadowgraphy(lwl, rf)
    assert not sh.on_device
    sh.two_lens_solve()
    sh.histogram(bin_scale=4)
    host = diag.Shadowgraphy(lwl, np.array(rf))
    host.two_lens_solve()
    host.histogram(bin_scale=4)
    assert np.array_equal(sh.H, host.H) and sh.H.sum() == clean.H.sum() - 1
    # the chain's output handed out and masked
    rf2, _, _ = p.solve(s0, dom, ext, lwl=lwl)
    a, b = diag.Shadowgraphy(lwl, rf2), diag.Shadowgraphy(lwl, np.array(rf2))
    for it in (a, b):
        it.two_lens_solve()
        it.rf[:, ::3] = np.nan
        it.histogram(bin_scale=4)
    assert not a.on_device and np.array_equal(a.H, b.H) and 0 < a.H.sum() < 0.7 * s0.shape[1]


def test_bundle_lifetime_between_solves(eng, orc, c1):
    """A diagnostic keeps depositing from ITS rays when the domain traces the next bundle; a bundle nobody deposits from is
    reused; clear_memory() and pickling bring the rays to the host."""
    from synthpy_amd.solvers_legacy import rtm_solver as rtm

    x, ne, s0, ext, lwl = c1
    dom, (rf, Jf) = _legacy_solve(c1)
    first = dom._rays
    sh = rtm.Shadowgraphy(rf)
    sh.two_lens_solve()
    s1 = s0.copy()
    s1[0] += 2e-4
    rf1 = dom.solve(s1)  # sh still holds the first bundle: the domain takes another one
    assert dom._rays is not first and first.alive
    sh.histogram(bin_scale=10)
    host = rtm.Shadowgraphy(rf.copy())
    host.two_lens_solve()
    host.histogram(bin_scale=10)
    assert sh.on_device and np.array_equal(sh.H, host.H)
    sh1 = rtm.Shadowgraphy(rf1)
    sh1.two_lens_solve()
    sh1.histogram(bin_scale=10, clear_mem=True)
    assert not np.array_equal(sh1.H, sh.H)
    second = dom._rays
    dom.solve(s0)  # nobody deposits from `second` any more: reused
    assert dom._rays is second
    # pickling (the reference's drivers pickle their diagnostics, pvti_trace_mpi.py:176-186) takes the rays to the host
    clone = pickle.loads(pickle.dumps(sh))
    assert not sh.on_device and not clone.on_device
    assert np.array_equal(clone.H, host.H) and np.array_equal(clone.rf, host.rf, equal_nan=True)
    clone.histogram(bin_scale=5)
    host.histogram(bin_scale=5)
    assert np.array_equal(clone.H, host.H)
    # clear_memory: a diagnostic still attached gets its rays on the host first
    dom, (rf, Jf) = _legacy_solve(c1)
    sc = rtm.Schlieren(rf)
    sc.DF_solve()
    dom.clear_memory()
    assert not sc.on_device
    sc.histogram(bin_scale=10)
    hs = rtm.Schlieren(rf.copy())
    hs.DF_solve()
    hs.histogram(bin_scale=10)
    assert np.array_equal(sc.H, hs.H)


def test_simulator_flow_deposits_from_hbm(eng, orc, c1, c1_oracle):
    """The JAX generation's classes (src/simulator/diagnostics.py:269-641) on the bundle propagator.solve() left in HBM:
    counts diagnostics, and Interferometry with a caller's reference beam + the one two_lens_solve adds (diagnostics.py:616)."""
    from synthpy_amd.simulator import diagnostics as diag, domain as d, propagator as p

    x, ne, s0, ext, lwl = c1
    dom = d.ScalarDomain(2 * ext, len(x), phaseshift=True)
    assert np.array_equal(dom.x, np.float32(x))
    dom.external_ne(ne)
    rf, Jf, _ = p.solve(s0, dom, ext, return_E=True, lwl=lwl)
    rf_o, Jf_o = c1_oracle
    for cls, solve, chain in COUNTS:
        dev = getattr(diag, cls)(lwl, rf)
        assert dev.on_device
        getattr(dev, solve)()
        dev.histogram(bin_scale=1)
        host = getattr(diag, cls)(lwl, rf.copy())
        assert not host.on_device
        getattr(host, solve)()
        host.histogram(bin_scale=1)
        r_o, _ = orc.optics(orc.m_to_mm(rf_o), getattr(orc, chain)())
        assert np.array_equal(dev.H, host.H) and np.array_equal(dev.H, orc.histogram(r_o, bin_scale=1)), (cls, solve)
        assert np.array_equal(dev.rf, host.rf, equal_nan=True)
    dev = diag.Interferometry(lwl, rf, Jf)
    host = diag.Interferometry(lwl, rf.copy(), Jf.copy())
    assert dev.on_device and not host.on_device
    assert np.array_equal(dev.rf, rf) and np.array_equal(dev.Jf, Jf)  # before any solve: the rays as given (metres)
    for it in (dev, host):
        it.interfere_ref_beam(10, 10)
    assert dev.on_device and np.array_equal(dev.Jf, host.Jf)
    for it in (dev, host):
        it.two_lens_solve()
        it.interferogram(bin_scale=10)
    assert dev.on_device and np.max(np.abs(dev.H - host.H)) <= 1e-9 * host.H.max()
    E_o = orc.interfere_ref_beam(rf_o, orc.interfere_ref_beam(rf_o, Jf_o, 10, 10), 10, 20)
    r_o, E_o = orc.optics(orc.m_to_mm(rf_o), orc.chain_shadow_two(), E=E_o, kwave=2 * np.pi / lwl)
    H_o = orc.interferogram(r_o, E_o, bin_scale=10)
    assert np.max(np.abs(dev.H - H_o)) <= 1e-5 * H_o.max()
    assert np.array_equal(dev.rf, host.rf, equal_nan=True) and np.array_equal(dev.Jf, host.Jf, equal_nan=True)
    # a fifth reference beam does not fit the deposit: the object carries on on the host, same image
    dev = diag.Interferometry(lwl, rf, Jf)
    host = diag.Interferometry(lwl, rf.copy(), Jf.copy())
    for it in (dev, host):
        for q in range(4):
            it.interfere_ref_beam(3 + q, 5 * q)
        it.two_lens_solve()
        it.interferogram(bin_scale=10)
    assert not dev.on_device and np.max(np.abs(dev.H - host.H)) <= 1e-9 * host.H.max()


def test_headline_shape_through_the_reference_api(eng, orc):
    """A dense bundle through the reference's API: 2e5 rays in a narrow beam (>= 16 rays per lateral cell) through
    bench.make_volume(256) -- solve() takes the tile kernel by itself, Interferometry deposits from HBM -- against the oracle
    from the same s0: exit rays, shadowgram counts, interferogram."""
    import bench
    from synthpy_amd.solvers_legacy import full_solver as fs, rtm_solver as rtm

    ne, x = bench.make_volume(256)
    ext, lwl, N = 5e-3, 1064e-9, 200000
    np.random.seed(5)
    s0 = fs.init_beam(N, 1.0e-3, 5e-5, ext, "circular", "z")  # pi * (1 mm / 39 um)^2 = 2050 cells: ~100 rays per cell
    dom = fs.ScalarDomain(x, x, x, ext, phaseshift=True)
    dom.external_ne(ne)
    dom.calc_dndr(lwl)
    rf, Jf = dom.solve(s0, return_E=True)
    assert dom._rays.tile_segments >= 1, "a dense float64 bundle did not take the tile path"
    odom = orc.Domain.from_ne(ne, x, x, x, lwl, phaseshift=True)
    sf_o, steps = orc.trace_rk4(odom, s0, (x[1] - x[0]) / orc.c, orc.default_t_end(ext), "z", "planes", 1)
    rf_o, Jf_o = orc.ray_to_jones(sf_o, ext, "z", "legacy")
    assert dom.trace_stats.ray_steps == steps
    assert np.max(np.abs(rf[0::2] - rf_o[0::2])) <= 1e-13 and np.max(np.abs(rf[1::2] - rf_o[1::2])) <= 1e-11
    sh = rtm.Shadowgraphy(rf)
    sh.two_lens_solve()
    sh.histogram(bin_scale=1)
    r_o, _ = orc.optics(orc.m_to_mm(rf_o), orc.chain_shadow_two())
    assert sh.on_device and np.array_equal(sh.H, orc.histogram(r_o, bin_scale=1))
    it = rtm.Interferometry(rf, E=Jf)
    it.two_lens_solve(wl=lwl)
    it.interferogram(bin_scale=1)
    r_o, E_o = orc.optics(orc.m_to_mm(rf_o), orc.chain_shadow_two(), E=Jf_o, kwave=2 * np.pi / lwl)
    H_o = orc.interferogram(r_o, E_o, bin_scale=1)
    assert it.on_device and np.max(np.abs(it.H - H_o)) <= 1e-5 * H_o.max()
