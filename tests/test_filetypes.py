"""VTK image-data reader / writer (src/utils/handle_filetypes.py of the reference) -- CPU only, no vtk / pyvista."""
import os

import numpy as np
import pytest

from synthpy_amd.utils import handle_filetypes as hf


@pytest.fixture
def cube():
    rng = np.random.default_rng(3)
    return 1e25 * rng.random((7, 5, 9))


MODES = [dict(mode="appended", encoding="base64", compress=True, header_type="UInt64"),   # what pyvista's save writes
         dict(mode="appended", encoding="base64", compress=True, header_type="UInt32"),
         dict(mode="appended", encoding="base64", compress=False, header_type="UInt64"),
         dict(mode="appended", encoding="raw", compress=True, header_type="UInt32"),
         dict(mode="appended", encoding="raw", compress=False, header_type="UInt64"),
         dict(mode="binary", compress=True, header_type="UInt32"),
         dict(mode="binary", compress=False, header_type="UInt64"),
         dict(mode="ascii")]


@pytest.mark.parametrize("kw", MODES, ids=lambda k: "-".join(str(v) for v in k.values()))
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_vti_round_trip_every_encoding(tmp_path, cube, kw, dtype):
    a = cube.astype(dtype)
    path = str(tmp_path / "c.vti")
    hf.vti_write(path, a, spacing=(1e-4, 2e-4, 3e-4), **kw)
    img, shape, spacing = hf.pvti_readin(path)
    assert shape == a.shape and img.dtype == dtype and np.array_equal(img, a)
    assert np.array_equal(spacing, [1e-4, 2e-4, 3e-4])


def test_disk_order_is_x_fastest_cell_data(tmp_path):
    """The reference stores arr.flatten(order="F") as CELL data of a grid with shape+1 points (handle_filetypes.py:42,60)."""
    a = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    path = str(tmp_path / "o.vti")
    hf.vti_write(path, a, mode="ascii")
    txt = open(path).read()
    assert 'WholeExtent="0 2 0 3 0 4"' in txt and "<CellData" in txt and 'Name="rnec"' in txt
    vals = [float(v) for v in txt.split('format="ascii">')[1].split("</DataArray>")[0].split()]
    assert vals == list(a.flatten(order="F"))


def test_large_array_spans_many_zlib_blocks(tmp_path):
    a = np.random.default_rng(0).random((40, 40, 40))  # 512 KB -> 16 blocks of 32 KiB
    path = str(tmp_path / "big.vti")
    hf.vti_write(path, a)
    assert np.array_equal(hf.pvti_readin(path)[0], a)
    hf.vti_write(path, a, mode="binary", header_type="UInt32")
    assert np.array_equal(hf.pvti_readin(path)[0], a)


def test_vector_cell_data(tmp_path):
    B = np.random.default_rng(1).standard_normal((4, 5, 6, 3))
    path = str(tmp_path / "B.vti")
    hf.vti_write(path, B, name="B")
    img, shape, _ = hf.pvti_readin(path)
    assert shape == (4, 5, 6, 3) and np.array_equal(img, B)


def test_export_pvti_and_readin(tmp_path, cube, capsys):
    """export_pvti -> pvti_readin round trip with the reference's spacing rule (handle_filetypes.py:46-58)."""
    base = str(tmp_path / "plasma")
    hf.export_pvti(cube, base, extent_x=3.0, extent_y=2.0, extent_z=4.0)
    assert os.path.exists(base + ".vti") and os.path.exists(base + ".pvti")
    img, shape, spacing = hf.pvti_readin(base + ".pvti")
    assert np.array_equal(img, cube) and shape == cube.shape
    assert np.allclose(spacing, [3.0 / (7 // 2), 2.0 / (5 // 2), 4.0 / (9 // 2)])
    assert 'Source="plasma.vti"' in open(base + ".pvti").read()
    with pytest.raises(Exception, match="No electron density"):
        hf.export_pvti(None, base)


def test_multi_piece_pvti(tmp_path, cube):
    """A .pvti whose pieces split the x range (what a parallel writer produces)."""
    hf.vti_write(str(tmp_path / "p0.vti"), cube[:3])
    hf.vti_write(str(tmp_path / "p1.vti"), cube[3:])
    # the second piece's own file says extent 0..4; the index places it at 3..7
    (tmp_path / "all.pvti").write_text('''<?xml version="1.0"?>
<VTKFile type="PImageData" version="0.1" byte_order="LittleEndian">
  <PImageData WholeExtent="0 7 0 5 0 9" GhostLevel="0" Origin="0 0 0" Spacing="1 1 1">
    <PCellData Scalars="rnec"><PDataArray type="Float64" Name="rnec"/></PCellData>
    <Piece Extent="0 3 0 5 0 9" Source="p0.vti"/>
    <Piece Extent="3 7 0 5 0 9" Source="p1.vti"/>
  </PImageData>
</VTKFile>''')
    img, shape, _ = hf.pvti_readin(str(tmp_path / "all.pvti"))
    assert shape == cube.shape and np.array_equal(img, cube)


def test_big_endian_uncompressed_binary(tmp_path):
    """Hand-built file in the other byte order: header UInt32 [nbytes] + data, one base64 stream."""
    import base64

    a = np.arange(8, dtype=">f4")
    blob = base64.b64encode(np.array([a.nbytes], ">u4").tobytes() + a.tobytes()).decode()
    (tmp_path / "be.vti").write_text(f'''<?xml version="1.0"?>
<VTKFile type="ImageData" version="0.1" byte_order="BigEndian">
  <ImageData WholeExtent="0 2 0 2 0 2" Origin="0 0 0" Spacing="0.5 0.5 0.5">
    <Piece Extent="0 2 0 2 0 2"><CellData><DataArray type="Float32" Name="rnec" format="binary">{blob}</DataArray></CellData></Piece>
  </ImageData>
</VTKFile>''')
    img, shape, spacing = hf.pvti_readin(str(tmp_path / "be.vti"))
    assert shape == (2, 2, 2) and np.array_equal(img.flatten(order="F"), np.arange(8)) and spacing[0] == 0.5


def test_unsupported_inputs(tmp_path):
    (tmp_path / "lz4.vti").write_text('<?xml version="1.0"?><VTKFile type="ImageData" compressor="vtkLZ4DataCompressor">'
                                      '<ImageData WholeExtent="0 1 0 1 0 1"><Piece Extent="0 1 0 1 0 1"><CellData/></Piece>'
                                      '</ImageData></VTKFile>')
    with pytest.raises(NotImplementedError):
        hf.pvti_readin(str(tmp_path / "lz4.vti"))
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="h5py"):
            hf.hdf_readin("x.h5")


def _amr_blocks(f, refine_first=True, ndim=3):
    """A FLASH-like block table on [0,1]^3 (or [0,1]^2 x one cell): 2 x 2 (x 2) level-1 blocks of 4^ndim cells, the
    first one refined into level-2 children; cell data = f at the cell centres, (B, nzb, nyb, nxb)."""
    nb = np.array([4, 4, 4 if ndim == 3 else 1])
    boxes, levels, types = [], [], []

    def add(lo, w, lvl, leaf):
        hi = lo + w
        boxes.append(np.stack([lo, hi], axis=1))
        levels.append(lvl)
        types.append(1 if leaf else 2)

    nblk = (2, 2, 2 if ndim == 3 else 1)
    w1 = np.array([0.5, 0.5, 0.5 if ndim == 3 else 1.0])
    for k in range(nblk[2]):
        for j in range(nblk[1]):
            for i in range(nblk[0]):
                lo = np.array([i, j, k]) * w1
                is_first = (i, j, k) == (0, 0, 0)
                add(lo, w1, 1, not (is_first and refine_first))
                if is_first and refine_first:
                    w2 = w1 / np.where(np.arange(3) < ndim, 2, 1)
                    for kk in range(2 if ndim == 3 else 1):
                        for jj in range(2):
                            for ii in range(2):
                                add(lo + np.array([ii, jj, kk]) * w2, w2, 2, True)
    bbox = np.array(boxes)
    data = np.zeros((len(boxes), nb[2], nb[1], nb[0]))
    for b, bx in enumerate(bbox):
        cx, cy, cz = (bx[d, 0] + (np.arange(nb[d]) + 0.5) * (bx[d, 1] - bx[d, 0]) / nb[d] for d in range(3))
        data[b] = f(cx[None, None, :], cy[None, :, None], cz[:, None, None])
    return bbox, np.array(levels), np.array(types), data


@pytest.mark.parametrize("ndim", [3, 2])
def test_flash_covering_grid(ndim):
    """hdf_readin's assembly step (yt's covering grid at the finest level, handle_filetypes.py:144-147): fine cells in
    place, coarse cells repeated, parents ignored; arrays come out (x, y, z)."""
    def f(x, y, z):
        return 1.0 + x + 10.0 * y + 100.0 * z + 0 * (x + y + z)

    bbox, lvl, typ, data = _amr_blocks(f, True, ndim)
    data[0] = -7.0  # the refined parent's own cells must not show up
    out, dims, spacing = hf.flash_covering_grid(bbox, lvl, typ, {"dens": data, "ye": 2 * data}, ndim)
    n = 16
    assert list(dims) == [n, n, n if ndim == 3 else 1]
    assert np.allclose(spacing[:2], [1 / n, 1 / n]) and np.isclose(spacing[2], 1 / n if ndim == 3 else 1.0)
    fine = (np.arange(n) + 0.5) / n
    coarse = (np.arange(n) // 2 + 0.5) / (n // 2)
    zf = fine if ndim == 3 else np.array([0.5])
    zc = coarse if ndim == 3 else np.array([0.5])
    X, Y, Z = np.meshgrid(fine, fine, zf, indexing="ij")
    Xc, Yc, Zc = np.meshgrid(coarse, coarse, zc, indexing="ij")
    in_ref = (X < 0.5) & (Y < 0.5) & ((Z < 0.5) if ndim == 3 else True)
    want = np.where(in_ref, f(X, Y, Z), f(Xc, Yc, Zc))
    assert np.allclose(out["dens"], want, rtol=0, atol=1e-12) and np.array_equal(out["ye"], 2 * out["dens"])
    # unrefined file: the level-1 grid itself
    bbox, lvl, typ, data = _amr_blocks(f, False, ndim)
    out, dims, _ = hf.flash_covering_grid(bbox, lvl, typ, {"dens": data}, ndim)
    assert list(dims) == [8, 8, 8 if ndim == 3 else 1]
    c8 = (np.arange(8) + 0.5) / 8
    X, Y, Z = np.meshgrid(c8, c8, c8 if ndim == 3 else np.array([0.5]), indexing="ij")
    assert np.allclose(out["dens"], f(X, Y, Z), rtol=0, atol=1e-12)
    # a hole in the leaves is an error, not zeros
    with pytest.raises(ValueError, match="cover"):
        hf.flash_covering_grid(bbox[1:], lvl[1:], typ[1:], {"dens": data[1:]}, ndim)


def test_export_scalar_field_spacing_rules(tmp_path, capsys):
    """export_scalar_field's two spacing formulas as written (full_solver.py:481-484, 498-500)."""
    class Dom:
        x = np.linspace(-5e-3, 5e-3, 9)
        y = np.linspace(-4e-3, 4e-3, 7)
        z = np.linspace(-3e-3, 3e-3, 5)
        ne = np.random.default_rng(2).random((9, 7, 5))

    base = str(tmp_path / "dom")
    hf.export_scalar_field(Dom, "ne", base)
    img, shape, sp_index = hf.pvti_readin(base + ".pvti")
    assert np.array_equal(img, Dom.ne)
    assert np.allclose(sp_index, [2 * 5e-3 / 9, 2 * 4e-3 / 7, 2 * 3e-3 / 5])
    _, _, sp_cell = hf.pvti_readin(base + ".vti")
    assert np.allclose(sp_cell, [5e-3 / 4, 4e-3 / 3, 3e-3 / 2])
