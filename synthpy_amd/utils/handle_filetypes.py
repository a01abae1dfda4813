"""File formats either side of the path: VTK image data (.vti / .pvti) without vtk or pyvista.

Mirror of src/utils/handle_filetypes.py:
    export_pvti(arr, fname, extent_x, extent_y, extent_z)     :11-90    n_e volume -> <fname>.vti + <fname>.pvti
    pvti_readin(filename) -> (img, img.shape, spacing)        :92-121   cell array 0 of a .pvti (or .vti)
    hdf_readin / hdf_to_pvti                                   :123-161  FLASH AMR blocks -> uniform grid (flash_covering_grid; file read by h5py or utils/hdf5_lite.py)

The reference writes with pyvista (`grid.cell_data["rnec"] = arr.flatten(order="F")`, handle_filetypes.py:60-62)
and reads with vtkXMLPImageDataReader (:99-119): CELL data named "rnec", x fastest.  This module reads and writes
that format directly: VTK XML ImageData, `ascii`, inline `binary` (base64) or `appended` (base64 or raw) arrays,
uncompressed or vtkZLibDataCompressor blocks, UInt32 or UInt64 block headers, either byte order.
Reading feeds ScalarDomain.external_ne(); nothing here touches the GPU.
"""
from __future__ import annotations

import base64
import os
import re
import xml.etree.ElementTree as ET
import zlib

import numpy as np

_VTK_TYPES = {"Float32": "f4", "Float64": "f8", "Int8": "i1", "UInt8": "u1", "Int16": "i2", "UInt16": "u2",
              "Int32": "i4", "UInt32": "u4", "Int64": "i8", "UInt64": "u8"}
_NP_TO_VTK = {np.dtype(v).str[1:]: k for k, v in _VTK_TYPES.items()}
_BLOCK = 1 << 15  # vtkXMLWriter's default compression block size


# ------------------------------------------------------------------------------------------------ reading
class _VtkXml:
    """One VTK XML file: the element tree plus the appended-data section (kept out of the XML parser, it may be raw)."""

    def __init__(self, path):
        raw = open(path, "rb").read()
        self.appended, self.app_encoding = None, None
        m = re.search(rb"<AppendedData[^>]*>", raw)
        if m:
            enc = re.search(rb'encoding\s*=\s*"(\w+)"', m.group(0))
            self.app_encoding = enc.group(1).decode() if enc else "base64"
            start = raw.index(b"_", m.end()) + 1
            end = raw.rindex(b"</AppendedData>")
            self.appended = raw[start:end]
            raw = raw[:m.start()] + raw[end + len(b"</AppendedData>"):]
        self.root = ET.fromstring(raw)
        if self.root.tag != "VTKFile":
            raise ValueError(f"{path}: not a VTK XML file")
        self.order = "<" if self.root.get("byte_order", "LittleEndian") == "LittleEndian" else ">"
        self.header = self.order + ("u8" if self.root.get("header_type", "UInt32") == "UInt64" else "u4")
        self.compressor = self.root.get("compressor")
        if self.compressor not in (None, "", "vtkZLibDataCompressor"):
            raise NotImplementedError(f"{path}: compressor {self.compressor} (only vtkZLibDataCompressor is read)")
        self.path = path

    # -- binary payloads -------------------------------------------------------------------------
    def _blocks(self, read, nbytes_expected):
        """read(offset, n) -> bytes of the decoded stream; returns the array's bytes."""
        hs = np.dtype(self.header).itemsize
        if not self.compressor:
            n = int(np.frombuffer(read(0, hs), self.header)[0])
            return read(hs, n)
        nblocks, bsize, last = (int(v) for v in np.frombuffer(read(0, 3 * hs), self.header))
        sizes = np.frombuffer(read(3 * hs, nblocks * hs), self.header).astype(np.int64)
        out, pos = [], (3 + nblocks) * hs
        for s in sizes:
            out.append(zlib.decompress(read(pos, int(s))))
            pos += int(s)
        return b"".join(out)

    def _decode_base64(self, text, nbytes_expected):
        """VTK base64: header and (when compressed) the blocks are separate base64 streams, back to back."""
        hs = np.dtype(self.header).itemsize
        if not self.compressor:
            buf = base64.b64decode(text)
            return self._blocks(lambda o, n: buf[o:o + n], nbytes_expected)
        enc_len = lambda nbytes: 4 * ((nbytes + 2) // 3)
        first = base64.b64decode(text[:enc_len(3 * hs)])
        nblocks = int(np.frombuffer(first[:hs], self.header)[0])
        hlen = enc_len((3 + nblocks) * hs)
        head = base64.b64decode(text[:hlen])[: (3 + nblocks) * hs]
        sizes = np.frombuffer(head[3 * hs:], self.header).astype(np.int64)
        body = base64.b64decode(text[hlen:hlen + enc_len(int(sizes.sum()))])
        out, pos = [], 0
        for s in sizes:
            out.append(zlib.decompress(body[pos:pos + int(s)]))
            pos += int(s)
        return b"".join(out)

    def array(self, el, n_tuples):
        """Decode one <DataArray> holding n_tuples tuples -> 1-D or (n_tuples, ncomp) array."""
        dt = np.dtype(self.order + _VTK_TYPES[el.get("type")])
        ncomp = int(el.get("NumberOfComponents", "1"))
        count = n_tuples * ncomp
        fmt = el.get("format", "ascii")
        if fmt == "ascii":
            a = np.array(el.text.split(), dtype=dt.newbyteorder("="))
        elif fmt == "binary":
            a = np.frombuffer(self._decode_base64("".join(el.text.split()).encode(), count * dt.itemsize), dt)
        elif fmt == "appended":
            off = int(el.get("offset", "0"))
            if self.appended is None:
                raise ValueError(f"{self.path}: appended array without an <AppendedData> section")
            if self.app_encoding == "raw":
                a = np.frombuffer(self._blocks(lambda o, n: self.appended[off + o:off + o + n], count * dt.itemsize), dt)
            else:
                text = b"".join(self.appended[off:].split())
                a = np.frombuffer(self._decode_base64(text, count * dt.itemsize), dt)
        else:
            raise ValueError(f"{self.path}: DataArray format {fmt!r}")
        if a.size < count:
            raise ValueError(f"{self.path}: array {el.get('Name')!r} holds {a.size} values, {count} expected")
        a = a[:count].astype(dt.newbyteorder("="), copy=False)
        return a.reshape(n_tuples, ncomp) if ncomp > 1 else a


def _ints(s):
    return [int(v) for v in s.split()]


def vti_read(filename, array=0):
    """One .vti -> (cells (nx, ny, nz[, ncomp]), extent [x0, x1, y0, y1, z0, z1], spacing (3,), origin (3,)).
    `array`: index or Name of the CELL data array."""
    f = _VtkXml(filename)
    img = f.root.find("ImageData")
    if img is None:
        raise ValueError(f"{filename}: no <ImageData>")
    piece = img.find("Piece")
    ext = _ints(piece.get("Extent", img.get("WholeExtent")))
    dims = (ext[1] - ext[0], ext[3] - ext[2], ext[5] - ext[4])
    arrays = piece.find("CellData").findall("DataArray")
    if not arrays:
        raise ValueError(f"{filename}: no cell data (the reference stores n_e as CELL data 'rnec')")
    el = arrays[array] if isinstance(array, int) else next(a for a in arrays if a.get("Name") == array)
    v = f.array(el, dims[0] * dims[1] * dims[2])
    shape = dims + ((v.shape[1],) if v.ndim == 2 else ())
    spacing = np.array([float(s) for s in img.get("Spacing", "1 1 1").split()])
    origin = np.array([float(s) for s in img.get("Origin", "0 0 0").split()])
    return v.reshape(shape, order="F"), ext, spacing, origin


def pvti_readin(filename):
    """Cell array 0 of a .pvti (all its pieces) or a single .vti -> (img, img.shape, spacing)
    (handle_filetypes.py:92-121): img is (nx, ny, nz) or (nx, ny, nz, n_comp), x fastest on disk."""
    if str(filename).endswith(".vti"):
        img, _, spacing, _ = vti_read(filename)
        return img, img.shape, spacing
    root = ET.parse(filename).getroot()
    pimg = root.find("PImageData")
    if root.tag != "VTKFile" or pimg is None:
        raise ValueError(f"{filename}: not a PImageData file")
    whole = _ints(pimg.get("WholeExtent"))
    spacing = np.array([float(s) for s in pimg.get("Spacing", "1 1 1").split()])
    base = os.path.dirname(os.path.abspath(filename))
    out = None
    for piece in pimg.findall("Piece"):
        part, ext, _, _ = vti_read(os.path.join(base, piece.get("Source")))
        pext = _ints(piece.get("Extent")) if piece.get("Extent") else ext
        if out is None:
            shape = (whole[1] - whole[0], whole[3] - whole[2], whole[5] - whole[4]) + part.shape[3:]
            out = np.empty(shape, part.dtype)
        out[pext[0] - whole[0]:pext[1] - whole[0], pext[2] - whole[2]:pext[3] - whole[2],
            pext[4] - whole[4]:pext[5] - whole[4]] = part
    if out is None:
        raise ValueError(f"{filename}: no <Piece>")
    return out, out.shape, spacing


# ------------------------------------------------------------------------------------------------ writing
def _encode(data: bytes, header: str, compress: bool, b64: bool) -> bytes:
    if not compress:
        blob = np.array([len(data)], header).tobytes() + data
        return base64.b64encode(blob) if b64 else blob
    blocks = [zlib.compress(data[i:i + _BLOCK]) for i in range(0, len(data), _BLOCK)] or [zlib.compress(b"")]
    last = len(data) % _BLOCK
    head = np.array([len(blocks), _BLOCK, last] + [len(b) for b in blocks], header).tobytes()
    body = b"".join(blocks)
    return base64.b64encode(head) + base64.b64encode(body) if b64 else head + body


def vti_write(filename, arr, spacing=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0), name="rnec", *, mode="appended",
              encoding="base64", compress=True, header_type="UInt64"):
    """Write arr (nx, ny, nz[, ncomp]) as CELL data `name` of an ImageData .vti, x fastest (order="F"), as pyvista's
    save does for the reference (handle_filetypes.py:60-64).  mode: "appended" | "binary" | "ascii"."""
    arr = np.asarray(arr)
    if arr.ndim not in (3, 4):
        raise ValueError(f"expected a (nx, ny, nz[, ncomp]) array, got shape {arr.shape}")
    key = arr.dtype.newbyteorder("=").str[1:]
    if key not in _NP_TO_VTK:
        raise TypeError(f"dtype {arr.dtype} has no VTK type")
    nx, ny, nz = arr.shape[:3]
    ncomp = arr.shape[3] if arr.ndim == 4 else 1
    flat = (arr.reshape(nx * ny * nz, ncomp, order="F") if arr.ndim == 4 else arr.flatten(order="F")).astype("<" + key)
    header = "<u8" if header_type == "UInt64" else "<u4"
    attrs = f'type="{_NP_TO_VTK[key]}" Name="{name}"' + (f' NumberOfComponents="{ncomp}"' if ncomp > 1 else "")
    finite = flat[np.isfinite(flat)] if flat.dtype.kind == "f" else flat
    if finite.size:
        attrs += f' RangeMin="{finite.min().item()!r}" RangeMax="{finite.max().item()!r}"'
    comp_attr = ' compressor="vtkZLibDataCompressor"' if compress and mode != "ascii" else ""
    ext = f"0 {nx} 0 {ny} 0 {nz}"
    head = (f'<?xml version="1.0"?>\n<VTKFile type="ImageData" version="1.0" byte_order="LittleEndian" '
            f'header_type="{header_type}"{comp_attr}>\n'
            f'  <ImageData WholeExtent="{ext}" Origin="{origin[0]!r} {origin[1]!r} {origin[2]!r}" '
            f'Spacing="{float(spacing[0])!r} {float(spacing[1])!r} {float(spacing[2])!r}" Direction="1 0 0 0 1 0 0 0 1">\n'
            f'    <Piece Extent="{ext}">\n      <PointData/>\n      <CellData Scalars="{name}">\n').encode()
    tail_piece = b"      </CellData>\n    </Piece>\n  </ImageData>\n"
    with open(filename, "wb") as fh:
        fh.write(head)
        if mode == "ascii":
            fh.write(f'        <DataArray {attrs} format="ascii">\n'.encode())
            fh.write(" ".join(repr(v.item()) for v in flat.ravel()).encode())
            fh.write(b"\n        </DataArray>\n" + tail_piece)
        elif mode == "binary":
            fh.write(f'        <DataArray {attrs} format="binary">\n'.encode())
            fh.write(_encode(flat.tobytes(), header, compress, True))
            fh.write(b"\n        </DataArray>\n" + tail_piece)
        elif mode == "appended":
            fh.write(f'        <DataArray {attrs} format="appended" offset="0"/>\n'.encode() + tail_piece)
            fh.write(f'  <AppendedData encoding="{encoding}">\n   _'.encode())
            fh.write(_encode(flat.tobytes(), header, compress, encoding == "base64"))
            fh.write(b"\n  </AppendedData>\n")
        else:
            raise ValueError(f"mode {mode!r}")
        fh.write(b"</VTKFile>\n")


def export_pvti(arr, fname=None, extent_x=None, extent_y=None, extent_z=None):
    """Save a 3-D array as <fname>.vti + <fname>.pvti (handle_filetypes.py:11-90): cell data "rnec", spacing
    max(linspace(-extent, extent, n)) / (n // 2) per axis, extents default to n // 2."""
    if fname is None:
        import datetime as dt

        now = dt.datetime.now()
        fname = f"./plasma_PVTI_{now.day}_{now.month}_{now.year}_{now.hour}_{now.minute}"
    try:
        shape = np.shape(arr)
        assert len(shape) == 3
    except Exception:
        raise Exception("No electron density currently loaded!")
    arr = np.asarray(arr)
    ext = [shape[k] // 2 if e is None else e for k, e in enumerate((extent_x, extent_y, extent_z))]
    size = [float(np.max(np.linspace(-ext[k], ext[k], shape[k])) / (shape[k] // 2)) for k in range(3)]
    vti_write(f"{fname}.vti", arr, spacing=size)
    print(f"VTI saved under {fname}.vti")
    _write_pvti(fname, shape, size, arr.dtype)
    print(f"Scalar Domain electron density succesfully saved under {fname}.pvti !")


def export_scalar_field(domain, property="ne", fname=None):
    """ScalarDomain.export_scalar_field (full_solver.py:442-512; domain.py:505-579): n_e as cell data "rnec" of <fname>.vti
    with cell size max(coord) / ((n - 1) // 2), and the .pvti index with spacing 2*max(coord) / n, both as written."""
    if fname is None:
        import datetime as dt

        now = dt.datetime.now()
        fname = f"./plasma_PVTI_{now.day}_{now.month}_{now.year}_{now.hour}_{now.minute}"
    if property != "ne":
        raise ValueError("only property='ne' is exported (as in the reference)")
    if getattr(domain, "ne", None) is None:
        raise Exception("No electron density currently loaded!")
    ne = np.asarray(domain.ne)
    cell = [float(np.max(c)) / ((ne.shape[k] - 1) // 2) for k, c in enumerate((domain.x, domain.y, domain.z))]
    vti_write(f"{fname}.vti", ne, spacing=cell)
    print(f"VTI saved under {fname}.vti")
    _write_pvti(fname, ne.shape, [2 * float(np.max(c)) / len(c) for c in (domain.x, domain.y, domain.z)], ne.dtype)
    print(f"Scalar Domain electron density succesfully saved under {fname}.pvti !")


def _write_pvti(fname, shape, spacing, dtype):
    rel = fname.split("/")[-1]
    vt = _NP_TO_VTK[np.dtype(dtype).newbyteorder("=").str[1:]]
    ext = f"0 {shape[0]} 0 {shape[1]} 0 {shape[2]}"
    with open(f"{fname}.pvti", "w") as fh:
        fh.write(f'''<?xml version="1.0"?>
<VTKFile type="PImageData" version="0.1" byte_order="LittleEndian" header_type="UInt32" compressor="vtkZLibDataCompressor">
  <PImageData WholeExtent="{ext}" GhostLevel="0" Origin="0 0 0" Spacing="{spacing[0]} {spacing[1]} {spacing[2]}">
    <PCellData Scalars="rnec">
      <PDataArray type="{vt}" Name="rnec">
      </PDataArray>
    </PCellData>
    <Piece Extent="{ext}" Source="{rel}.vti"/>
  </PImageData>
</VTKFile>''')


def flash_covering_grid(bbox, level, node_type, fields, ndim=3):
    """Uniform grid at the finest refinement level from the blocks of a FLASH AMR file: what yt's
    `ds.covering_grid(max_level, left_edge=domain_left_edge, dims=domain_dimensions * 2**max_level)` returns for
    cell-centred data (handle_filetypes.py:144-147) -- every leaf block's cells copied into place, a coarser block's
    cells repeated 2**(levels below the finest) times along each refined axis, no interpolation.

    bbox (B, 3, 2): lower / upper corner of each block; level (B,): 1-based refinement level; node_type (B,): 1 = leaf;
    fields: dict name -> (B, nzb, nyb, nxb) block data (FLASH's order).  Returns ({name: (nx, ny, nz) float64}, dims,
    spacing)."""
    bbox = np.asarray(bbox, dtype=np.float64)
    level = np.asarray(level).astype(np.int64).ravel()
    leaf = np.asarray(node_type).ravel() == 1
    if bbox.ndim != 3 or bbox.shape[1:] != (3, 2) or len(level) != bbox.shape[0] or len(leaf) != bbox.shape[0]:
        raise ValueError("flash_covering_grid: bbox must be (blocks, 3, 2), one level and node type per block")
    if not leaf.any():
        raise ValueError("flash_covering_grid: no leaf blocks")
    first = next(iter(fields.values()))
    nb = np.array(first.shape[:0:-1])  # cells per block along x, y, z
    lmax = int(level[leaf].max())
    lo, hi = bbox[:, :, 0].min(axis=0), bbox[:, :, 1].max(axis=0)
    refined = np.arange(3) < ndim
    # cell width at the finest level, from any block of that level
    bmax = int(np.flatnonzero(leaf & (level == lmax))[0])
    dx = (bbox[bmax, :, 1] - bbox[bmax, :, 0]) / nb
    step = np.where(dx > 0, dx, 1.0)  # an unused axis of a 1-D / 2-D file may have no extent at all (bounds 0, 0)
    dims = np.where(refined, np.rint((hi - lo) / step), nb).astype(np.int64)
    out = {k: np.zeros(tuple(dims), dtype=np.float64) for k in fields}
    filled = np.zeros(tuple(dims), dtype=bool)
    for bi in np.flatnonzero(leaf):
        rep = np.where(refined, 2 ** (lmax - level[bi]), 1)
        i0 = np.where(refined, np.rint((bbox[bi, :, 0] - lo) / step), 0).astype(np.int64)
        i1 = i0 + nb * rep
        if (i0 < 0).any() or (i1 > dims).any():
            raise ValueError(f"flash_covering_grid: block {bi} does not lie on the finest level's grid")
        sl = tuple(slice(a, b) for a, b in zip(i0, i1))
        for k, v in fields.items():
            blk = np.asarray(v[bi], dtype=np.float64).transpose(2, 1, 0)  # (nzb, nyb, nxb) -> x, y, z
            for ax in range(3):
                if rep[ax] > 1:
                    blk = np.repeat(blk, rep[ax], axis=ax)
            out[k][sl] = blk
        filled[sl] = True
    if not filled.all():
        raise ValueError("flash_covering_grid: the leaf blocks do not cover the domain")
    return out, dims, [float(v) for v in dx]


def _open_hdf5(filename):
    """h5py where it is installed, else this package's own reader (hdf5_lite: numpy + zlib, the structures the HDF5
    library writes by default -- FLASH's files -- and most of H5F_LIBVER_LATEST's)."""
    try:
        import h5py
        return h5py.File(filename, "r")
    except ImportError:
        from . import hdf5_lite
        return hdf5_lite.File(filename)


def hdf_readin(filename):
    """n_e on the finest level's uniform grid from a FLASH AMR file (handle_filetypes.py:121-150): n_e =
    6.022e23 * dens * ye * sumy per cell, returned with the grid's dims and spacing in the file's units, as the
    reference returns them (it reads through yt; here the block tables are read directly -- `bounding box`, `refine
    level`, `node type` and the three variables -- and assembled by flash_covering_grid).  The file is opened with h5py
    where that is installed and with utils/hdf5_lite.py otherwise.  FLASH names its variables by four characters, blank
    padded (`ye  `): names are compared without their blanks, as yt does."""
    with _open_hdf5(filename) as f:
        by_name = {str(k).strip(): k for k in f.keys()}
        missing = [k for k in ("bounding box", "refine level", "node type", "dens", "ye", "sumy") if k not in by_name]
        if missing:
            raise KeyError(f"{filename}: not a FLASH file with dens/ye/sumy (missing {missing})")
        get = lambda k: np.asarray(f[by_name[k]][...])
        bbox, level, ntype = get("bounding box"), get("refine level"), get("node type")
        fields = {k: get(k) for k in ("dens", "ye", "sumy")}
        ndim = 3
        if "integer scalars" in by_name:
            for rec in get("integer scalars"):
                name, val = rec[0], rec[1]
                if (name.decode() if isinstance(name, bytes) else str(name)).strip() == "dimensionality":
                    ndim = int(val)
    if bbox.shape[1] < 3:  # lower-dimensional files carry fewer rows: pad with unit extents
        pad = np.zeros((bbox.shape[0], 3 - bbox.shape[1], 2))
        pad[:, :, 1] = 1.0
        bbox = np.concatenate([bbox, pad], axis=1)
    g, dims, spacing = flash_covering_grid(bbox, level, ntype, fields, ndim)
    return 6.022e23 * g["dens"] * g["ye"] * g["sumy"], dims, spacing


def hdf_to_pvti(hdf_filename, pvti_filename):
    ne, dims, spacing = hdf_readin(hdf_filename)
    export_pvti(ne, fname=pvti_filename, extent_x=dims[0] * spacing[0] / 2, extent_y=dims[1] * spacing[1] / 2,
                extent_z=dims[2] * spacing[2] / 2)
