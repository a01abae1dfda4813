"""Makes the HDF5 fixtures of tests/test_filetypes.py with the HDF5 library itself, then reads every dataset back with the
library's own h5dump into expected.npz -- what synthpy_amd/utils/hdf5_lite.py (numpy + zlib only) has to reproduce.

    python tests/golden/hdf5/make_fixtures.py [HDF5 prefix, default /opt/conda]

Needs an HDF5 installation with headers and tools (1.10.6 under /opt/conda in the build container; the interpreter there has
no h5py, which is why the reader exists).  The tests need only the committed .h5 files and expected.npz."""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PREFIX = sys.argv[1] if len(sys.argv) > 1 else "/opt/conda"
MODES = {"flash_default": 0, "flash_repacked": 1, "flash_latest": 2}
INT_REC = np.dtype([("name", "S80"), ("value", "<i4")])
REAL_REC = np.dtype([("name", "S80"), ("value", "<f8")])
REC = np.dtype([("id", "u1"), ("vec", "<f4", (3,)), ("flag", "u1"), ("last", "<i2")])  # h5dump writes compound records PACKED (16 bytes;
# make_flash_h5.c's rec_t is 20 in memory and in the file): the test compares field by field
VAR = (12, 3, 2, 4)
DATASETS = {  # name -> (little-endian dtype h5dump -b LE writes, shape, modes it exists in)
    "dens": ("<f4", VAR, (0, 1, 2)), "ye  ": ("<f4", VAR, (0, 1, 2)), "sumy": ("<f8", VAR, (0, 1, 2)),
    "bounding box": ("<f8", (12, 3, 2), (0, 1, 2)), "coordinates": ("<f8", (12, 3), (0, 1)),
    "block size": ("<f8", (12, 3), (0, 1)), "refine level": ("<i4", (12,), (0, 1, 2)), "node type": ("<i4", (12,), (0, 1, 2)),
    "integer scalars": (INT_REC, (6,), (0, 1, 2)), "real scalars": (REAL_REC, (2,), (0, 1)),
    "unknown names": ("S4", (3, 1), (0, 1)), "extra/int64 table": ("<i8", (7, 3), (0, 1)),
    "extra/never written": ("<f4", (7, 3), (0, 1)), "extra/uses the committed type": ("<i2", (4,), (0, 1)), "records": (REC, (5,), (0, 1)), "idx/records": (REC, (5,), (2,)),
}
for nm in ("paged", "paged gz", "implicit"):
    DATASETS["idx/" + nm] = ("<u2", (50, 45), (2,))
for q in range(40):
    DATASETS[f"table {q:02d}"] = ("<u2", (q % 5,), (0, 1) if q < 3 else (0,))


def link_names(n):
    """make_flash_h5.c, mode n >= 100"""
    return [f"link {q:06d} {'x' if q % 3 else 'a longer name than the others'}" for q in range(n)]


def main():
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "make_flash_h5")
        subprocess.check_call(["gcc", "-O1", "-o", exe, os.path.join(HERE, "make_flash_h5.c"), f"-I{PREFIX}/include",
                               f"-L{PREFIX}/lib", "-lhdf5", f"-Wl,-rpath,{PREFIX}/lib"])
        expected = {}
        for stem, mode in MODES.items():
            path = os.path.join(HERE, stem + ".h5")
            subprocess.check_call([exe, path, str(mode)])
            for name, (dtype, shape, modes) in DATASETS.items():
                if mode not in modes:
                    continue
                n = int(np.prod(shape))
                if n == 0:
                    expected[f"{stem}:{name}"] = np.zeros(shape, dtype)
                    continue
                out = os.path.join(tmp, "d.bin")
                # h5dump writes compound records only in their memory layout (NATIVE; x86-64: little-endian, 84 / 88 bytes)
                order = "NATIVE" if np.dtype(dtype).kind in "VS" else "LE"  # (strings likewise)
                subprocess.check_call([f"{PREFIX}/bin/h5dump", "-d", "/" + name, "-b", order, "-o", out, path],
                                      stdout=subprocess.DEVNULL)
                raw = open(out, "rb").read()
                a = np.frombuffer(raw, dtype=dtype)
                assert a.size == n, (stem, name, a.size, n)
                expected[f"{stem}:{name}"] = a.reshape(shape).copy()
        subprocess.check_call([exe, os.path.join(HERE, "dense_links.h5"), "3"])
        subprocess.check_call([exe, os.path.join(HERE, "dense_many.h5"), "700"])  # 700 hard links: a B-tree of depth 1, 18 heap blocks
        # larger groups are checked here and not committed (0.3 / 1.9 MB): a version-2 B-tree of depth 2 / 3, nested indirect heap blocks
        sys.path.insert(0, os.path.join(HERE, "..", "..", ".."))
        from synthpy_amd.utils import hdf5_lite
        for n in (6000, 40000):
            big = os.path.join(tmp, f"many_{n}.h5")
            subprocess.check_call([exe, big, str(n)])
            with hdf5_lite.File(big) as f:
                names = f["many"].keys()
                assert sorted(names) == sorted(link_names(n)), n
                assert list(f["many"][names[n // 3]][...]) == [7, 8, 9]
            print(f"dense group of {n} links: read back by hdf5_lite")
        np.savez_compressed(os.path.join(HERE, "expected.npz"), **expected)
        ver = subprocess.run([f"{PREFIX}/bin/h5dump", "--version"], capture_output=True, text=True).stdout.strip()
        open(os.path.join(HERE, "README.txt"), "w").write(
            "HDF5 fixtures written by make_flash_h5.c through the HDF5 library (" + ver + "), expected.npz read back from them by\n"
            "that library's h5dump (make_fixtures.py).  flash_default.h5: the library's defaults, as FLASH writes; flash_repacked.h5:\n"
            "chunked + shuffle + gzip + fletcher32, big-endian members; flash_latest.h5: H5F_LIBVER_LATEST structures; dense_links.h5, dense_many.h5:\n"
            "latest-format groups of 10 / 700 objects (dense link storage: fractal heap + version-2 B-tree; one object with 12 attributes).\n")
        print(len(expected), "datasets read back;", ver)


if __name__ == "__main__":
    main()
