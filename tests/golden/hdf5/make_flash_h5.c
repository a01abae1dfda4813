/* Writes the HDF5 fixtures of tests/test_filetypes.py with the HDF5 library itself (1.10.6 found under /opt/conda in the build
 * container; see make_fixtures.py beside this file, which compiles and runs it and then has h5dump read every dataset back).
 * The files follow the layout of a FLASH plot file as the reference's hdf_readin meets it through yt
 * (reference src/utils/handle_filetypes.py:121-150): per-block tables `bounding box`, `refine level`, `node type`, ... and one
 * (blocks, nzb, nyb, nxb) dataset per variable, named by four characters (`dens`, `ye  `, `sumy`).
 *
 *   make_flash_h5 <out.h5> <mode>     mode 0: the library's defaults, contiguous datasets (what FLASH writes)
 *                                     mode 1: variables chunked + shuffle + gzip + fletcher32, tables chunked (a repacked file)
 *                                     mode 2: H5F_LIBVER_LATEST (superblock 3, version-2 object headers), few objects
 *                                     mode 3: H5F_LIBVER_LATEST, ten groups in the root group and nothing else
 *                                     mode n >= 100: H5F_LIBVER_LATEST, a group of n hard links to one dataset
 * The AMR tree: 2 x 2 x 1 root blocks of 4 x 2 x 3 cells on [0,2] x [0,1] x [-0.75,0.75]; root block 0 is refined once.
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NXB 4
#define NYB 2
#define NZB 3
#define NBLK 12

static double lo[NBLK][3], hi[NBLK][3];
static int level[NBLK], ntype[NBLK];

static void tree(void) {
  int b = 0;
  for (int j = 0; j < 2; ++j)
    for (int i = 0; i < 2; ++i, ++b) {
      lo[b][0] = i, hi[b][0] = i + 1.0;
      lo[b][1] = 0.5 * j, hi[b][1] = 0.5 * (j + 1);
      lo[b][2] = -0.75, hi[b][2] = 0.75;
      level[b] = 1;
      ntype[b] = b == 0 ? 2 : 1;
    }
  for (int k = 0; k < 2; ++k)
    for (int j = 0; j < 2; ++j)
      for (int i = 0; i < 2; ++i, ++b) {
        lo[b][0] = 0.5 * i, hi[b][0] = 0.5 * (i + 1);
        lo[b][1] = 0.25 * j, hi[b][1] = 0.25 * (j + 1);
        lo[b][2] = -0.75 + 0.75 * k, hi[b][2] = -0.75 + 0.75 * (k + 1);
        level[b] = 2;
        ntype[b] = 1;
      }
}

static double var(int which, int b, int k, int j, int i) {
  if (ntype[b] != 1) return -999.0;  /* a parent's cells: must never reach the covering grid */
  const double x = lo[b][0] + (i + 0.5) * (hi[b][0] - lo[b][0]) / NXB;
  const double y = lo[b][1] + (j + 0.5) * (hi[b][1] - lo[b][1]) / NYB;
  const double z = lo[b][2] + (k + 0.5) * (hi[b][2] - lo[b][2]) / NZB;
  if (which == 0) return 1.0 + x + 10.0 * y + 100.0 * z * z;
  if (which == 1) return 0.4 + 0.1 * x;
  return 0.9 + 0.05 * y - 0.01 * z;
}

typedef struct { char name[80]; int value; } int_rec;
typedef struct { char name[80]; double value; } real_rec;

static void pad(char *dst, const char *s, size_t n) {  /* FLASH pads its names with blanks, no terminator */
  memset(dst, ' ', n);
  memcpy(dst, s, strlen(s));
}

#define CHECK(x) do { if ((x) < 0) { fprintf(stderr, "HDF5 call failed: %s (line %d)\n", #x, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
  if (argc != 3) return 2;
  const int mode = atoi(argv[2]);
  tree();
  hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
  if (mode >= 2) CHECK(H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST));
  hid_t f = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
  CHECK(f);

  if (mode >= 100) {  /* `mode` hard links to one small dataset, in a latest-format group: dense link storage at scale (a fractal heap of
                       * several direct blocks under indirect ones, a version-2 B-tree of depth 1 or more) */
    hsize_t one[1] = {3};
    hid_t sp = H5Screate_simple(1, one, NULL);
    hid_t d = H5Dcreate2(f, "target", H5T_STD_I32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    int v[3] = {7, 8, 9};
    CHECK(H5Dwrite(d, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v));
    H5Dclose(d); H5Sclose(sp);
    hid_t g = H5Gcreate2(f, "many", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    for (int q = 0; q < mode; ++q) {
      char nm[64];
      snprintf(nm, sizeof nm, "link %06d %s", q, (q % 3) ? "x" : "a longer name than the others");
      CHECK(H5Lcreate_hard(f, "target", g, nm, H5P_DEFAULT, H5P_DEFAULT));
    }
    H5Gclose(g);
    H5Fclose(f);
    H5Pclose(fapl);
    return 0;
  }
  if (mode == 3) {  /* only this: ten groups in a latest-format root group = dense link storage (a fractal heap + a version-2 B-tree) */
    for (int q = 0; q < 10; ++q) {
      char nm[16];
      snprintf(nm, sizeof nm, "g%d", q);
      hid_t g = H5Gcreate2(f, nm, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(g);
      if (q == 4) {  /* twelve attributes on one object: dense ATTRIBUTE storage */
        hid_t as = H5Screate(H5S_SCALAR);
        for (int a = 0; a < 12; ++a) {
          char an[16];
          snprintf(an, sizeof an, "a%02d", a);
          hid_t at = H5Acreate2(g, an, H5T_STD_I32LE, as, H5P_DEFAULT, H5P_DEFAULT);
          CHECK(H5Awrite(at, H5T_NATIVE_INT, &a));
          H5Aclose(at);
        }
        H5Sclose(as);
      }
      H5Gclose(g);
    }
    H5Fclose(f);
    H5Pclose(fapl);
    return 0;
  }
  /* ---- the variables */
  const char *names[3] = {"dens", "ye  ", "sumy"};
  static float v32[NBLK][NZB][NYB][NXB];
  static double v64[NBLK][NZB][NYB][NXB];
  hsize_t vd[4] = {NBLK, NZB, NYB, NXB};
  for (int w = 0; w < 3; ++w) {
    double mn = 1e300, mx = -1e300;
    for (int b = 0; b < NBLK; ++b)
      for (int k = 0; k < NZB; ++k)
        for (int j = 0; j < NYB; ++j)
          for (int i = 0; i < NXB; ++i) {
            const double q = var(w, b, k, j, i);
            v32[b][k][j][i] = (float)q;
            v64[b][k][j][i] = q;
            if (ntype[b] == 1) { mn = q < mn ? q : mn; mx = q > mx ? q : mx; }
          }
    hid_t sp = H5Screate_simple(4, vd, NULL), dcpl = H5Pcreate(H5P_DATASET_CREATE);
    if (mode == 1) {
      hsize_t ch[4] = {1, 2, 1, 2};  /* 12 * 2 * 2 * 2 = 96 chunks: two levels of the chunk B-tree at the default K = 32 */
      CHECK(H5Pset_chunk(dcpl, 4, ch));
      if (w != 2) CHECK(H5Pset_shuffle(dcpl));
      CHECK(H5Pset_deflate(dcpl, w == 0 ? 9 : 1));
      if (w == 1) CHECK(H5Pset_fletcher32(dcpl));
    }
    if (mode == 2 && w == 1) {
      hsize_t ch[4] = {5, 2, 2, 3};  /* edge chunks stick out of the dataset; fixed-array chunk index */
      CHECK(H5Pset_chunk(dcpl, 4, ch));
      CHECK(H5Pset_deflate(dcpl, 4));
    }
    if (mode == 2 && w == 2) {
      hsize_t ch[4] = {NBLK, NZB, NYB, NXB};  /* one chunk: the "single chunk" index */
      CHECK(H5Pset_chunk(dcpl, 4, ch));
      CHECK(H5Pset_shuffle(dcpl));
      CHECK(H5Pset_deflate(dcpl, 4));
    }
    /* plot files hold float32, checkpoint files float64: sumy is written as float64, big-endian in mode 1 */
    hid_t ftype = w == 2 ? (mode == 1 ? H5T_IEEE_F64BE : H5T_IEEE_F64LE) : H5T_IEEE_F32LE;
    hid_t d = H5Dcreate2(f, names[w], ftype, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    CHECK(d);
    if (w == 2) CHECK(H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, v64));
    else CHECK(H5Dwrite(d, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, v32));
    hid_t as = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(d, "minimum", H5T_IEEE_F64LE, as, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(H5Awrite(a, H5T_NATIVE_DOUBLE, &mn));
    H5Aclose(a);
    a = H5Acreate2(d, "maximum", H5T_IEEE_F64LE, as, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(H5Awrite(a, H5T_NATIVE_DOUBLE, &mx));
    H5Aclose(a);
    H5Sclose(as); H5Dclose(d); H5Pclose(dcpl); H5Sclose(sp);
  }

  /* ---- the block tables */
  static double bb[NBLK][3][2], coord[NBLK][3], bsize[NBLK][3];
  for (int b = 0; b < NBLK; ++b)
    for (int a = 0; a < 3; ++a) {
      bb[b][a][0] = lo[b][a];
      bb[b][a][1] = hi[b][a];
      coord[b][a] = 0.5 * (lo[b][a] + hi[b][a]);
      bsize[b][a] = hi[b][a] - lo[b][a];
    }
  {
    hsize_t d3[3] = {NBLK, 3, 2}, d2[2] = {NBLK, 3}, d1[1] = {NBLK};
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    if (mode == 1) { hsize_t ch[3] = {5, 3, 2}; CHECK(H5Pset_chunk(dcpl, 3, ch)); }  /* 12 = 5 + 5 + 2: a chunk past the edge */
    if (mode == 2) CHECK(H5Pset_layout(dcpl, H5D_COMPACT));                           /* data inside the object header */
    hid_t sp = H5Screate_simple(3, d3, NULL);
    hid_t d = H5Dcreate2(f, "bounding box", H5T_IEEE_F64LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    CHECK(d);
    CHECK(H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, bb));
    H5Dclose(d); H5Sclose(sp); H5Pclose(dcpl);
    if (mode != 2) {
      sp = H5Screate_simple(2, d2, NULL);
      d = H5Dcreate2(f, "coordinates", H5T_IEEE_F64LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, coord));
      H5Dclose(d);
      d = H5Dcreate2(f, "block size", H5T_IEEE_F64LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(H5Dwrite(d, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, bsize));
      H5Dclose(d); H5Sclose(sp);
    }
    sp = H5Screate_simple(1, d1, NULL);
    d = H5Dcreate2(f, "refine level", mode == 1 ? H5T_STD_I32BE : H5T_STD_I32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(H5Dwrite(d, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, level));
    H5Dclose(d);
    d = H5Dcreate2(f, "node type", H5T_STD_I32LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(H5Dwrite(d, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, ntype));
    H5Dclose(d); H5Sclose(sp);
  }

  /* ---- scalars as FLASH writes them: compound (name: 80 blank-padded characters, value) */
  {
    hid_t s80 = H5Tcopy(H5T_C_S1);
    CHECK(H5Tset_size(s80, 80));
    CHECK(H5Tset_strpad(s80, H5T_STR_SPACEPAD));
    const char *in[] = {"nxb", "nyb", "nzb", "dimensionality", "globalnumblocks", "nstep"};
    const int iv[] = {NXB, NYB, NZB, 3, NBLK, 417};
    int_rec ir[6];
    for (int q = 0; q < 6; ++q) { pad(ir[q].name, in[q], 80); ir[q].value = iv[q]; }
    hid_t mt = H5Tcreate(H5T_COMPOUND, sizeof(int_rec));
    CHECK(H5Tinsert(mt, "name", HOFFSET(int_rec, name), s80));
    CHECK(H5Tinsert(mt, "value", HOFFSET(int_rec, value), H5T_NATIVE_INT));
    hsize_t n6[1] = {6};
    hid_t sp = H5Screate_simple(1, n6, NULL);
    hid_t d = H5Dcreate2(f, "integer scalars", mt, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(d);
    CHECK(H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, ir));
    H5Dclose(d); H5Sclose(sp); H5Tclose(mt);
    if (mode != 2) {
      real_rec rr[2];
      pad(rr[0].name, "time", 80); rr[0].value = 1.25e-9;
      pad(rr[1].name, "dt", 80); rr[1].value = 3.5e-13;
      mt = H5Tcreate(H5T_COMPOUND, sizeof(real_rec));
      CHECK(H5Tinsert(mt, "name", HOFFSET(real_rec, name), s80));
      CHECK(H5Tinsert(mt, "value", HOFFSET(real_rec, value), H5T_NATIVE_DOUBLE));
      hsize_t n2[1] = {2};
      sp = H5Screate_simple(1, n2, NULL);
      d = H5Dcreate2(f, "real scalars", mt, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, rr));
      H5Dclose(d); H5Sclose(sp); H5Tclose(mt);
      /* unknown names: (3, 1) strings of 4 characters */
      hid_t s4 = H5Tcopy(H5T_C_S1);
      CHECK(H5Tset_size(s4, 4));
      char un[3][4];
      for (int w = 0; w < 3; ++w) memcpy(un[w], names[w], 4);
      hsize_t nd[2] = {3, 1};
      sp = H5Screate_simple(2, nd, NULL);
      d = H5Dcreate2(f, "unknown names", s4, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(H5Dwrite(d, s4, H5S_ALL, H5S_ALL, H5P_DEFAULT, un));
      H5Dclose(d); H5Sclose(sp); H5Tclose(s4);
      /* filler tables, so that the root group's symbol table spans several nodes in mode 0 (FLASH files hold ~25 objects) */
      for (int q = 0; q < (mode == 0 ? 40 : 3); ++q) {
        char nm[32];
        snprintf(nm, sizeof nm, "table %02d", q);
        hsize_t one[1] = {(hsize_t)(q % 5)};  /* some of them empty */
        sp = H5Screate_simple(1, one, NULL);
        d = H5Dcreate2(f, nm, H5T_STD_U16LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        unsigned short vals[5] = {(unsigned short)q, 1, 2, 3, 65535};
        if (q % 5) CHECK(H5Dwrite(d, H5T_NATIVE_USHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, vals));
        H5Dclose(d); H5Sclose(sp);
      }
      /* a sub-group with one dataset and a never-written dataset with a fill value */
      hid_t g = H5Gcreate2(f, "extra", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      hsize_t n7[2] = {7, 3};
      sp = H5Screate_simple(2, n7, NULL);
      long long big[7][3];
      for (int a = 0; a < 7; ++a) for (int b = 0; b < 3; ++b) big[a][b] = (a - 3) * 4000000000LL + b;
      d = H5Dcreate2(g, "int64 table", H5T_STD_I64LE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(H5Dwrite(d, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, big));
      H5Dclose(d);
      hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
      float fv = 2.5f;
      CHECK(H5Pset_fill_value(dcpl, H5T_NATIVE_FLOAT, &fv));
      d = H5Dcreate2(g, "never written", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
      H5Dclose(d); H5Pclose(dcpl); H5Sclose(sp);
      /* a datatype committed to the file (an object of its own in the group), and a dataset that uses it */
      hid_t ct = H5Tcopy(H5T_STD_I16BE);
      CHECK(H5Tcommit2(g, "int16 big-endian", ct, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
      hsize_t n4[1] = {4};
      sp = H5Screate_simple(1, n4, NULL);
      short sv[4] = {-2, -1, 0, 300};
      d = H5Dcreate2(g, "uses the committed type", ct, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
      CHECK(H5Dwrite(d, H5T_NATIVE_SHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, sv));
      H5Dclose(d); H5Sclose(sp); H5Tclose(ct);
      H5Gclose(g);
    }
    H5Tclose(s80);
  }
  /* ---- mode 2: the other chunk indexes of the latest format, in a sub-group (the root keeps 8 links: compact storage) */
  if (mode == 2) {
    hid_t g = H5Gcreate2(f, "idx", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    static unsigned short tab[50][45];
    for (int a = 0; a < 50; ++a) for (int b = 0; b < 45; ++b) tab[a][b] = (unsigned short)(a * 1000 + b * 7);
    hsize_t dd[2] = {50, 45}, ch1[2] = {1, 1}, ch2[2] = {7, 4};
    hid_t sp = H5Screate_simple(2, dd, NULL);
    for (int q = 0; q < 3; ++q) {
      hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
      CHECK(H5Pset_chunk(dcpl, 2, q == 2 ? ch2 : ch1));           /* 2250 one-element chunks: a paged fixed array (1024 per page) */
      if (q == 1) CHECK(H5Pset_deflate(dcpl, 1));                  /* ... of filtered chunks (address, size, mask) */
      if (q == 2) CHECK(H5Pset_alloc_time(dcpl, H5D_ALLOC_TIME_EARLY));  /* no filter + early allocation: the implicit index */
      const char *nm[3] = {"paged", "paged gz", "implicit"};
      hid_t d = H5Dcreate2(g, nm[q], H5T_STD_U16LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
      CHECK(d);
      if (q == 0) {  /* the first dataset: only every other row is written -- whole pages of the index stay empty */
        hsize_t start[2] = {0, 0}, stride[2] = {1, 1}, count[2] = {20, 45}, blk[2] = {1, 1};
        hid_t fs = H5Dget_space(d), ms = H5Screate_simple(2, count, NULL);
        static unsigned short part[20][45];
        memcpy(part, tab, sizeof part);
        CHECK(H5Sselect_hyperslab(fs, H5S_SELECT_SET, start, stride, count, blk));
        CHECK(H5Dwrite(d, H5T_NATIVE_USHORT, ms, fs, H5P_DEFAULT, part));
        H5Sclose(fs); H5Sclose(ms);
      } else {
        CHECK(H5Dwrite(d, H5T_NATIVE_USHORT, H5S_ALL, H5S_ALL, H5P_DEFAULT, tab));
      }
      H5Dclose(d); H5Pclose(dcpl);
    }
    H5Sclose(sp); H5Gclose(g);
  }
  /* ---- a table of records with an array member and an enumeration member (datatype classes 10 and 8), in every mode */
  {
    typedef struct { unsigned char id; float vec[3]; unsigned char flag; short pad_free; } rec_t;
    rec_t recs[5];
    memset(recs, 0, sizeof recs);
    for (int q = 0; q < 5; ++q) {
      recs[q].id = (unsigned char)(200 + q);
      for (int a = 0; a < 3; ++a) recs[q].vec[a] = 0.5f * q - a;
      recs[q].flag = (unsigned char)(q & 1);
      recs[q].pad_free = (short)(-1000 * q);
    }
    hsize_t three[1] = {3}, five[1] = {5};
    hid_t arr = H5Tarray_create2(H5T_NATIVE_FLOAT, 1, three);
    hid_t en = H5Tenum_create(H5T_NATIVE_UCHAR);
    unsigned char ev = 0;
    CHECK(H5Tenum_insert(en, "OFF", &ev));
    ev = 1;
    CHECK(H5Tenum_insert(en, "SWITCHED_ON", &ev));
    hid_t mt = H5Tcreate(H5T_COMPOUND, sizeof(rec_t));
    CHECK(H5Tinsert(mt, "id", HOFFSET(rec_t, id), H5T_NATIVE_UCHAR));
    CHECK(H5Tinsert(mt, "vec", HOFFSET(rec_t, vec), arr));
    CHECK(H5Tinsert(mt, "flag", HOFFSET(rec_t, flag), en));
    CHECK(H5Tinsert(mt, "last", HOFFSET(rec_t, pad_free), H5T_NATIVE_SHORT));
    hid_t sp = H5Screate_simple(1, five, NULL);
    hid_t loc = mode == 2 ? H5Gopen2(f, "idx", H5P_DEFAULT) : f;  /* (mode 2: the root group must keep to 8 links) */
    hid_t d = H5Dcreate2(loc, "records", mt, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(d);
    CHECK(H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, recs));
    if (mode == 2) H5Gclose(loc);
    H5Dclose(d); H5Sclose(sp); H5Tclose(mt); H5Tclose(en); H5Tclose(arr);
  }
  /* file-level attributes: a string and an int array */
  {
    hid_t st = H5Tcopy(H5T_C_S1);
    CHECK(H5Tset_size(st, 16));
    hid_t as = H5Screate(H5S_SCALAR);
    hid_t a = H5Acreate2(f, "setup", st, as, H5P_DEFAULT, H5P_DEFAULT);
    char txt[16] = "laser_slab";
    CHECK(H5Awrite(a, st, txt));
    H5Aclose(a); H5Sclose(as); H5Tclose(st);
    hsize_t n3[1] = {3};
    as = H5Screate_simple(1, n3, NULL);
    a = H5Acreate2(f, "block cells", H5T_STD_I32LE, as, H5P_DEFAULT, H5P_DEFAULT);
    int nb[3] = {NXB, NYB, NZB};
    CHECK(H5Awrite(a, H5T_NATIVE_INT, nb));
    H5Aclose(a); H5Sclose(as);
  }
  H5Fclose(f);
  H5Pclose(fapl);
  return 0;
}
