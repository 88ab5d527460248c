/* Fixture generator (build container only): writes tests/golden/libhdf5_written_{earliest,v18}.h5 with the REAL HDF5
 * library (gcc -I/opt/conda/include make_hdf5_fixture.c -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib; ./a.out FILE 0|1):
 * the files h5lite's reader is checked against. */
#include "hdf5.h"
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "libhdf5_written.h5";
    int latest = argc > 2 && atoi(argv[2]);
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    if (latest) H5Pset_libver_bounds(fapl, H5F_LIBVER_V18, H5F_LIBVER_V18);
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    hid_t g = H5Gcreate2(f, "/Mesh", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t g0 = H5Gcreate2(g, "0", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    hid_t gm = H5Gcreate2(g0, "mesh", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    /* contiguous doubles */
    double geom[11][2];
    for (int i = 0; i < 11; ++i) { geom[i][0] = 0.1 * i; geom[i][1] = -0.5 * i * i; }
    hsize_t d2[2] = {11, 2};
    hid_t sp = H5Screate_simple(2, d2, NULL);
    hid_t ds = H5Dcreate2(gm, "geometry", H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, geom);
    H5Dclose(ds); H5Sclose(sp);
    /* contiguous int64 with attributes: fixed string, uint64 array, double scalar, vlen string */
    long long topo[10][2];
    for (int i = 0; i < 10; ++i) { topo[i][0] = i; topo[i][1] = i + 1; }
    hsize_t d3[2] = {10, 2};
    sp = H5Screate_simple(2, d3, NULL);
    ds = H5Dcreate2(gm, "topology", H5T_NATIVE_LLONG, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_LLONG, H5S_ALL, H5S_ALL, H5P_DEFAULT, topo);
    hid_t st = H5Tcopy(H5T_C_S1); H5Tset_size(st, 8);
    hid_t as = H5Screate(H5S_SCALAR);
    hid_t at = H5Acreate2(ds, "celltype", st, as, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(at, st, "interval"); H5Aclose(at);
    unsigned long long part[2] = {0, 10}; hsize_t pd[1] = {2};
    hid_t ps = H5Screate_simple(1, pd, NULL);
    at = H5Acreate2(ds, "partition", H5T_NATIVE_ULLONG, ps, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(at, H5T_NATIVE_ULLONG, part); H5Aclose(at);
    double tv = 2.75;
    at = H5Acreate2(ds, "time", H5T_NATIVE_DOUBLE, as, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(at, H5T_NATIVE_DOUBLE, &tv); H5Aclose(at);
    hid_t vt = H5Tcopy(H5T_C_S1); H5Tset_size(vt, H5T_VARIABLE);
    const char *vs = "a variable-length string";
    at = H5Acreate2(ds, "note", vt, as, H5P_DEFAULT, H5P_DEFAULT);
    H5Awrite(at, vt, &vs); H5Aclose(at);
    H5Dclose(ds); H5Sclose(sp);
    /* 21 datasets in one group: more links than one symbol-table node holds */
    hid_t gv = H5Gcreate2(f, "/VisualisationVector", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    for (int k = 0; k < 21; ++k) {
        char nm[16]; sprintf(nm, "%d", k);
        double v[11][1];
        for (int i = 0; i < 11; ++i) v[i][0] = k + 0.01 * i;
        hsize_t d1[2] = {11, 1};
        sp = H5Screate_simple(2, d1, NULL);
        ds = H5Dcreate2(gv, nm, H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        H5Dwrite(ds, H5T_NATIVE_DOUBLE, H5S_ALL, H5S_ALL, H5P_DEFAULT, v);
        H5Dclose(ds); H5Sclose(sp);
    }
    /* chunked, chunked + shuffle + deflate, compact, float32, int32, big-endian */
    float big[37][5];
    for (int i = 0; i < 37; ++i) for (int j = 0; j < 5; ++j) big[i][j] = (float)(i * 5 + j) * 0.5f;
    hsize_t db[2] = {37, 5}, ch[2] = {8, 3};
    sp = H5Screate_simple(2, db, NULL);
    hid_t pl = H5Pcreate(H5P_DATASET_CREATE); H5Pset_chunk(pl, 2, ch);
    ds = H5Dcreate2(f, "/chunked_f32", H5T_NATIVE_FLOAT, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, big); H5Dclose(ds);
    H5Pset_shuffle(pl); H5Pset_deflate(pl, 6);
    ds = H5Dcreate2(f, "/chunked_deflate_f32", H5T_NATIVE_FLOAT, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_FLOAT, H5S_ALL, H5S_ALL, H5P_DEFAULT, big); H5Dclose(ds);
    H5Pclose(pl); H5Sclose(sp);
    int small[6] = {3, -1, 4, -1, 5, -9}; hsize_t ds6[1] = {6};
    sp = H5Screate_simple(1, ds6, NULL);
    pl = H5Pcreate(H5P_DATASET_CREATE); H5Pset_layout(pl, H5D_COMPACT);
    ds = H5Dcreate2(f, "/compact_i32", H5T_NATIVE_INT, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, small); H5Dclose(ds); H5Pclose(pl);
    ds = H5Dcreate2(f, "/bigendian_i32", H5T_STD_I32BE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dwrite(ds, H5T_NATIVE_INT, H5S_ALL, H5S_ALL, H5P_DEFAULT, small); H5Dclose(ds); H5Sclose(sp);
    /* an allocated-late, never written dataset */
    sp = H5Screate_simple(1, ds6, NULL);
    ds = H5Dcreate2(f, "/never_written", H5T_NATIVE_DOUBLE, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Dclose(ds); H5Sclose(sp);
    at = H5Acreate2(f, "generator", st, as, H5P_DEFAULT, H5P_DEFAULT); H5Awrite(at, st, "libhdf5"); H5Aclose(at);
    H5Gclose(gv); H5Gclose(gm); H5Gclose(g0); H5Gclose(g); H5Fclose(f);
    return 0;
}
