"""Writes REAL HDF5 files in the reference's dump layout with the HDF5 C library itself (libhdf5 through ctypes - what h5py wraps;
h5py is not installed here), for the tests of imp_release_amd.h5lite / data.H5PairStore:

    python tools/make_h5_fixture.py            ->  tests/golden/reader_dump_latest.hdf5, tests/golden/reader_dump_chunked.hdf5

(The dump itself, tests/golden/reader_dump.hdf5, is written by h5py with the statements of the reference's dumper:
tools/make_h5_fixture_h5py.py.  This script covers the corners of the format that dump does not touch.)
* reader_dump_latest.hdf5: libver='latest' (superblock 3, version-2 object headers, compact link messages, a dense group), the
  version-4 layout message with the single-chunk, fixed-array (plain, paged, filtered) and implicit chunk indexes, a compact dataset,
  fixed- and variable-length strings, integer types.
* reader_dump_chunked.hdf5: a default (libver 'earliest') file with chunked datasets (version-1 B-tree chunk index, shuffle +
  deflate, edge chunks, an unfiltered one) and a group large enough for several symbol-table nodes.

Needs a libhdf5 shared library (HDF5_LIB=<path>, default: the one under /opt/conda/lib in the build image).  The product never
loads it: the fixtures are data."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, ROOT)

LIB = os.environ.get('HDF5_LIB', '/opt/conda/lib/libhdf5.so.103')
L = C.CDLL(LIB)
hid = C.c_int64
L.H5open()
for fn, res, args in [
    ('H5Fcreate', hid, [C.c_char_p, C.c_uint, hid, hid]), ('H5Fclose', C.c_int, [hid]),
    ('H5Gcreate2', hid, [hid, C.c_char_p, hid, hid, hid]), ('H5Gclose', C.c_int, [hid]),
    ('H5Screate_simple', hid, [C.c_int, C.POINTER(C.c_uint64), C.c_void_p]), ('H5Sclose', C.c_int, [hid]),
    ('H5Dcreate2', hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ('H5Dwrite', C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
    ('H5Dclose', C.c_int, [hid]), ('H5Tcopy', hid, [hid]), ('H5Tset_size', C.c_int, [hid, C.c_size_t]), ('H5Tclose', C.c_int, [hid]),
    ('H5Pcreate', hid, [hid]), ('H5Pclose', C.c_int, [hid]), ('H5Pset_libver_bounds', C.c_int, [hid, C.c_int, C.c_int]),
    ('H5Pset_chunk', C.c_int, [hid, C.c_int, C.POINTER(C.c_uint64)]), ('H5Pset_deflate', C.c_int, [hid, C.c_uint]),
    ('H5Pset_shuffle', C.c_int, [hid]), ('H5Pset_layout', C.c_int, [hid, C.c_int]), ('H5Pset_alloc_time', C.c_int, [hid, C.c_int]),
    ('H5Pset_fletcher32', C.c_int, [hid]),
]:
    f = getattr(L, fn)
    f.restype, f.argtypes = res, args
G = lambda name: hid.in_dll(L, name).value      # noqa: E731
FILE_T = {np.dtype('float32'): G('H5T_IEEE_F32LE_g'), np.dtype('float64'): G('H5T_IEEE_F64LE_g'), np.dtype('int64'): G('H5T_STD_I64LE_g'),
          np.dtype('int32'): G('H5T_STD_I32LE_g'), np.dtype('uint8'): G('H5T_STD_U8LE_g'), np.dtype('int16'): G('H5T_STD_I16LE_g')}
MEM_T = {np.dtype('float32'): G('H5T_NATIVE_FLOAT_g'), np.dtype('float64'): G('H5T_NATIVE_DOUBLE_g'), np.dtype('int64'): G('H5T_NATIVE_INT64_g'),
         np.dtype('int32'): G('H5T_NATIVE_INT32_g'), np.dtype('uint8'): G('H5T_NATIVE_UINT8_g'), np.dtype('int16'): G('H5T_NATIVE_INT16_g')}
H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 2, 0, 0


def ok(v, what):
    if v < 0:
        raise RuntimeError(f'libhdf5: {what} failed')
    return v


def write_array(loc, name, a, dcpl=H5P_DEFAULT):
    a = np.ascontiguousarray(a)
    dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
    sp = ok(L.H5Screate_simple(a.ndim, dims, None), 'H5Screate_simple')
    ds = ok(L.H5Dcreate2(loc, name.encode(), FILE_T[a.dtype], sp, H5P_DEFAULT, dcpl, H5P_DEFAULT), f'H5Dcreate2 {name}')
    ok(L.H5Dwrite(ds, MEM_T[a.dtype], H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)), 'H5Dwrite')
    L.H5Dclose(ds); L.H5Sclose(sp)


def write_vlen_strings(loc, name, strings):
    """shape [len(strings)] of variable-length ASCII strings = dg.create_dataset(name, [1], h5py.string_dtype(encoding='ascii'), data=...)"""
    t = ok(L.H5Tcopy(G('H5T_C_S1_g')), 'H5Tcopy')
    ok(L.H5Tset_size(t, C.c_size_t(-1).value), 'H5Tset_size(H5T_VARIABLE)')
    dims = (C.c_uint64 * 1)(len(strings))
    sp = ok(L.H5Screate_simple(1, dims, None), 'H5Screate_simple')
    ds = ok(L.H5Dcreate2(loc, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f'H5Dcreate2 {name}')
    buf = (C.c_char_p * len(strings))(*[s.encode('ascii') for s in strings])
    ok(L.H5Dwrite(ds, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, C.cast(buf, C.c_void_p)), 'H5Dwrite')
    L.H5Dclose(ds); L.H5Sclose(sp); L.H5Tclose(t)


def write_fixed_string(loc, name, s):
    t = ok(L.H5Tcopy(G('H5T_C_S1_g')), 'H5Tcopy')
    ok(L.H5Tset_size(t, len(s)), 'H5Tset_size')
    dims = (C.c_uint64 * 1)(1)
    sp = ok(L.H5Screate_simple(1, dims, None), 'H5Screate_simple')
    ds = ok(L.H5Dcreate2(loc, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f'H5Dcreate2 {name}')
    ok(L.H5Dwrite(ds, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, C.cast(C.c_char_p(s.encode('ascii')), C.c_void_p)), 'H5Dwrite')
    L.H5Dclose(ds); L.H5Sclose(sp); L.H5Tclose(t)


def main():
    from helpers import load_golden, make_reader_records
    spec, _ = load_golden('reader_standard')
    recs = make_reader_records(spec['seed'])
    # ---- the other corners ----
    out2 = os.path.join(ROOT, 'tests', 'golden', 'reader_dump_latest.hdf5')
    fapl = ok(L.H5Pcreate(G('H5P_CLS_FILE_ACCESS_ID_g')), 'H5Pcreate')
    ok(L.H5Pset_libver_bounds(fapl, 2, 2), 'H5Pset_libver_bounds(latest)')
    f = ok(L.H5Fcreate(out2.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, fapl), 'H5Fcreate')
    r = recs[0]
    g = ok(L.H5Gcreate2(f, b'pair', H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2')
    write_array(g, 'K1', np.asarray(r['K1']))
    write_array(g, 'kpt1', np.asarray(r['kpt1']))
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')
    ramp = (np.arange(100 * 37, dtype=np.float32).reshape(100, 37) % 17).astype(np.float32)      # (compressible; chunks do not divide the shape)
    ok(L.H5Pset_chunk(dcpl, 2, (C.c_uint64 * 2)(32, 16)), 'H5Pset_chunk')
    ok(L.H5Pset_shuffle(dcpl), 'H5Pset_shuffle'); ok(L.H5Pset_deflate(dcpl, 4), 'H5Pset_deflate')
    write_array(g, 'ramp_chunked_shuffle_deflate', ramp, dcpl)
    L.H5Pclose(dcpl)
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')
    ok(L.H5Pset_chunk(dcpl, 2, (C.c_uint64 * 2)(100, 37)), 'H5Pset_chunk')
    ok(L.H5Pset_deflate(dcpl, 6), 'H5Pset_deflate')
    write_array(g, 'ramp_single_chunk_deflate', ramp, dcpl)
    L.H5Pclose(dcpl)
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')
    ok(L.H5Pset_layout(dcpl, 0), 'H5Pset_layout(compact)')
    write_array(g, 'ids_compact_i32', np.arange(-5, 20, dtype=np.int32), dcpl)
    L.H5Pclose(dcpl)
    write_array(g, 'counts_i64', np.array([[1, -2, 3], [2 ** 40, 5, -6]], dtype=np.int64))
    write_fixed_string(g, 'path_fixed', r['img_path1'])
    write_vlen_strings(g, 'paths_vlen', [r['img_path1'], r['img_path2'], ''])
    L.H5Gclose(g)                                                         # (8 links: still compact link messages; a 9th makes the group dense)
    g = ok(L.H5Gcreate2(f, b'more', H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2')
    write_array(g, 'bytes_u8', np.arange(200, dtype=np.uint8))
    write_array(g, 'shorts_i16', np.array([-300, 7, 300], dtype=np.int16))
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')                      # 1500 chunks: a PAGED fixed array (1024 elements per page)
    ok(L.H5Pset_chunk(dcpl, 1, (C.c_uint64 * 1)(2)), 'H5Pset_chunk')
    write_array(g, 'ids_1500_chunks', np.arange(3000, dtype=np.int32) * 3, dcpl)
    L.H5Pclose(dcpl)
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')                      # early allocation, no filter: the implicit index
    ok(L.H5Pset_chunk(dcpl, 2, (C.c_uint64 * 2)(8, 5)), 'H5Pset_chunk')
    ok(L.H5Pset_alloc_time(dcpl, 1), 'H5Pset_alloc_time(early)')
    write_array(g, 'grid_implicit', np.arange(20 * 12, dtype=np.float64).reshape(20, 12), dcpl)
    L.H5Pclose(dcpl)
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')                      # filtered + paged + checksummed chunks
    ok(L.H5Pset_chunk(dcpl, 1, (C.c_uint64 * 1)(4)), 'H5Pset_chunk')
    ok(L.H5Pset_deflate(dcpl, 1), 'H5Pset_deflate'); ok(L.H5Pset_fletcher32(dcpl), 'H5Pset_fletcher32')
    write_array(g, 'ids_1100_chunks_deflate_fletcher', np.arange(4400, dtype=np.int16), dcpl)
    L.H5Pclose(dcpl)
    L.H5Gclose(g)
    g = ok(L.H5Gcreate2(f, b'dense', H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2')     # 12 links: fractal heap (the decoder must say so)
    for i in range(12):
        write_array(g, str(i), np.array([i], dtype=np.int32))
    L.H5Gclose(g)
    L.H5Fclose(f); L.H5Pclose(fapl)
    print(out2, os.path.getsize(out2), 'bytes')

    # the same chunked dataset in a default (libver earliest) file: version-3 layout message, version-1 B-tree chunk index, version-1 filter pipeline
    out3 = os.path.join(ROOT, 'tests', 'golden', 'reader_dump_chunked.hdf5')
    f = ok(L.H5Fcreate(out3.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), 'H5Fcreate')
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')
    ok(L.H5Pset_chunk(dcpl, 2, (C.c_uint64 * 2)(32, 16)), 'H5Pset_chunk')
    ok(L.H5Pset_shuffle(dcpl), 'H5Pset_shuffle'); ok(L.H5Pset_deflate(dcpl, 4), 'H5Pset_deflate')
    write_array(f, 'ramp', ramp, dcpl)
    L.H5Pclose(dcpl)
    dcpl = ok(L.H5Pcreate(G('H5P_CLS_DATASET_CREATE_ID_g')), 'H5Pcreate')
    ok(L.H5Pset_chunk(dcpl, 1, (C.c_uint64 * 1)(7)), 'H5Pset_chunk')
    write_array(f, 'ids_chunked_unfiltered', np.arange(50, dtype=np.int64), dcpl)
    L.H5Pclose(dcpl)
    g = ok(L.H5Gcreate2(f, b'many', H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), 'H5Gcreate2')      # > one symbol-table node: a B-tree with several leaves
    for i in range(40):
        write_array(g, str(i), np.array([i, i * i], dtype=np.int32))
    L.H5Gclose(g)
    L.H5Fclose(f)
    print(out3, os.path.getsize(out3), 'bytes')


if __name__ == '__main__':
    main()
