"""The reference's pair dump as a REAL HDF5 file, written by h5py ITSELF with the statements of the reference's dumper
(dump/dumper/base_dumper.py:85-111) - the image holds an Anaconda interpreter with h5py 3.3 beside the system Python (which has torch
but no h5py):

    python tools/make_h5_fixture_h5py.py          # system Python: exports the records, then re-runs itself under /opt/conda/bin/python3.9

-> tests/golden/reader_dump.hdf5: the records of tests/helpers.make_reader_records (the ones the reference's own reader was run on:
tools/make_golden.py case_reader -> reader_standard.npz) in the reference's dump layout, file-format defaults of h5py (libver 'earliest').  Test infrastructure only: the product never runs that interpreter."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONDA = os.environ.get('H5PY_PYTHON', '/opt/conda/bin/python3.9')
TYPES = ('K1', 'K2', 'R', 'T', 'e', 'f')


def export(tmp):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, ROOT)
    from helpers import load_golden, make_reader_records
    recs = make_reader_records(load_golden('reader_standard')[0]['seed'])
    flat = {'n': np.array(len(recs))}
    for i, r in enumerate(recs):
        for k, v in r.items():
            flat[f'{k}_{i}'] = np.asarray(v)
    np.savez(tmp, **flat)


def write(tmp, out):
    import h5py
    z = np.load(tmp)
    n = int(z['n'])
    with h5py.File(out, 'w') as f:                                        # base_dumper.py:85
        for type in TYPES:                                                # :86-91
            dg = f.create_group(type)
            for idx in range(n):
                data_item = np.asarray(z[f'{type}_{idx}'])
                dg.create_dataset(str(idx), data_item.shape, data_item.dtype, data=data_item)
        for type in ('img_path1', 'img_path2'):                           # :92-98
            dg = f.create_group(type)
            for idx in range(n):
                dg.create_dataset(str(idx), [1], h5py.string_dtype(encoding='ascii'), data=str(z[f'{type}_{idx}']).encode('ascii'))
        desc1_g, desc2_g, kpt1_g, kpt2_g = f.create_group('desc1'), f.create_group('desc2'), f.create_group('kpt1'), f.create_group('kpt2')   # :100-111
        for idx in range(n):
            for g, k in ((desc1_g, 'desc1'), (desc2_g, 'desc2'), (kpt1_g, 'kpt1'), (kpt2_g, 'kpt2')):
                a = np.asarray(z[f'{k}_{idx}'])
                g.create_dataset(str(idx), a.shape, a.dtype, data=a)
    print(out, os.path.getsize(out), 'bytes, h5py', h5py.__version__, 'HDF5', h5py.version.hdf5_version)


if __name__ == '__main__':
    tmp = '/tmp/reader_records.npz'
    out = os.path.join(ROOT, 'tests', 'golden', 'reader_dump.hdf5')
    if len(sys.argv) > 1 and sys.argv[1] == '--write':
        write(tmp, out)
    else:
        export(tmp)
        subprocess.run([CONDA, os.path.abspath(__file__), '--write'], check=True, env={k: v for k, v in os.environ.items() if not k.startswith('PYTHON')})
