"""Point-cloud file readers (interface of utils/io_module.py of the reference: `IO.get(path)`).

.npy / .txt / .dat as in the reference; .pcd is parsed here directly (ASCII, binary and
binary_compressed are not all needed: the PCN files and the reference's demo/samples are ASCII or
uncompressed binary) because open3d is not installed; .h5 needs h5py and raises if it is absent.
"""
import os
import pickle

import numpy as np


def read_pcd(path):
    """x/y/z of a .pcd file (FIELDS containing x y z; DATA ascii | binary) -> float64 [N,3]."""
    with open(path, 'rb') as fh:
        header = {}
        while True:
            line = fh.readline()
            if not line:
                raise ValueError(f'{path}: no DATA line in PCD header')
            text = line.decode('ascii', errors='replace').strip()
            if not text or text.startswith('#'):
                continue
            key, _, val = text.partition(' ')
            header[key.upper()] = val.split()
            if key.upper() == 'DATA':
                break
        fields = [f.lower() for f in header.get('FIELDS', [])]
        sizes = [int(s) for s in header.get('SIZE', ['4'] * len(fields))]
        types = header.get('TYPE', ['F'] * len(fields))
        counts = [int(c) for c in header.get('COUNT', ['1'] * len(fields))]
        npts = int(header.get('POINTS', header.get('WIDTH', ['0']))[0])
        kind = header['DATA'][0].lower()
        if not all(a in fields for a in ('x', 'y', 'z')):
            raise ValueError(f'{path}: PCD without x y z fields')
        if kind == 'ascii':
            cols, off = {}, 0
            for f, c in zip(fields, counts):
                cols[f] = off
                off += c
            data = np.loadtxt(fh, dtype=np.float64, ndmin=2)
            data = data.reshape(-1, off) if data.size else np.zeros((0, off))
            return data[:, [cols['x'], cols['y'], cols['z']]]
        if kind == 'binary':
            np_types = {('F', 4): '<f4', ('F', 8): '<f8', ('U', 1): 'u1', ('U', 2): '<u2', ('U', 4): '<u4',
                        ('I', 1): 'i1', ('I', 2): '<i2', ('I', 4): '<i4'}
            dt = np.dtype([(f, np_types[(t.upper(), s)], (c,)) if c > 1 else (f, np_types[(t.upper(), s)])
                           for f, t, s, c in zip(fields, types, sizes, counts)])
            raw = np.frombuffer(fh.read(npts * dt.itemsize), dtype=dt, count=npts)
            return np.stack([raw['x'], raw['y'], raw['z']], axis=1).astype(np.float64)
        raise ValueError(f'{path}: PCD DATA {kind} is not supported (ascii and binary are)')


class IO:
    @classmethod
    def get(cls, file_path):
        ext = os.path.splitext(file_path)[1]
        if ext == '.npy':
            return np.load(file_path)
        if ext in ('.pcd',):
            return read_pcd(file_path)
        if ext == '.h5':
            import h5py
            with h5py.File(file_path, 'r') as f:
                return f['data'][()]
        if ext == '.txt':
            return np.loadtxt(file_path)
        if ext == '.dat':
            with open(file_path, 'rb') as f:
                return pickle.load(f)
        raise Exception('Unsupported file extension: %s' % ext)
