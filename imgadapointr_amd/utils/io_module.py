"""Point-cloud file readers (interface of utils/io_module.py of the reference: `IO.get(path)`).

.npy / .txt / .dat as in the reference; .pcd is parsed here directly (DATA ascii, binary and PCL's
binary_compressed = LZF over a field-major block) because open3d is not installed; .h5 needs h5py and raises if it
is absent.
"""
import os
import pickle

import numpy as np


def lzf_decompress(src, out_len):
    """liblzf stream -> bytes.  Control byte c < 32: c + 1 literal bytes follow; otherwise a back reference of
    (c >> 5) + 2 bytes (length 7 is extended by the next byte) to offset ((c & 31) << 8 | next byte) + 1 behind the
    output position; references may overlap their own output (run-length use)."""
    out = bytearray(out_len)
    i, o, n = 0, 0, len(src)
    while i < n:
        c = src[i]
        i += 1
        if c < 32:
            run = c + 1
            if i + run > n or o + run > out_len:
                raise ValueError('corrupt LZF stream (literal run past the end)')
            out[o:o + run] = src[i:i + run]
            i += run
            o += run
            continue
        length = c >> 5
        if length == 7:
            length += src[i]
            i += 1
        ref = o - (((c & 31) << 8) | src[i]) - 1
        i += 1
        length += 2
        if ref < 0 or o + length > out_len:
            raise ValueError('corrupt LZF stream (bad back reference)')
        if ref + length <= o:
            out[o:o + length] = out[ref:ref + length]
        else:                                           # overlapping copy: byte by byte
            for k in range(length):
                out[o + k] = out[ref + k]
        o += length
    if o != out_len:
        raise ValueError(f'LZF stream decoded to {o} bytes, header says {out_len}')
    return bytes(out)


def read_pcd(path):
    """x/y/z of a .pcd file (FIELDS containing x y z; DATA ascii | binary | binary_compressed) -> float64 [N,3]."""
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
        np_types = {('F', 4): '<f4', ('F', 8): '<f8', ('U', 1): 'u1', ('U', 2): '<u2', ('U', 4): '<u4',
                    ('I', 1): 'i1', ('I', 2): '<i2', ('I', 4): '<i4'}
        if kind == 'binary_compressed':
            # u32 compressed size, u32 raw size, LZF data; the raw block is field-major (all x, then all y, ...)
            csize, rsize = np.frombuffer(fh.read(8), dtype='<u4')
            raw = lzf_decompress(fh.read(int(csize)), int(rsize))
            cols, off = {}, 0
            for f, t, sz, c in zip(fields, types, sizes, counts):
                if f in ('x', 'y', 'z'):
                    cols[f] = np.frombuffer(raw, dtype=np_types[(t.upper(), sz)], count=npts * c, offset=off)[::c]
                off += npts * sz * c
            return np.stack([cols['x'], cols['y'], cols['z']], axis=1).astype(np.float64)
        if kind == 'binary':
            dt = np.dtype([(f, np_types[(t.upper(), s)], (c,)) if c > 1 else (f, np_types[(t.upper(), s)])
                           for f, t, s, c in zip(fields, types, sizes, counts)])
            raw = np.frombuffer(fh.read(npts * dt.itemsize), dtype=dt, count=npts)
            return np.stack([raw['x'], raw['y'], raw['z']], axis=1).astype(np.float64)
        raise ValueError(f'{path}: PCD DATA {kind} is not supported')


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
