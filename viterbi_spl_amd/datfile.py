"""Reader/writer for the reference's ``.dat`` array files (HMM parameter fixtures).

Format (self_defined/save_np_array_to_file.py:4-39, load_np_array_from_file.py:3-27 in the
reference repo): one ASCII header line ``<name> [C|F] <dtype> <dim0> <dim1> ...\\n`` followed by
the raw little-endian bytes in C order.  Older files (e.g. msnet/viterbi_*.dat) omit the C/F flag.
An ``F`` flag means: stored C-order, but handed back Fortran-contiguous.
"""
from __future__ import annotations

import os

import numpy as np


def load_np_array_from_file_fn(file_name):
    """Returns (record_name, array)."""
    with open(file_name, 'rb') as fh:
        header = fh.readline().decode('utf-8').split()
        payload = fh.read()
    rec_name, rest = header[0], header[1:]
    order = 'C'
    if rest[0] in ('C', 'F'):
        order, rest = rest[0], rest[1:]
    dtype, dims = np.dtype(rest[0]), [int(d) for d in rest[1:]]
    arr = np.frombuffer(payload, dtype=dtype).reshape(*dims)
    if order == 'F' and len(dims) > 1:
        arr = np.require(arr, requirements=['F'])
    return rec_name, arr


def save_np_array_to_file_fn(file_name, output, rec_name):
    assert isinstance(rec_name, str) and len(rec_name) and ' ' not in rec_name
    assert isinstance(output, np.ndarray) and output.ndim >= 1
    c_flag, f_flag = output.flags['C_CONTIGUOUS'], output.flags['F_CONTIGUOUS']
    if output.ndim == 1:
        order = 'C'
    else:
        assert c_flag != f_flag, "array must be exactly one of C- or F-contiguous"
        order = 'C' if c_flag else 'F'
    body = np.ascontiguousarray(output)
    header = ' '.join([rec_name, order, str(output.dtype)] + [str(d) for d in output.shape]) + '\n'
    with open(file_name, 'wb') as fh:
        fh.write(header.encode('utf-8'))
        fh.write(body.tobytes())
        fh.flush()
        os.fsync(fh.fileno())
