"""Minimal MAT-file (Level 5) reader for the reference's dataset files.

``Data_Generation.py:218-219`` stores each graph with ``scipy.io.savemat``: a sparse ``adj`` (MAT keeps sparse
matrices column-compressed: ``jc`` / ``ir`` / ``pr``), dense ``weights``, and a few scalars.  ``scipy.io.loadmat``
spends ~0.8 ms per file, mostly in generic object construction; 500 files are then 1000x the GPU time of solving
them.  This reader walks the element tags itself and hands out NumPy views: numeric arrays as they are stored
(column-major -> returned transposed to MATLAB's shape), sparse matrices as ``SparseCSC(shape, jc, ir, data)``
without building a SciPy object.  Supported: little-endian v5 files, numeric / sparse / char arrays,
``miCOMPRESSED`` elements; cells, structs and objects raise ``NotImplementedError`` (the datasets hold none).
"""
from __future__ import annotations

import struct
import zlib
from collections import namedtuple

import numpy as np

SparseCSC = namedtuple("SparseCSC", "shape jc ir data")

_MI = {1: np.int8, 2: np.uint8, 3: np.int16, 4: np.uint16, 5: np.int32, 6: np.uint32, 7: np.float32, 9: np.float64,
       12: np.int64, 13: np.uint64}
_MI_MATRIX, _MI_COMPRESSED, _MI_UTF8 = 14, 15, 16
_MX_CELL, _MX_STRUCT, _MX_OBJECT, _MX_CHAR, _MX_SPARSE = 1, 2, 3, 4, 5


def _tag(buf, pos):
    """-> (type, nbytes, data offset, offset of the next element)."""
    word = struct.unpack_from("<I", buf, pos)[0]
    if word >> 16:  # small data element: type in the low half, byte count in the high half, data in the tag
        return word & 0xffff, word >> 16, pos + 4, pos + 8
    nbytes = struct.unpack_from("<I", buf, pos + 4)[0]
    end = pos + 8 + nbytes
    return word, nbytes, pos + 8, (end + 7) & ~7


def _numeric(buf, pos):
    t, n, off, nxt = _tag(buf, pos)
    if t == _MI_UTF8:
        return np.frombuffer(buf, dtype=np.uint8, count=n, offset=off), nxt
    if t not in _MI:
        raise NotImplementedError("MAT data type %d" % t)
    dt = np.dtype(_MI[t]).newbyteorder("<")
    return np.frombuffer(buf, dtype=dt, count=n // dt.itemsize, offset=off), nxt


def _matrix(buf, pos, end):
    flags, pos = _numeric(buf, pos)
    klass = int(flags[0]) & 0xff
    if int(flags[0]) & 0x0800:
        raise NotImplementedError("complex arrays")
    dims, pos = _numeric(buf, pos)
    name, pos = _numeric(buf, pos)
    name = bytes(name).decode("latin1")
    dims = tuple(int(d) for d in dims)
    if klass in (_MX_CELL, _MX_STRUCT, _MX_OBJECT):
        raise NotImplementedError("cell / struct / object arrays (variable %r)" % name)
    if klass == _MX_SPARSE:
        ir, pos = _numeric(buf, pos)
        jc, pos = _numeric(buf, pos)
        data = None
        if pos < end:
            data, pos = _numeric(buf, pos)
        nnz = int(jc[-1]) if jc.size else 0
        return name, SparseCSC(dims, jc, ir[:nnz], None if data is None else data[:nnz])
    real, pos = _numeric(buf, pos)
    arr = real.reshape(dims[::-1]).T if real.size == int(np.prod(dims)) else real
    if klass == _MX_CHAR:
        arr = "".join(chr(int(c)) for c in np.asarray(arr).ravel(order="F"))
    return name, arr


def loadmat(path):
    """{variable name: ndarray | SparseCSC | str} of a MAT v5 file."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < 128 or buf[126:128] != b"IM":
        raise ValueError("%s: not a little-endian MAT v5 file" % path)
    out, pos = {}, 128
    while pos + 8 <= len(buf):
        t, n, off, nxt = _tag(buf, pos)
        if t == _MI_COMPRESSED:
            inner = zlib.decompress(buf[off:off + n])
            it, inb, ioff, _ = _tag(inner, 0)
            if it != _MI_MATRIX:
                raise NotImplementedError("compressed element of type %d" % it)
            name, val = _matrix(inner, ioff, ioff + inb)
            nxt = off + n  # compressed elements are not padded
        elif t == _MI_MATRIX:
            name, val = _matrix(buf, off, off + n)
        else:
            raise NotImplementedError("top-level element of type %d" % t)
        out[name] = val
        pos = nxt
    return out


def symmetric_csr(sp_csc: SparseCSC, strict: bool = False):
    """(indptr, indices) of a SYMMETRIC 0/1 matrix stored column-compressed: for a symmetric matrix the CSC arrays
    are its CSR arrays.  Always checks squareness, sorted rows and in-degree == out-degree; ``strict`` compares
    the full transposed pattern."""
    (n, m), jc, ir = sp_csc.shape, sp_csc.jc, sp_csc.ir
    if n != m:
        raise ValueError("adjacency must be square, got %s" % ((n, m),))
    indptr = np.asarray(jc, dtype=np.int64)
    indices = np.asarray(ir, dtype=np.int64)
    if indptr.size != n + 1:
        raise ValueError("malformed sparse matrix")
    if indices.size:
        if not np.array_equal(np.bincount(indices, minlength=n), np.diff(indptr)):
            raise ValueError("adjacency is not symmetric")
        inner = np.ones(indices.size, dtype=bool)
        inner[indptr[1:-1][indptr[1:-1] < indices.size]] = False  # positions where a new column starts
        if np.any((np.diff(indices) <= 0) & inner[1:]):
            raise ValueError("row indices are not strictly increasing within a column")
        if strict:
            cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(indptr))
            if not np.array_equal(np.sort(indices * n + cols), cols * n + indices):
                raise ValueError("adjacency is not symmetric")
    return indptr, indices
