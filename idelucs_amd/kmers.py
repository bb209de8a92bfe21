"""idelucs_amd.kmers -- drop-in for the reference's Cython module `idelucs.kmers`.

Same two functions, same argument conventions, same in-place accumulate semantics
(reference idelucs/kmers.pyx:2 `kmer_counts`, :53 `cgr`); the counting runs on the MI355X through
libidelucs_hip.so (idl_kmer_counts / idl_cgr).  No CPU implementation exists here.
"""
import ctypes

import numpy as np

from . import _lib

_DTYPE_NAMES = {"b": "signed char", "B": "unsigned char", "h": "short", "H": "unsigned short", "i": "int",
                "I": "unsigned int", "l": "long", "L": "unsigned long", "q": "long long", "Q": "unsigned long long",
                "f": "float", "d": "double", "?": "bool"}


def _seq_view(seq):
    """`unsigned char[::1] seq` (kmers.pyx:2): writable, C-contiguous, 1-D, itemsize 1, format 'B'."""
    try:
        mv = memoryview(seq)
    except TypeError:
        raise TypeError(f"a bytes-like object is required, not '{type(seq).__name__}'")
    if mv.readonly:
        raise BufferError("Object is not writable.")
    if mv.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % mv.ndim)
    if mv.format not in ("B",):
        raise ValueError("Buffer dtype mismatch, expected 'unsigned char' but got '%s'" % _DTYPE_NAMES.get(mv.format, mv.format))
    if not mv.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return mv


def _counts_view(counts):
    """`int[::1] counts` (kmers.pyx:2): writable, C-contiguous, 1-D int32."""
    mv = memoryview(counts)
    if mv.readonly:
        raise BufferError("Object is not writable.")
    if mv.ndim != 1:
        raise ValueError("Buffer has wrong number of dimensions (expected 1, got %d)" % mv.ndim)
    if mv.format != "i":
        raise ValueError("Buffer dtype mismatch, expected 'int' but got '%s'" % _DTYPE_NAMES.get(mv.format, mv.format))
    if not mv.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return mv


def _run(fn, seq, k, counts):
    sv, cv = _seq_view(seq), _counts_view(counts)
    k = int(k)
    if not 1 <= k <= _lib.MAX_K:
        raise ValueError(f"k={k} is outside 1..{_lib.MAX_K} (one wavefront's LDS histogram holds 4^k uint32 bins)")
    if cv.shape[0] < 4 ** k:
        # the reference has bounds checks off (kmers.pyx:1) and would corrupt memory here
        raise ValueError(f"counts has {cv.shape[0]} entries; k={k} needs {4 ** k}")
    _lib.require_gpu()
    s = np.frombuffer(sv, dtype=np.uint8)
    c = np.frombuffer(cv, dtype=np.int32)
    _lib.check(fn(s.ctypes.data if s.size else None, s.size, k, c.ctypes.data))


def kmer_counts(seq, k, counts):
    """Accumulate the k-mer counts of `seq` (ASCII; only A,C,G,T count) into `counts` (int32[4^k])."""
    _run(_lib.lib.idl_kmer_counts, seq, k, counts)


def cgr(seq, k, CGR):
    """Accumulate the 2^k x 2^k chaos-game-representation counts of `seq` into `CGR` (int32[4^k])."""
    _run(_lib.lib.idl_cgr, seq, k, CGR)
