"""ctypes wrapper of oracle/libseekr_oracle.so (C restatement; test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libseekr_oracle.so")
_h = None


def lib():
    global _h
    if _h is None:
        if not os.path.exists(_LIB):
            subprocess.run(["make", "-s", "-C", _HERE], check=True)
        _h = C.CDLL(_LIB)
        _h.orc_count_u32.restype = C.c_int
        _h.orc_per_kb_f32.restype = C.c_int
        _h.orc_colsum_seq_f32.restype = None
    return _h


def count_u32(blob, offsets, k, alphabet="AGTC"):
    """blob: uint8 array of concatenated ASCII; offsets: int64 [n+1] -> uint32 [n, A^k]."""
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    out = np.empty((n, len(alphabet) ** k), dtype=np.uint32)
    alpha = alphabet.encode("latin-1")
    lib().orc_count_u32(blob.ctypes.data_as(C.c_void_p), offsets.ctypes.data_as(C.c_void_p), C.c_int64(n), C.c_int(k),
                        alpha, C.c_int(len(alpha)), out.ctypes.data_as(C.c_void_p))
    return out


def per_kb_f32(counts, lengths, k):
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    out = np.empty(counts.shape, dtype=np.float32)
    rc = lib().orc_per_kb_f32(counts.ctypes.data_as(C.c_void_p), lengths.ctypes.data_as(C.c_void_p),
                              C.c_int64(counts.shape[0]), C.c_int(k), C.c_int64(counts.shape[1]),
                              out.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise ZeroDivisionError("division by zero")
    return out


def colsum_seq_f32(x, acc=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    acc = np.zeros(x.shape[1], dtype=np.float32) if acc is None else acc
    lib().orc_colsum_seq_f32(x.ctypes.data_as(C.c_void_p), C.c_int64(x.shape[0]), C.c_int64(x.shape[1]),
                             acc.ctypes.data_as(C.c_void_p))
    return acc


def seqs_to_blob(seqs):
    lengths = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    offsets = np.zeros(len(seqs) + 1, dtype=np.int64)
    np.cumsum(lengths, out=offsets[1:])
    blob = np.frombuffer("".join(seqs).encode("latin-1", "replace"), dtype=np.uint8)
    return blob, offsets
