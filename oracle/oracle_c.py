"""ctypes loader of the C oracle (oracle_c.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle_c.so")


def load():
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "oracle_c.c")):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    lib = C.CDLL(_LIB)
    lib.xh_oracle_rows.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_void_p,
                                   C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
    return lib


def bincount_rows(samples, edges, weights=None):
    """same contract as oracle_np.bincount_rows (float64 compare domain)"""
    lib = load()
    s = [np.ascontiguousarray(a, dtype=np.float64) for a in samples]
    e = [np.ascontiguousarray(b, dtype=np.float64) for b in edges]
    rows, cols = s[0].shape
    d = len(s)
    nb = tuple(len(b) - 1 for b in e)
    sp = (C.c_void_p * d)(*[a.ctypes.data for a in s])
    ep = (C.c_void_p * d)(*[b.ctypes.data for b in e])
    ne = (C.c_int64 * d)(*[len(b) for b in e])
    if weights is None:
        out = np.zeros((rows,) + nb, dtype=np.int64)
        lib.xh_oracle_rows(d, sp, ep, ne, None, rows, cols, out.ctypes.data, None)
    else:
        w = np.ascontiguousarray(weights, dtype=np.float64)
        out = np.zeros((rows,) + nb, dtype=np.float64)
        lib.xh_oracle_rows(d, sp, ep, ne, w.ctypes.data, rows, cols, None, out.ctypes.data)
    return out
