#!/usr/bin/env python3
"""k_split_emit's time launch by launch (dev probe, GPU box): does it drift (clocks) or jump (where the buffers land)?"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from custrings_amd import _lib, nvstrings
L = _lib.lib
_lib.ensure_init(0)
rows = 100_000_000
out = C.c_void_p()
_lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
def split(keep):
    arr = C.POINTER(C.c_void_p)(); n = C.c_int()
    L.cs_prof_reset(); L.cs_prof_enable(1)
    _lib.check(L.cs_split(col.m_cptr, b" ", -1, None, C.byref(arr), C.byref(n)))
    L.cs_prof_enable(0)
    ms, k = C.c_double(), C.c_int64()
    L.cs_prof_get(b"k_split_emit", C.byref(ms), C.byref(k))
    m2 = C.c_double(); L.cs_prof_get(b"k_split_measure", C.byref(m2), C.byref(k))
    v = _lib.ColumnView(); L.cs_column_get_view(arr[0], C.byref(v))
    a0 = int(v.chars or 0)
    v1 = _lib.ColumnView(); L.cs_column_get_view(arr[1], C.byref(v1))
    a1 = int(v1.chars or 0)
    cols = [arr[i] for i in range(n.value)]
    return ms.value, m2.value, a0, a1, cols, arr
held = None
for i in range(24):
    keep = (i // 6) % 2 == 1   # rounds of six: outputs freed at once / held over one more call
    ms, m2, a0, a1, cols, arr = split(keep)
    print("call %2d emit %.3f ms measure %.3f  col0 chars @%x (mod 2M %6x)  col1-col0 %d  %s" % (i, ms, m2, a0, a0 & 0x1FFFFF, a1 - a0, "held" if keep else ""), flush=True)
    if held is not None:
        for c in held[0]: L.cs_column_destroy(c)
        L.cs_free(held[1]); held = None
    if keep: held = (cols, arr)
    else:
        for c in cols: L.cs_column_destroy(c)
        L.cs_free(arr)
