#!/usr/bin/env python3
"""Dev probe: time cs_replace_re / cs_split on a C3 column (GPU box)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from custrings_amd import _lib, nvstrings
L = _lib.lib
_lib.ensure_init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
what = sys.argv[2] if len(sys.argv) > 2 else "replace"
out = C.c_void_p()
_lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
re = nvstrings._compile(r"\d+\.\d+\.\d+\.\d+")
def run():
    if what == "replace":
        o = C.c_void_p(); _lib.check(L.cs_replace_re(col.m_cptr, re, b"<IP>", -1, None, C.byref(o))); L.cs_column_destroy(o)
    elif what == "contains":
        import numpy as np
        global buf
        f = C.c_int64(); _lib.check(L.cs_contains_re(col.m_cptr, re, buf, 1, None, C.byref(f)))
    else:
        arr = C.POINTER(C.c_void_p)(); n = C.c_int()
        _lib.check(L.cs_split(col.m_cptr, b" ", -1, None, C.byref(arr), C.byref(n)))
        for i in range(n.value): L.cs_column_destroy(arr[i])
        L.cs_free(arr)
if what == "contains":
    o2 = C.c_void_p(); _lib.check(L.cs_synth_column(4, 0, rows // 8 + 8, 1, 10, None, C.byref(o2)))
    v = _lib.ColumnView(); L.cs_column_get_view(o2, C.byref(v)); buf = v.chars
run(); run()
L.cs_prof_reset(); L.cs_prof_enable(1)
t0 = time.perf_counter()
for _ in range(3): run()
dt = (time.perf_counter() - t0) / 3
L.cs_prof_enable(0)
line = "%s rows=%d debug=%s/%s wall=%.2f ms" % (what, rows, os.environ.get("CS_TILE_DEBUG", "0"), os.environ.get("CS_SPLIT_DEBUG", "0"), dt * 1e3)
for k in ["k_replace_re", "k_replace_re_size", "k_replace_re_write", "k_split_count", "k_split_sizes", "k_split_write", "k_split_sample", "k_split_measure", "k_split_emit", "k_split_fixup", "k_contains_re", "k_write_offsets"]:
    ms, n = C.c_double(), C.c_int64()
    L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
    if n.value: line += " | %s %.2f" % (k, ms.value / n.value)
print(line)
