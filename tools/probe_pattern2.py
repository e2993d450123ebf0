#!/usr/bin/env python3
"""Dev probe: replace_re of one pattern on a C3 column: python tools/probe_pattern2.py ROWS PATTERN [REPL]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from custrings_amd import _lib, nvstrings
L = _lib.lib
_lib.ensure_init(0)
rows = int(sys.argv[1]); pat = sys.argv[2]; repl = (sys.argv[3] if len(sys.argv) > 3 else "=").encode()
out = C.c_void_p()
_lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
re = nvstrings._compile(pat)
def run():
    o = C.c_void_p(); _lib.check(L.cs_replace_re(col.m_cptr, re, repl, -1, None, C.byref(o))); L.cs_column_destroy(o)
for i in range(3):
    t0 = time.perf_counter(); run(); print("call %d: %.2f ms, fallbacks %d" % (i, (time.perf_counter() - t0) * 1e3, L.cs_fallback_count()), flush=True)
