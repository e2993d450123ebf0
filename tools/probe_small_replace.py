import ctypes as C, os, sys, time
sys.path.insert(0, os.getcwd())
from custrings_amd import _lib, nvstrings
L = _lib.lib
_lib.ensure_init(0)
rows = int(sys.argv[1])
out = C.c_void_p()
_lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
re = nvstrings._compile(r"(\d+)\.(\d+)")
for repl in (b"<IP>", b"<a-much-longer-one>"):
    def run():
        o = C.c_void_p(); _lib.check(L.cs_replace_re(col.m_cptr, re, repl, -1, None, C.byref(o))); L.cs_column_destroy(o)
    run(); run()
    L.cs_prof_reset(); L.cs_prof_enable(1)
    for _ in range(5): run()
    L.cs_prof_enable(0)
    ms, n = C.c_double(), C.c_int64()
    L.cs_prof_get(b"k_replace_re", C.byref(ms), C.byref(n))
    print("rows", rows, repl, "k_replace_re launches", n.value, "avg ms %.4f" % (ms.value / max(1, n.value)), "fallbacks", L.cs_fallback_count())
