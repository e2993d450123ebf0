#!/usr/bin/env python3
"""The bit-parallel regex route (regex_bits.h) against the automaton routes, pattern by pattern, on the C3 column:
wall time per call (device synchronised) of contains_re / count_re / replace_re with the route forced on and off.
Usage: python tools/probe_bits.py [rows]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from custrings_amd import _lib, nvstrings  # noqa: E402

L = _lib.lib
_lib.ensure_init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 3
out = C.c_void_p()
_lib.check(L.cs_synth_column(kind, 0, rows, 20240607, 0, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
PATTERNS = [r"(\bin\b)|(\ba\b)|(\bthe\b)", r"[aeiou]+", r"\bthe\b", r"cat|cot|cut", r"POST|PUT", r"[^ ]+", r"#\w+", r"x[0-9][0-9]", r"ing$", r"^GET|^PUT", r"e", r"[0-9]+"]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
        del r
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def setsw(name, value):
    L.cs_config_set(name.encode(), None if value is None else str(value).encode())


for pat in PATTERNS:
    re = nvstrings._compile(pat)
    res = torch.zeros(rows, dtype=torch.uint8, device="cuda")
    cnt = torch.zeros(rows, dtype=torch.int32, device="cuda")
    found = C.c_int64()
    line = {"pattern": pat, "rows": rows}
    for mode, sw in (("bits", ("CS_BITS_ALWAYS", "CS_NO_BITS_FORM")), ("auto", (None, None)), ("off", ("CS_NO_BITS_FORM", "CS_BITS_ALWAYS"))):
        if sw[0]:
            setsw(sw[0], 1)
        if sw[1]:
            setsw(sw[1], None)
        if mode == "auto":
            setsw("CS_BITS_ALWAYS", None)
            setsw("CS_NO_BITS_FORM", None)
        line[mode] = {
            "contains_ms": round(timed(lambda: _lib.check(L.cs_contains_re(col.m_cptr, re, res.data_ptr(), 1, None, C.byref(found)))), 3),
            "route_c": L.cs_debug_last_route().decode(),
            "count_ms": round(timed(lambda: _lib.check(L.cs_count_re(col.m_cptr, re, cnt.data_ptr(), 1, None, C.byref(found)))), 3),
            "replace_ms": round(timed(lambda: col.replace(pat, "=")), 3),
            "route_r": L.cs_debug_last_route().decode(),
            "fallbacks": int(L.cs_fallback_count()),
        }
    setsw("CS_BITS_ALWAYS", None)
    setsw("CS_NO_BITS_FORM", None)
    L.cs_regex_destroy(re)
    print(json.dumps(line), flush=True)
