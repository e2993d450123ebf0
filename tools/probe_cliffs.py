#!/usr/bin/env python3
"""Ops on columns / patterns outside the fast paths' envelope (rows beyond the 96-byte masks, patterns on the list simulator,
patterns that match the empty string): wall time per call, device synchronised.  One JSON line each."""
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


def synth(kind, rows):
    out = C.c_void_p()
    _lib.check(L.cs_synth_column(kind, 0, rows, 20240607, 0, None, C.byref(out)))
    return nvstrings.nvstrings(out.value)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
        del r
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def line(what, rows, nbytes, dt):
    print(json.dumps({"case": what, "rows": rows, "ms": round(dt * 1e3, 2), "input_GBps": round(nbytes / dt / 1e9, 1)}), flush=True)


c5 = synth(5, 62_500_000)
b5 = int(L.cs_column_nbytes(c5.m_cptr))
f0 = int(L.cs_fallback_count())
line("C5 (rows of 40-150 bytes) split(' ')", 62_500_000, b5, timed(lambda: c5.split(" ")))
line("C5 split(' ', 3)", 62_500_000, b5, timed(lambda: c5.split(" ", 3)))
line("C5 replace_re([aeiou]+ -> '*')", 62_500_000, b5, timed(lambda: c5.replace("[aeiou]+", "*")))
line("C5 replace_re(\\\\bthe\\\\b -> 'THE')", 62_500_000, b5, timed(lambda: c5.replace(r"\bthe\b", "THE")))
res = torch.empty(62_500_000, dtype=torch.uint8, device="cuda")
line("C5 contains_re(ing\\\\b)", 62_500_000, b5, timed(lambda: c5.contains(r"ing\b", devptr=res.data_ptr())))
line("C5 lower", 62_500_000, b5, timed(lambda: c5.lower()))
line("C5 strip", 62_500_000, b5, timed(lambda: c5.strip()))
del c5
c3 = synth(3, 100_000_000)
b3 = int(L.cs_column_nbytes(c3.m_cptr))
res = torch.empty(100_000_000, dtype=torch.uint8, device="cuda")
P = r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}"
line("C3 contains_re(IPv4 {1,3} without \\\\b: eight-slot DFA)", 100_000_000, b3, timed(lambda: c3.contains(P, devptr=res.data_ptr()), reps=1))
line("C3 replace_re(IPv4 {1,3} without \\\\b: eight-slot DFA)", 100_000_000, b3, timed(lambda: c3.replace(P, "<IP>"), reps=1))
line("C3 replace_re(x* -> '-')", 100_000_000, b3, timed(lambda: c3.replace("x*", "-"), reps=1))
line("C3 split('/')", 100_000_000, b3, timed(lambda: c3.split("/")))
line("C3 split(' /')  (two-byte delimiter)", 100_000_000, b3, timed(lambda: c3.split(" /")))
print(json.dumps({"fallbacks": int(L.cs_fallback_count()) - f0}))
