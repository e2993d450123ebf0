#!/usr/bin/env python3
"""One long row among millions of short ones (GPU box): the C3 column with a single row of `size` bytes inserted in the
middle -- does one outlier move the whole column off the tile kernels?  usage: python tools/probe_outlier.py [size]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402
from custrings_amd import nvstrings  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
rows = 50_000_000
a = B.synth(3, rows)
b = B.synth(3, rows)
long_row = nvstrings.to_device([("GET /x 10.1.2.3 " * (size // 16 + 1))[:size]])
arr = [a, long_row, b]
import ctypes as C
L = B.L
ptrs = (C.c_void_p * 3)(*[c.m_cptr for c in arr])
out = C.c_void_p()
B._lib.check(L.cs_column_concat(ptrs, 3, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
plain = B.synth(3, 2 * rows)
res8 = torch.empty(2 * rows + 1, dtype=torch.uint8, device="cuda")
for name, fn in (("split(sp, 20)", lambda c: c.split(" ", 20)), ("replace_re(IPv4)", lambda c: c.replace(B.IPV4, "<IP>")),
                 ("contains_re(IPv4)", lambda c: c.contains(B.IPV4, devptr=res8.data_ptr())), ("lower", lambda c: c.lower()),
                 ("strip", lambda c: c.strip()), ("tokenize", lambda c: __import__("custrings_amd").nvtext.tokenize(c))):
    t = []
    for c in (plain, col):
        try:
            r = fn(c); del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(c); torch.cuda.synchronize()
            t.append((time.perf_counter() - t0) * 1e3); del r
        except Exception as e:
            t.append(float("nan")); print(name, type(e).__name__, str(e)[:80])
    print("%-22s plain %8.2f ms   with one %d-byte row %9.2f ms" % (name, t[0], size, t[1]), flush=True)
print("fallbacks", int(L.cs_fallback_count()))
