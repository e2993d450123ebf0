#!/usr/bin/env python3
"""The hot-path ops on the C3 column with every third row null (GPU box): per call, beside the column without nulls."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402
from custrings_amd import nvcategory, nvstrings, nvtext  # noqa: E402

rows = 50_000_000
plain = B.synth(3, rows)
pos = torch.arange(0, rows, 3, dtype=torch.int32, device="cuda")
out = C.c_void_p()
B._lib.check(B.L.cs_scatter_scalar(plain.m_cptr, None, pos.data_ptr(), pos.numel(), 1, None, C.byref(out)))
nulls = nvstrings.nvstrings(out.value)
res8 = torch.empty(rows, dtype=torch.uint8, device="cuda")
OPS = [("split(' ')", lambda c: c.split(" ")), ("replace_re(IPv4)", lambda c: c.replace(B.IPV4, "<IP>")), ("contains_re(IPv4)", lambda c: c.contains(B.IPV4, devptr=res8.data_ptr())),
       ("lower", lambda c: c.lower()), ("strip", lambda c: c.strip()), ("tokenize", lambda c: nvtext.tokenize(c)), ("findall(IPv4)", lambda c: c.findall(B.IPV4)),
       ("category", lambda c: nvcategory.from_strings(c))]
for name, fn in OPS:
    t = []
    for c in (plain, nulls):
        r = fn(c); del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(c); torch.cuda.synchronize()
        t.append((time.perf_counter() - t0) * 1e3); del r
    print("%-20s no nulls %8.2f ms   every third row null %8.2f ms" % (name, t[0], t[1]), flush=True)
print("fallbacks", int(B.L.cs_fallback_count()))
