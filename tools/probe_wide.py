#!/usr/bin/env python3
"""Rows of ~520 bytes (eight C3 lines joined): the hot-path ops per call and per GB beside the 64-byte rows (GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402
from custrings_amd import nvcategory, nvtext  # noqa: E402

rows = 48_000_000
c3 = B.synth(3, rows)
n8 = rows // 8
parts = [c3.sublist(i * n8, (i + 1) * n8) for i in range(8)]
wide = parts[0].cat(parts[1:], sep=" ")
del parts
res8 = torch.empty(rows, dtype=torch.uint8, device="cuda")
OPS = [("split(' ', 8)", lambda c: c.split(" ", 8)), ("replace_re(IPv4)", lambda c: c.replace(B.IPV4, "<IP>")), ("contains_re(IPv4)", lambda c: c.contains(B.IPV4, devptr=res8.data_ptr())),
       ("count_re(IPv4)", lambda c: c.count(B.IPV4, devptr=0) if False else c.contains(r"\d+$", devptr=res8.data_ptr())),
       ("lower", lambda c: c.lower()), ("strip", lambda c: c.strip()), ("tokenize", lambda c: nvtext.tokenize(c)), ("findall(IPv4)", lambda c: c.findall(B.IPV4))]
for cname, c in (("64-byte rows", c3), ("520-byte rows", wide)):
    gb = int(B.L.cs_column_nbytes(c.m_cptr)) / 1e9
    for name, fn in OPS:
        try:
            r = fn(c); del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(c); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3; del r
            print("%-14s %-20s %8.2f ms  %6.2f ms/GB" % (cname, name, dt, dt / gb), flush=True)
        except Exception as e:
            print("%-14s %-20s %s" % (cname, name, type(e).__name__ + ": " + str(e)[:70]), flush=True)
print("fallbacks", int(B.L.cs_fallback_count()))
