#!/usr/bin/env python3
"""Column shapes other than the benchmark's (GPU box): the same ops on (a) the C3 column, (b) the same with a two-byte
character in every row, (c) with every second row null, (d) rows of 1-6 bytes -- milliseconds per call and per GB of
input, to spot a shape that falls off the fast paths."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402
from custrings_amd import nvtext  # noqa: E402

rows = 50_000_000
plain = B.synth(3, rows)
accent = plain.replace("/", "é", regex=False)  # (every log line holds a path: two-byte characters in every row)
idx = torch.arange(0, rows, 2, dtype=torch.int32, device="cuda")
nulls = plain.scatter(B.nvstrings.to_device([None]), idx[:1]) if False else None
short = plain.split(" ")[8]  # a column of short tokens
cols = [("C3", plain), ("two-byte char per row", accent), ("short rows", short)] if not os.environ.get("SHAPES_ONLY_ACCENT") else [("two-byte char per row", accent)]
res8 = torch.empty(rows, dtype=torch.uint8, device="cuda")
OPS = [("replace_re(IPv4)", lambda c: c.replace(B.IPV4, "<IP>")), ("contains_re(IPv4)", lambda c: c.contains(B.IPV4, devptr=res8.data_ptr())),
       ("replace_re(\\d+)", lambda c: c.replace(r"\d+", "#")), ("split(' ', 4)", lambda c: c.split(" ", 4)), ("lower", lambda c: c.lower()), ("strip", lambda c: c.strip()),
       ("tokenize", lambda c: nvtext.tokenize(c)), ("category", lambda c: __import__("custrings_amd").nvcategory.from_strings(c))]
for cname, c in cols:
    gb = int(B.L.cs_column_nbytes(c.m_cptr)) / 1e9
    for name, fn in OPS:
        try:
            r = fn(c); del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(c); torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e3; del r
            print("%-24s %-20s %8.2f ms  (%.2f GB: %6.1f ms/GB)" % (cname, name, dt, gb, dt / gb), flush=True)
        except Exception as e:
            print("%-24s %-20s %s" % (cname, name, type(e).__name__ + ": " + str(e)[:70]), flush=True)
print("fallbacks", int(B.L.cs_fallback_count()))
