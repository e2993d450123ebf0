#!/usr/bin/env python3
"""One op of the ops table, three calls (for rocprofv3 --kernel-trace --stats): python tools/probe_op.py <op> [scale]
ops: extract backrefs findall tokenize strip cat1m cat1k ipv4b dense"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402
from custrings_amd import nvcategory, nvtext  # noqa: E402

op = sys.argv[1]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
if op in ("extract", "backrefs", "findall", "ipv4b", "dense", "densec"):
    c = B.synth(3, int(100_000_000 * scale))
    resb = torch.empty(c.size(), dtype=torch.uint8, device="cuda")
    fn = {"extract": lambda: c.extract(r"(\d+)\.(\d+)\.\d+\.(\d+) "),
          "backrefs": lambda: c.replace_with_backrefs(r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"),
          "findall": lambda: c.findall(B.IPV4),
          "ipv4b": lambda: c.replace(B.IPV4B, "<IP>"),
          "dense": lambda: c.replace(B.GTEST, "="),
          "densec": lambda: c.contains(B.GTEST, devptr=resb.data_ptr())}[op]
elif op == "tokenize":
    c = B.synth(5, int(62_500_000 * scale))
    fn = lambda: nvtext.tokenize(c)
elif op == "strip":
    c = B.synth(2, int(10_000_000 * scale)).lower()
    fn = lambda: c.strip()
elif op in ("cat1m", "cat1k"):
    c = B.synth(4, int(125_000_000 * scale), 1 << 20 if op == "cat1m" else 1000)
    fn = lambda: nvcategory.from_strings(c)
else:
    raise SystemExit("unknown op " + op)
import time
from custrings_amd import _lib
r = fn()
del r
torch.cuda.synchronize()
f0 = int(_lib.lib.cs_fallback_count())
t0 = time.perf_counter()
for _ in range(3):
    r = fn()
    del r
torch.cuda.synchronize()
print("%s: %.3f ms per call (fallbacks %d)" % (op, (time.perf_counter() - t0) / 3 * 1e3, int(_lib.lib.cs_fallback_count()) - f0), flush=True)
