#!/usr/bin/env python3
"""replace_with_backrefs on the C3 column with 0 / 1 / 4 references (dev probe, GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
c3 = B.synth(3, rows)
pat = r"(\d+)\.(\d+)\.(\d+)\.(\d+)"
for repl in ("<IP>", r"\0", r"\1", r"\4.\3.\2.\1"):
    c3.replace_with_backrefs(pat, repl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c3.replace_with_backrefs(pat, repl)
    torch.cuda.synchronize()
    print(repr(repl), "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
t0 = time.perf_counter()
c3.replace(B.IPV4, "<IP>")
torch.cuda.synchronize()
from custrings_amd import _lib
print("replace_re", "%.2f ms" % ((time.perf_counter() - t0) * 1e3), "fallbacks so far", int(_lib.lib.cs_fallback_count()), flush=True)
