#!/usr/bin/env python3
"""replace_re as the FIRST op on a freshly built column, many times (dev probe, GPU box): an intermittent 127 ms"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import tools.bench_ops as B
from custrings_amd import _lib
L = _lib.lib
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    c = B.synth(3, rows)
    if os.environ.get("FRESH_SYNC"): torch.cuda.synchronize(); time.sleep(0.05)
    f0 = int(L.cs_fallback_count())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = c.replace(B.IPV4, "<IP>")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    r2 = c.replace(B.IPV4, "<IP>")
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t1
    print("fresh column %2d: first replace %.2f ms, second %.2f ms, fallbacks %d" % (i, dt * 1e3, dt2 * 1e3, int(L.cs_fallback_count()) - f0), flush=True)
    del r, r2, c
