#!/usr/bin/env python3
"""extract on the C3 column (dev probe, GPU box): python tools/probe_extract.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
c3 = B.synth(3, rows)
for pat in (r"(\d+)\.(\d+)\.\d+\.(\d+) ", r"(GET|POST) (/\S*)", r"(\w+) (\S+) "):
    c3.extract(pat)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    c3.extract(pat)
    torch.cuda.synchronize()
    print(pat, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
