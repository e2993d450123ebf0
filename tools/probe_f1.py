#!/usr/bin/env python3
"""findall / extract / replace_with_backrefs / the gtest alternation on the C3 column, for a kernel trace (GPU box):
rocprofv3 --kernel-trace --stats -- python tools/probe_f1.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
c3 = B.synth(3, rows)
GT = r"(\bin\b)|(\ba\b)|(\bthe\b)"
OPS = [("findall", lambda: c3.findall(B.IPV4)),
       ("extract", lambda: c3.extract(r"(\d+)\.(\d+)\.\d+\.(\d+) ")),
       ("backrefs", lambda: c3.replace_with_backrefs(r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1")),
       ("contains_gtest", lambda: c3.contains(GT)),
       ("replace_gtest", lambda: c3.replace(GT, "="))]
only = os.environ.get("F1_ONLY")
for name, fn in OPS:
    if only and name not in only.split(","):
        continue
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    print(name, "%.2f ms" % ((time.perf_counter() - t0) * 1e3 / 3), flush=True)
