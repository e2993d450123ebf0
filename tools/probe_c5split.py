#!/usr/bin/env python3
"""split(' ') on the C5 column (rows of 40-150 bytes), for a kernel trace (GPU box)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
c5 = B.synth(5, rows)
for rep in range(2):
    t0 = time.perf_counter()
    cols = c5.split(" ")
    torch.cuda.synchronize()
    print("split: %d columns, %.2f ms" % (len(cols), (time.perf_counter() - t0) * 1e3), flush=True)
    del cols
