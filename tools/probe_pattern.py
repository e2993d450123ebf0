#!/usr/bin/env python3
"""contains_re / replace_re of one pattern on the C3 column (rows, pattern, repl from argv): which kernels run, how often the
host fell back.  Meant to run under rocprofv3 --kernel-trace --stats."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from custrings_amd import _lib, nvstrings  # noqa: E402

L = _lib.lib
_lib.ensure_init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
pat = sys.argv[2] if len(sys.argv) > 2 else r"(\bin\b)|(\ba\b)|(\bthe\b)"
repl = sys.argv[3] if len(sys.argv) > 3 else "="
out = C.c_void_p()
_lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
res = torch.empty(rows, dtype=torch.uint8, device="cuda")
f0 = int(L.cs_fallback_count())
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    col.contains(pat, devptr=res.data_ptr())
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    r = col.replace(pat, repl)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("contains %.2f ms, replace %.2f ms, out bytes %d, hits %d, fallbacks so far %d" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, int(L.cs_column_nbytes(r.m_cptr)),
                                                                                              int(res.sum().item()), int(L.cs_fallback_count()) - f0), flush=True)
    del r
