#!/usr/bin/env python3
"""Category build of the C4 shard (125M rows) over 2^40 names (about 58M distinct keys): run under rocprofv3 --kernel-trace --stats."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

from custrings_amd import _lib, nvcategory, nvstrings  # noqa: E402

L = _lib.lib
_lib.ensure_init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 40
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 125_000_000
out = C.c_void_p()
_lib.check(L.cs_synth_column(4, 0, rows, 20240607, K, None, C.byref(out)))
col = nvstrings.nvstrings(out.value)
for _ in range(2):
    cat = nvcategory.from_strings(col)
    print(cat.keys_size())
    del cat
