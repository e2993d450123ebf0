#!/usr/bin/env python3
"""one regex op on a synthetic column, a few times (dev probe for rocprofv3 / CS_STREAM_INFO): python tools/probe_one.py kind rows op pattern"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tools.bench_ops as B
from custrings_amd import _lib, nvstrings
L = _lib.lib
kind, rows, op, pat = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
col = B.synth(kind, rows)
re = nvstrings._compile(pat)
res = torch.empty(rows, dtype=torch.uint8, device="cuda")
cnt = torch.empty(rows, dtype=torch.int32, device="cuda")
f = C.c_int64()
for i in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if op == "contains":
        _lib.check(L.cs_contains_re(col.m_cptr, re, C.c_void_p(res.data_ptr()), 1, None, C.byref(f)))
    elif op == "count":
        _lib.check(L.cs_count_re(col.m_cptr, re, C.c_void_p(cnt.data_ptr()), 1, None, C.byref(f)))
    else:
        r = col.replace(pat, sys.argv[5]); del r
    torch.cuda.synchronize()
    print("call %d: %.3f ms route %s" % (i, (time.perf_counter() - t0) * 1e3, L.cs_debug_last_route().decode()), flush=True)
