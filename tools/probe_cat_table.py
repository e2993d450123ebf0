#!/usr/bin/env python3
"""category build at K = 1M / 1k against the first table's size (dev probe, GPU box): python tools/probe_cat_table.py"""
import ctypes as C, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    import tools.bench_ops as B
    from custrings_amd import _lib, nvcategory
    L = _lib.lib
    k = int(sys.argv[1])
    c = B.synth(4, 125_000_000, k)
    for _ in range(2):
        r = nvcategory.from_strings(c); del r
    torch.cuda.synchronize()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); r = nvcategory.from_strings(c); torch.cuda.synchronize(); t.append((time.perf_counter() - t0) * 1e3); del r
    L.cs_prof_reset(); L.cs_prof_enable(1)
    for _ in range(3):
        r = nvcategory.from_strings(c); del r
    L.cs_prof_enable(0)
    line = "K=%d log2=%s grid_x=%s | build %.3f ms (min of 5)" % (k, os.environ.get("CS_CAT_FIRST_LOG2", "22"), os.environ.get("CS_CAT_GRID_X", "4"), min(t))
    for kn in ["k_cat_insert", "k_cat_values", "k_cat_sort"]:
        ms, n = C.c_double(), C.c_int64()
        L.cs_prof_get(kn.encode(), C.byref(ms), C.byref(n))
        if n.value: line += " | %s %.3f" % (kn, ms.value / n.value)
    print(line, flush=True)
else:
    for k in (1 << 20, 1000):
        for gx in ("1000", "1", "2", "4", "8", "32"):
            env = dict(os.environ, CS_CAT_GRID_X=gx)
            subprocess.run([sys.executable, __file__, str(k)], env=env)
