#!/usr/bin/env python3
"""category build with the library's kernel timers (dev probe, GPU box): python tools/prof_cat.py cat1m|cat1k"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
import tools.bench_ops as B
from custrings_amd import _lib, nvcategory
L = _lib.lib
op = sys.argv[1]
c = B.synth(4, 125_000_000, 1 << 20 if op == "cat1m" else 1000)
for _ in range(2):
    r = nvcategory.from_strings(c); del r
L.cs_prof_reset(); L.cs_prof_enable(1)
for _ in range(3):
    r = nvcategory.from_strings(c); del r
L.cs_prof_enable(0)
line = op
for k in ["k_cat_insert", "k_cat_values", "k_cat_sort", "k_cat_keys"]:
    ms, n = C.c_double(), C.c_int64()
    L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
    if n.value: line += " | %s %.3f" % (k, ms.value / n.value)
print(line)
