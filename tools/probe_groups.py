#!/usr/bin/env python3
"""findall / extract on the 100M-row C3 column, one op per invocation, for a rocprofv3 --kernel-trace --stats run
(per-kernel breakdown of the ops that build several output columns).  Usage: python tools/probe_groups.py findall|extract [rows]
"""
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


def main():
    op = sys.argv[1]
    rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    out = C.c_void_p()
    _lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out)))
    c3 = nvstrings.nvstrings(out.value)
    if op == "findall":
        fn = lambda: c3.findall(r"\d+\.\d+\.\d+\.\d+")
    elif op == "extract":
        fn = lambda: c3.extract(r"(\d+)\.(\d+)\.\d+\.(\d+) ")
    else:
        raise SystemExit("findall|extract")
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        r = fn()
        del r
    torch.cuda.synchronize()
    print(op, rows, "%.3f ms" % ((time.perf_counter() - t0) / 3 * 1e3))


if __name__ == "__main__":
    main()
