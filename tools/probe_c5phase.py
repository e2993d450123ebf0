#!/usr/bin/env python3
"""replace_re on the C5 column with the profiling build's per-phase cycle counters (stderr).  GPU box:
CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_prof.so CS_STREAM_INFO=1 python tools/probe_c5phase.py [rows] [kind]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

import tools.bench_ops as B  # noqa: E402
from custrings_amd import _lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
kind = int(sys.argv[2]) if len(sys.argv) > 2 else 5
col = B.synth(kind, rows)
for name, pat, repl in (("ipv4", r"\d+\.\d+\.\d+\.\d+", "<IP>"), ("hash", r"#\w+", "<tag>"), ("mail", r"\w+@\w+", "<m>"), ("gtest", r"(\bin\b)|(\ba\b)|(\bthe\b)", "=")):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = col.replace(pat, repl)
        torch.cuda.synchronize()
        sys.stderr.write("== %s replace_re rep %d: %.2f ms route %s\n" % (name, rep, (time.perf_counter() - t0) * 1e3, _lib.lib.cs_debug_last_route().decode()))
        sys.stderr.flush()
        del r
