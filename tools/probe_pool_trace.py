#!/usr/bin/env python3
"""The buffer pool's requests and releases over the round-6 pool test's steps (stderr: `pool+ bytes capacity hit|miss`,
`pool- bytes capacity`), for replaying allocation policies offline.  GPU box."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gpuutil  # noqa: E402
from custrings_amd import nvstrings  # noqa: E402

L = gpuutil.lib()
re = gpuutil.compile_re(r"\d+\.\d+\.\d+\.\d+")


def step(rows, seed):
    g = gpuutil.synth(3, 0, rows, seed=seed)
    cols = g.split(" ")
    o = C.c_void_p()
    L.check(L.lib.cs_replace_re(g.m_cptr, re, b"<IP>", -1, None, C.byref(o)))
    out = nvstrings.nvstrings(o.value)
    del cols, out, g


base = 3_000_000
L.check(L.lib.cs_config_set(b"CS_POOL_TRACE", b"2"))
for i, f in enumerate((1.01, 0.99, 1.0, 0.995, 1.008, 1.01, 1.003, 0.992)):
    sys.stderr.write("step %d rows %d\n" % (i, int(base * f)))
    sys.stderr.flush()
    step(int(base * f), 11 + i)
