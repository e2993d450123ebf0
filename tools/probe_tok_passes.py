#!/usr/bin/env python3
"""tokenize on one GPU's C5 shard (62.5M rows): the two passes' device times (HIP events on the launch stream) and the
call's wall time -- what a single pass could save at most (VERDICT r4 missing #6)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from custrings_amd import _lib, nvstrings, nvtext
L = _lib.lib; _lib.ensure_init(0)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
out = C.c_void_p(); _lib.check(L.cs_synth_column(5, 0, rows, 20240607, 0, None, C.byref(out))); c5 = nvstrings.nvstrings(out.value)
for _ in range(2):
    r = nvtext.tokenize(c5); del r
L.cs_prof_reset(); L.cs_prof_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    r = nvtext.tokenize(c5); n = r.size(); nb = L.cs_column_nbytes(r.m_cptr); del r
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
L.cs_prof_enable(0)
line = "tokenize C5 rows=%d in %.2f GB -> %d tokens, %.2f GB: wall %.3f ms" % (rows, L.cs_column_nbytes(c5.m_cptr) / 1e9, n, nb / 1e9, wall)
for k in ["k_tok_count", "k_tok_write", "k_write_offsets"]:
    ms, cnt = C.c_double(), C.c_int64()
    L.cs_prof_get(k.encode(), C.byref(ms), C.byref(cnt))
    if cnt.value: line += " | %s %.3f ms" % (k, ms.value / cnt.value)
print(line)
