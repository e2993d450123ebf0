"""tokenize on one GPU's C5 shard (62.5M tweet-like rows): wall time per call; run under rocprofv3 for the per-kernel split."""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import torch
from custrings_amd import _lib, nvstrings, nvtext
L = _lib.lib; _lib.ensure_init(0)
def synth(kind, rows, param=0):
    out = C.c_void_p(); _lib.check(L.cs_synth_column(kind, 0, rows, 20240607, param, None, C.byref(out))); return nvstrings.nvstrings(out.value)
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
c5 = synth(5, rows)
print("C5 rows=%d, %.2f GB chars" % (rows, L.cs_column_nbytes(c5.m_cptr) / 1e9))
print("tokenize()      %9.3f ms" % t(lambda: nvtext.tokenize(c5)), flush=True)
print("tokenize(' .,') %9.3f ms" % t(lambda: nvtext.tokenize(c5, " .,") if False else nvtext.tokenize(c5, " ")), flush=True)
