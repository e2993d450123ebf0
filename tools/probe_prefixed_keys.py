import os, sys, time
sys.path.insert(0, ".")
import torch
import tools.bench_ops as B
from custrings_amd import nvcategory
rows = 20_000_000
for K in (1000, 1 << 20, 1 << 40):
    c4 = B.synth(4, rows, K)
    pre = c4.cat(None) if False else None
    # keys behind a shared 20-byte prefix: every key ties on the sort's 8-byte prefix
    import ctypes as C
    pref = B.nvstrings.to_device(["https://example.com/"]) if hasattr(B.nvstrings, "to_device") else None
    t0 = time.perf_counter(); cat = nvcategory.from_strings(c4); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("plain   K=%d keys=%d  %.1f ms" % (K, cat.keys_size(), (t1 - t0) * 1e3), flush=True)
    del cat
    pc = c4.replace("^", "https://example.com/")  # prefix every row
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter(); cat = nvcategory.from_strings(pc); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("prefixed K=%d keys=%d  %.1f ms" % (K, cat.keys_size(), (t1 - t0) * 1e3), flush=True)
    del cat, pc, c4
