#!/usr/bin/env python3
"""Dev probe: replace_re with many matches per row (GPU box)."""
import ctypes as C, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from custrings_amd import _lib, nvstrings
L = _lib.lib; _lib.ensure_init(0)
rows = 100_000_000
out = C.c_void_p(); _lib.check(L.cs_synth_column(3, 0, rows, 20240607, 0, None, C.byref(out))); col = nvstrings.nvstrings(out.value)
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for pat, repl in ((r"[aeiou]", "*"), (r"\s+", " "), (r"\d", "#"), (r"[a-z]+", "w"), (r"\s", "__"), (r"[aeiou]", "<v>"), (r"\d+", "<number>"),
                  (r"\d+\.\d+\.\d+\.\d+", "<redacted-ip>")):
    print("replace_re(%r, %r): %.2f ms" % (pat, repl, t(lambda: col.replace(pat, repl))))
print("literal replace(' ', '  '): %.2f ms" % t(lambda: col.replace(" ", "  ", regex=False)))
