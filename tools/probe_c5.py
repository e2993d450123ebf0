import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
import torch
from custrings_amd import _lib, nvstrings, nvtext
L = _lib.lib; _lib.ensure_init(0)
def synth(kind, rows, param=0):
    out = C.c_void_p(); _lib.check(L.cs_synth_column(kind, 0, rows, 20240607, param, None, C.byref(out))); return nvstrings.nvstrings(out.value)
def t(fn, reps=2):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): r = fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
rows = 10_000_000
c5 = synth(5, rows)
resb = torch.empty(rows, dtype=torch.uint8, device="cuda"); resi = torch.empty(rows, dtype=torch.int32, device="cuda")
print("C5 shape rows=%d, %.2f GB chars (rows 40-150 B)" % (rows, L.cs_column_nbytes(c5.m_cptr)/1e9))
for name, fn in [
 ("lower()", lambda: c5.lower()),
 ("strip()", lambda: c5.strip()),
 ("find('ab')", lambda: c5.find("ab", devptr=resi.data_ptr())),
 ("contains('ab')", lambda: c5.contains("ab", regex=False, devptr=resb.data_ptr())),
 ("replace('ab','x')", lambda: c5.replace("ab", "x", regex=False)),
 ("contains_re('[a-c]+x')", lambda: c5.contains(r"[a-c]+x", devptr=resb.data_ptr())),
 ("count_re('[a-c]+x')", lambda: c5.count(r"[a-c]+x", devptr=resi.data_ptr())),
 ("replace_re('[a-c]+x','<>')", lambda: c5.replace(r"[a-c]+x", "<>")),
 ("split(' ')", lambda: c5.split(" ")),
 ("split(' ', 3)", lambda: c5.split(" ", 3)),
 ("split()", lambda: c5.split()),
 ("tokenize()", lambda: nvtext.tokenize(c5)),
]:
    print("%-30s %9.3f ms" % (name, t(fn)), flush=True)
