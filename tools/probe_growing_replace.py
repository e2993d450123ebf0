import ctypes as C, sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from custrings_amd import _lib, nvstrings
L=_lib.lib; _lib.ensure_init(0)
rows=100_000_000
out=C.c_void_p(); _lib.check(L.cs_synth_column(3,0,rows,20240607,0,None,C.byref(out))); col=nvstrings.nvstrings(out.value)
def t(fn,reps=2):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): r=fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
pat=r"\d+\.\d+\.\d+\.\d+"
print("replace_re '<IP>' %.2f ms | '<redacted-ip>' %.2f ms | match %.2f ms"%(t(lambda: col.replace(pat,"<IP>")), t(lambda: col.replace(pat,"<redacted-ip>")), t(lambda: col.match(pat))))
