import ctypes as C, sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from custrings_amd import _lib, nvstrings
L=_lib.lib; _lib.ensure_init(0)
rows=20_000_000
# fixed-width 64-byte ASCII rows: "GET /aaaa 10.1.2.3 200 xxxxx..." built on host once (small pattern tiled)
rng=np.random.default_rng(1)
base=[]
for i in range(4096):
    ip="%d.%d.%d.%d"%tuple(rng.integers(0,256,4))
    s=("GET /p%03d %s 200 "%(i%1000,ip)).encode()
    s=s+b"abcdefgh ijklmnop qrstuvwx yz012345 67890abc defghijk lmnopqrs"[:64-len(s)]
    assert len(s)==64
    base.append(s)
blob=np.frombuffer(b"".join(base),dtype=np.uint8)
chars=np.tile(blob, rows//4096+1)[:rows*64].copy()
offs=(np.arange(rows+1,dtype=np.int64)*64)
col=nvstrings.from_offsets64(chars, offs, rows, None)
re=nvstrings._compile(r"\d+\.\d+\.\d+\.\d+")
def t(fn,reps=3):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): r=fn(); del r
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/reps*1e3
print("fixed-64 rows=%d replace_re %.2f ms split %.2f ms lower %.2f ms"%(rows, t(lambda: col.replace(r"\d+\.\d+\.\d+\.\d+","<IP>")), t(lambda: col.split(" ")), t(lambda: col.lower())))
