import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import tools.bench_ops as B
from custrings_amd import _lib
c = B.synth(3, 100_000_000)
res = torch.empty(c.size(), dtype=torch.uint8, device="cuda")
res32 = torch.empty(c.size(), dtype=torch.int32, device="cuda")
def t(name, fn, n=3):
    r = fn(); del r
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn(); del r
    torch.cuda.synchronize()
    print("%-40s %.3f ms  route %s" % (name, (time.perf_counter() - t0) / n * 1e3, _lib.lib.cs_debug_last_route().decode()), flush=True)
for nm, pat in (("IPV4", B.IPV4), ("IPV4B", B.IPV4B)):
    t("contains " + nm, lambda: c.contains(pat, devptr=res.data_ptr()))
    t("count " + nm, lambda: c.count(pat, devptr=res32.data_ptr()))
    t("findall " + nm, lambda: c.findall(pat))
    t("replace " + nm, lambda: c.replace(pat, "<IP>"))
t("extract quad groups", lambda: c.extract(r"\b(\d{1,3})\.(\d{1,3})\.(\d{1,3})\.(\d{1,3})\b"))
t("extract quad groups plain", lambda: c.extract(r"(\d+)\.(\d+)\.(\d+)\.(\d+)"))
t("backrefs B", lambda: c.replace_with_backrefs(r"\b(\d{1,3})\.(\d{1,3})\.(\d{1,3})\.(\d{1,3})\b", r"\4.\3.\2.\1"))
t("backrefs", lambda: c.replace_with_backrefs(r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"))
