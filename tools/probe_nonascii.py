import sys, time, torch
sys.path.insert(0, ".")
import tools.bench_ops as B
rows = 50_000_000
plain = B.synth(3, rows)
accent = plain.replace("/", "é", regex=False)
res32 = torch.empty(rows, dtype=torch.int32, device="cuda")
P = r"[0-9]+\.[0-9]+\.[0-9]+\.[0-9]+"
for cname, c in (("ascii", plain), ("non-ascii", accent)):
    for name, fn in (("replace_re([0-9] quad)", lambda c: c.replace(P, "<IP>")), ("count_re([0-9] quad)", lambda c: c.count(P, devptr=res32.data_ptr())),
                     ("findall([0-9] quad)", lambda c: c.findall(P)), ("replace_re(\\d quad)", lambda c: c.replace(B.IPV4, "<IP>"))):
        r = fn(c); del r; torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(c); torch.cuda.synchronize()
        print("%-10s %-26s %8.2f ms" % (cname, name, (time.perf_counter() - t0) * 1e3), flush=True); del r
print("fallbacks", int(B.L.cs_fallback_count()))
res8 = torch.empty(rows, dtype=torch.uint8, device="cuda")
for cname, c in (("ascii", plain), ("non-ascii", accent)):
    for name, pat in (("contains_re([0-9] quad)", P), ("contains_re(\\d quad)", B.IPV4)):
        c.contains(pat, devptr=res8.data_ptr()); torch.cuda.synchronize(); t0 = time.perf_counter(); c.contains(pat, devptr=res8.data_ptr()); torch.cuda.synchronize()
        print("%-10s %-26s %8.2f ms" % (cname, name, (time.perf_counter() - t0) * 1e3), flush=True)
