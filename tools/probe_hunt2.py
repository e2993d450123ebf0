import os, sys, time
sys.path.insert(0, ".")
import torch
import tools.bench_ops as B
rows = 100_000_000
c3 = B.synth(3, rows)
res32 = torch.empty(rows, dtype=torch.int32, device="cuda")
idx = torch.randperm(rows, device="cuda", dtype=torch.int32)
half = c3.sublist(0, rows // 2)
other = c3.sublist(rows // 2, rows)
CASES = [
    ("replace_multi([IPv4, GET], [<IP>, PUT])", lambda: c3.replace_multi([B.IPV4, "GET"], ["<IP>", "PUT"])),
    ("replace_multi(['GET','POST'],['G','P'], regex=False)", lambda: c3.replace_multi(["GET", "POST"], ["G", "P"], regex=False)),
    ("gather(random permutation)", lambda: c3.gather(idx.data_ptr(), rows) if False else c3.gather(idx)),
    ("sublist(0, rows, 2)", lambda: c3.sublist(0, rows, 2)),
    ("copy", lambda: c3.copy()),
    ("order", lambda: half.order(devptr=res32.data_ptr())),
    ("sort (50M rows)", lambda: half.sort()),
    ("cat(other, sep=' ')", lambda: half.cat(other, sep=" ")),
    ("join(',') (50M rows)", lambda: half.join(",")),
    ("split_record(' ') (50M rows)", lambda: half.split_record(" ", flat=True)),
    ("partition(' ') (50M rows)", lambda: half.partition(" ", flat=True)),
    ("findall_record(IPv4) (50M rows)", lambda: half.findall_record(B.IPV4, flat=True)),
    ("null_count", lambda: c3.null_count()),
    ("byte_count", lambda: c3.byte_count(res32.data_ptr(), True)),
    ("digest", lambda: c3.digest()),
]
for name, fn in CASES:
    try:
        r = fn(); del r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        del r
        print("%-54s %9.2f ms" % (name, dt), flush=True)
    except Exception as e:
        print("%-54s %s" % (name, type(e).__name__ + ": " + str(e)[:100]), flush=True)
print("fallbacks", int(B.L.cs_fallback_count()))
