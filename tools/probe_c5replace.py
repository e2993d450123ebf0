import sys, time, torch
sys.path.insert(0, ".")
import tools.bench_ops as B
c5 = B.synth(5, 62_500_000)
for pat, repl in ((r"[aeiou]+", "*"), (r"\bthe\b", "THE")):
    c5.replace(pat, repl); torch.cuda.synchronize(); t0 = time.perf_counter(); r = c5.replace(pat, repl); torch.cuda.synchronize()
    print(pat, repr(repl), "%.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True); del r
print("fallbacks", int(B.L.cs_fallback_count()))
