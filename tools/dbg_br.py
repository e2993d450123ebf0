import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import gpuutil
g = gpuutil.synth(3, 9_000_000, 200_000)
pat, repl = r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"
a = g.replace_with_backrefs(pat, repl).to_host()
os.environ["CS_BACKREFS_TWO_PASS"] = "1"
b = g.replace_with_backrefs(pat, repl).to_host()
src = g.to_host()
n = 0
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print(i, i % 64, repr(src[i]), "\n   fast", repr(x), "\n   slow", repr(y))
        n += 1
        if n > 6: break
print("diffs shown", n)
