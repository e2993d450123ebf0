#!/usr/bin/env python3
"""Plausible ops on the 100M-row C3 column, one line each: time, route, fallbacks -- looking for cliffs (an op that leaves the
single-pass kernels and takes ten times what its neighbours take).  python tools/cliff_scan.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402
from custrings_amd import _lib  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
c = B.synth(3, rows)
res8 = torch.empty(rows, dtype=torch.uint8, device="cuda")
res32 = torch.empty(rows, dtype=torch.int32, device="cuda")


def t(name, fn, n=2):
    f0 = int(_lib.lib.cs_fallback_count())
    r = fn()
    del r
    torch.cuda.synchronize()
    m0 = int(_lib.lib.cs_debug_malloc_count())
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
        del r
    torch.cuda.synchronize()
    # (mallocs: hipMalloc calls inside the timed calls -- the pool missed; idle: GB the pool holds afterwards)
    print("%-58s %8.2f ms  route %-16s fallbacks %d  mallocs %d  idle %.0f GB" % (name, (time.perf_counter() - t0) / n * 1e3, _lib.lib.cs_debug_last_route().decode() or "-",
                                                       int(_lib.lib.cs_fallback_count()) - f0, int(_lib.lib.cs_debug_malloc_count()) - m0,
                                                       int(_lib.lib.cs_pool_cached_bytes()) / 1e9), flush=True)


for pat, repl in ((r"\d+", "#"), (r"\d+", "<number>"), (r"\d", "##"), (r"[a-z]+", "w"), (r"\s+", " "), (r" ", "  "), (r"GET|POST", "VERB"), (r"HTTP/1\.[01]", "H"),
                  (r"\[.*\]", "[]"), (r"\[[^\]]*\]", "[]"), (r"^\d+", "N"), (r"\d+$", "N"), (r"[^ ]+$", "LAST"), (r"\.", ""), (r"-", "<a-replacement-of-thirty-two-bytes>"),
                  (r"\b\d{3}\b", "NNN"), (r"(?:\d+\.){3}\d+", "<IP>"), (r"\w+@\w+", "@")):
    t("replace_re %s -> %s" % (pat, repl[:12]), lambda: c.replace(pat, repl))
for pat in (r"\d+", r"[a-z]+", r"\d+\.\d+", r"/\S*", r"\[[^\]]*\]"):
    t("findall %s" % pat, lambda: c.findall(pat))
    t("count_re %s" % pat, lambda: c.count(pat, devptr=res32.data_ptr()))
for pat in (r"(\d+)\.(\d+)", r"\[([^\]]*)\]", r"(GET|POST) (\S+)", r"(\d+)$", r"^(\S+) "):
    t("extract %s" % pat, lambda: c.extract(pat))
for pat, repl in ((r"(\d+)", r"<\1>"), (r"\[([^\]]*)\]", r"(\1)"), (r"(GET|POST) (\S+)", r"\2 \1"), (r"(\w+)/(\w+)", r"\2/\1"), (r"(\d+)\.(\d+)", r"\2,\1")):
    t("backrefs %s -> %s" % (pat, repl), lambda: c.replace_with_backrefs(pat, repl))
for d, n in ((" ", -1), ("  ", -1), (".", 2), (None, -1), ("/", -1), (" - ", -1), ("\"", -1)):
    t("split %r %d" % (d, n), lambda: c.split(d, n))
t("rsplit ' ' 1", lambda: c.rsplit(" ", 1))
t("replace literal ' ' -> '_'", lambda: c.replace(" ", "_", regex=False))
t("replace literal 'GET' -> 'get'", lambda: c.replace("GET", "get", regex=False))
t("contains literal 'POST'", lambda: c.contains("POST", regex=False, devptr=res8.data_ptr()))
t("find '/'", lambda: c.find("/", devptr=res32.data_ptr()))
t("strip", lambda: c.strip())
t("lower", lambda: c.lower())
