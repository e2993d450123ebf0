#!/usr/bin/env python3
"""Cliff hunt (GPU box): wall time of many op / argument combinations on the C3 column, one line each; anything far above
its neighbours is a path worth a look.  usage: python tools/probe_hunt.py [rows]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import tools.bench_ops as B  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
c3 = B.synth(3, rows)
res8 = torch.empty(rows, dtype=torch.uint8, device="cuda")
res32 = torch.empty(rows, dtype=torch.int32, device="cuda")
IP = B.IPV4
CASES = [
    ("replace_re(IPv4,'<IP>',n=1)", lambda: c3.replace(IP, "<IP>", 1)),
    ("replace_re(IPv4,'')", lambda: c3.replace(IP, "")),
    ("replace_re(\\d+,'#')", lambda: c3.replace(r"\d+", "#")),
    ("replace_re(\\d,'##')", lambda: c3.replace(r"\d", "##")),
    ("replace_re([A-Z]+,'x')", lambda: c3.replace(r"[A-Z]+", "x")),
    ("replace_re(GET|POST,'VERB')", lambda: c3.replace(r"GET|POST", "VERB")),
    ("replace literal('GET','PUT')", lambda: c3.replace("GET", "PUT", regex=False)),
    ("replace literal('/','//')", lambda: c3.replace("/", "//", regex=False)),
    ("contains literal('404')", lambda: c3.contains("404", regex=False, devptr=res8.data_ptr())),
    ("contains_re(40[34])", lambda: c3.contains(r"40[34]", devptr=res8.data_ptr())),
    ("contains_re(^\\d+)", lambda: c3.contains(r"^\d+", devptr=res8.data_ptr())),
    ("contains_re(\\d+$)", lambda: c3.contains(r"\d+$", devptr=res8.data_ptr())),
    ("count_re(\\d+)", lambda: c3.count(r"\d+", devptr=res32.data_ptr())),
    ("count_re(/)", lambda: c3.count(r"/", devptr=res32.data_ptr())),
    ("findall(\\d+) (many columns)", lambda: c3.findall(r"\d+")),
    ("findall(/\\S+)", lambda: c3.findall(r"/\S+")),
    ("extract((GET|POST) (/\\S*))", lambda: c3.extract(r"(GET|POST) (/\S*)")),
    ("extract((\\d+)\\.(\\d+)\\.(\\d+)\\.(\\d+))", lambda: c3.extract(r"(\d+)\.(\d+)\.(\d+)\.(\d+)")),
    ("backrefs((\\d+)\\.(\\d+) -> \\2.\\1)", lambda: c3.replace_with_backrefs(r"(\d+)\.(\d+)", r"\2.\1")),
    ("find(' ')", lambda: c3.find(" ", devptr=res32.data_ptr())),
    ("find('HTTP')", lambda: c3.find("HTTP", devptr=res32.data_ptr())),
    ("split(' ', 2)", lambda: c3.split(" ", 2)),
    ("split('/')", lambda: c3.split("/")),
    ("split('. ') two bytes", lambda: c3.split(". ")),
    ("rsplit() whitespace", lambda: c3.rsplit()),
    ("strip('0123456789. ')", lambda: c3.strip("0123456789. ")),
    ("lstrip()", lambda: c3.lstrip()),
    ("lower", lambda: c3.lower()),
    ("upper", lambda: c3.upper()),
    ("len", lambda: c3.len(devptr=res32.data_ptr())),
    ("slice(5,20)", lambda: c3.slice(5, 20)),
    ("startswith('1')", lambda: c3.startswith("1", devptr=res8.data_ptr())),
]
only = os.environ.get("HUNT_ONLY")
for name, fn in CASES:
    if only and only not in name:
        continue
    try:
        r = fn()
        del r
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        del r
        print("%-44s %8.2f ms" % (name, dt), flush=True)
    except Exception as e:  # an op the mirror does not take in this form
        print("%-44s %s" % (name, type(e).__name__ + ": " + str(e)[:80]), flush=True)
print("fallbacks", int(B.L.cs_fallback_count()))
