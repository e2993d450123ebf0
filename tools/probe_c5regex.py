#!/usr/bin/env python3
"""The regex ops VERDICT r05 (next 2) names, on the C5 column (62.5M tweet-like rows of 40-150 bytes) and on C3 (100M log
lines) for the per-byte comparison: ms per call, ns per input KB, the route each took.  GPU box.
usage: python tools/probe_c5regex.py [c5_rows] [c3_rows]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402

import tools.bench_ops as B  # noqa: E402
from custrings_amd import _lib, nvstrings  # noqa: E402

L = _lib.lib
IPV4 = r"\d+\.\d+\.\d+\.\d+"
PATS = [("ipv4", IPV4, "<IP>"), ("gtest", r"(\bin\b)|(\ba\b)|(\bthe\b)", "="), ("mail", r"\w+@\w+", "<m>"), ("ipv4b", r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", "<IP>"),
        ("hash", r"#\w+", "<tag>"), ("vowels", r"[aeiou]+", "_")]


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
        del r
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    c5_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 62_500_000
    c3_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
    for name, kind, rows in (("c5", 5, c5_rows), ("c3", 3, c3_rows)):
        col = B.synth(kind, rows)
        nb = B.nbytes(col)
        if kind == 5:  # the FIRST regex op on a fresh long-row column also builds the column's pieces (cs_virtual.hip), once
            re0 = nvstrings._compile(IPV4)
            tmp = torch.empty(rows, dtype=torch.uint8, device="cuda")
            f = C.c_int64()
            for attempt, what in ((0, "cold pool"), (1, "warm pool: a second fresh column")):
                if attempt:
                    col = None
                    out2 = C.c_void_p()
                    _lib.check(L.cs_synth_column(kind, 0, rows, B.SEED + 7, 0, None, C.byref(out2)))
                    col = nvstrings.nvstrings(out2.value)
                L.cs_prof_reset()
                L.cs_prof_enable(1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _lib.check(L.cs_contains_re(col.m_cptr, re0, C.c_void_p(tmp.data_ptr()), 1, None, C.byref(f)))
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                L.cs_prof_enable(0)
                ks = {}
                for k in ("k_virt_count", "k_virt_write", "k_contains_re"):
                    ms, n = C.c_double(), C.c_int64()
                    L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
                    if n.value:
                        ks[k] = round(ms.value / n.value, 3)
                print(json.dumps({"config": name, "op": "contains_re, FIRST regex op on a fresh column (builds its pieces), " + what, "pattern": "ipv4", "rows": rows,
                                  "ms": round(dt * 1e3, 3), "kernels_ms": ks, "route": L.cs_debug_last_route().decode()}), flush=True)
            L.cs_regex_destroy(re0)
            del tmp
            # ... and the first replace_re on a fresh column also lists the column's rows with bytes >= 0x80 (cs_virtual.hip: OddRows)
            col = None
            out3 = C.c_void_p()
            _lib.check(L.cs_synth_column(kind, 0, rows, B.SEED + 8, 0, None, C.byref(out3)))
            col = nvstrings.nvstrings(out3.value)
            L.cs_prof_reset()
            L.cs_prof_enable(1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = col.replace(IPV4, "<IP>")
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            L.cs_prof_enable(0)
            del r
            ks = {}
            for k in ("k_virt_count", "k_virt_write", "k_odd_masks", "k_tdfa_replace_list", "k_replace_re"):
                ms, n = C.c_double(), C.c_int64()
                L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
                if n.value:
                    ks[k] = round(ms.value / n.value, 3)
            print(json.dumps({"config": name, "op": "replace_re, FIRST regex op on a fresh column (builds its pieces and the list of their rows with bytes >= 0x80), warm pool",
                              "pattern": "ipv4", "rows": rows, "ms": round(dt * 1e3, 3), "kernels_ms": ks, "route": L.cs_debug_last_route().decode()}), flush=True)
            nb = B.nbytes(col)
        import numpy as np

        res = torch.empty(rows, dtype=torch.uint8, device="cuda")
        cnt = torch.empty(rows, dtype=torch.int32, device="cuda")
        for pname, pat, repl in PATS:
            re = nvstrings._compile(pat)
            found = C.c_int64()

            def contains():
                _lib.check(L.cs_contains_re(col.m_cptr, re, C.c_void_p(res.data_ptr()), 1, None, C.byref(found)))

            def count():
                _lib.check(L.cs_count_re(col.m_cptr, re, C.c_void_p(cnt.data_ptr()), 1, None, C.byref(found)))

            for op, fn in (("contains_re", contains), ("count_re", count), ("replace_re", lambda: col.replace(pat, repl))):
                f0 = int(L.cs_fallback_count())
                dt = timed(fn)
                route = L.cs_debug_last_route().decode()
                ks = {}
                if os.environ.get("PROBE_KERNELS"):  # (one more call under the library's kernel timers)
                    L.cs_prof_reset()
                    L.cs_prof_enable(1)
                    r = fn()
                    del r
                    torch.cuda.synchronize()
                    L.cs_prof_enable(0)
                    for k in ("k_contains_re", "k_count_re", "k_tdfa_scan_list", "k_virt_reduce", "k_replace_re"):
                        ms, n = C.c_double(), C.c_int64()
                        L.cs_prof_get(k.encode(), C.byref(ms), C.byref(n))
                        if n.value:
                            ks[k] = round(ms.value / n.value, 3)
                line = {"config": name, "op": op, "pattern": pname, "rows": rows, "ms": round(dt * 1e3, 3), "input_GBps": round(nb / dt / 1e9, 1),
                        "ps_per_byte": round(dt / nb * 1e12, 3), "route": route, "fallbacks": int(L.cs_fallback_count()) - f0}
                if ks:
                    line["kernels_ms"] = ks
                print(json.dumps(line), flush=True)
            L.cs_regex_destroy(re)
        dt = timed(lambda: col.split(" "))
        print(json.dumps({"config": name, "op": "split(' ')", "rows": rows, "ms": round(dt * 1e3, 3), "input_GBps": round(nb / dt / 1e9, 1), "ps_per_byte": round(dt / nb * 1e12, 3),
                          "route": L.cs_debug_last_route().decode()}), flush=True)
        del col


if __name__ == "__main__":
    main()
