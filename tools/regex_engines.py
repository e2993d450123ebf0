#!/usr/bin/env python3
"""Which executor every pattern of the committed fixtures gets (host only: the compile needs no GPU).
Prints one line per pattern: instructions, engine, DFA states / live threads, unit decomposition; and a summary.
usage: python tools/regex_engines.py [> profiles/r03/regex_engines.txt]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from custrings_amd import _lib  # noqa: E402

L = _lib.lib


def engine(pat):
    re = C.c_void_p()
    if L.cs_regex_compile(pat.encode(), C.byref(re)) != 0:
        return None
    e, n = int(L.cs_regex_engine(re)), int(L.cs_regex_inst_count(re))
    L.cs_regex_destroy(re)
    return e, n


def main():
    progs = json.load(open(os.path.join(ROOT, "tests", "golden", "regex_programs.json")))["programs"]
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_tests.json")))
    ref_pats = set()
    for c in ref:
        a = c["args"]
        if "pat" in a and c["op"] not in ("replace", "contains", "find"):
            ref_pats.add(a["pat"])
        for p in a.get("pats", []):
            ref_pats.add(p)
    rows, tally = [], {"dfa": 0, "list": 0, "refused": 0, "ref_dfa": 0, "ref_list": 0}
    for pat in sorted(progs):
        r = engine(pat)
        if r is None:
            tally["refused"] += 1
            continue
        e, n = r
        dfa = bool(e & 1)
        tally["dfa" if dfa else "list"] += 1
        if pat in ref_pats:
            tally["ref_dfa" if dfa else "ref_list"] += 1
        rows.append((pat, n, dfa, e >> 16, (e >> 8) & 15, bool(e & 2), pat in ref_pats))
    for pat, n, dfa, states, thr, units, isref in rows:
        shown = pat if len(pat) <= 60 else pat[:57] + "..."
        print("%-62s %3d inst  %-14s %s%s" % (json.dumps(shown, ensure_ascii=False), n, "tagged DFA" if dfa else "LIST SIMULATOR",
                                             ("%3d states %d threads%s" % (states, thr, " units" if units else "")) if dfa else "",
                                             "  [reference test]" if isref else ""))
    print("\n%d patterns: %d on the tagged DFA, %d on the list simulator, %d refused by the compiler; of the %d patterns the "
          "reference's own tests use: %d DFA, %d list simulator" % (len(progs), tally["dfa"], tally["list"], tally["refused"],
                                                                    tally["ref_dfa"] + tally["ref_list"], tally["ref_dfa"], tally["ref_list"]))


if __name__ == "__main__":
    main()
