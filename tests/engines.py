"""Uniform list-in / list-out drivers used by the parity tests.

  OracleEngine -- oracle/liboracle.so (CPU restatement; the checker)
  EmuEngine    -- tests/rowemu: the product's per-row device logic compiled for the host
  GpuEngine    -- the product: libcustrings_amd.so through the C ABI (needs an MI355X)

All engines return results at the C++ API level (bools: null row -> False,
find: null row -> -2, count_re: null row -> 0); `run_case` lifts them to the
python-list level (null -> None) when a golden case asks for it.
"""
import ctypes as C
import json
import os

import numpy as np

import cpulibs
from cpulibs import Col

ROOT = cpulibs.ROOT
GOLDEN = os.path.join(ROOT, "tests", "golden")

_PROGRAMS = None


def golden_programs():
    global _PROGRAMS
    if _PROGRAMS is None:
        with open(os.path.join(GOLDEN, "regex_programs.json")) as f:
            _PROGRAMS = {k: np.array(v, dtype=np.int32) for k, v in json.load(f)["programs"].items()}
    return _PROGRAMS


def product_blob(pattern):
    """Program blob from the PRODUCT's host compiler (cs_regex_compile runs without a GPU)."""
    from custrings_amd import _lib

    re = C.c_void_p()
    _lib.check(_lib.lib.cs_regex_compile(pattern.encode("utf8"), C.byref(re)))
    words = C.POINTER(C.c_int32)()
    n = C.c_int()
    _lib.check(_lib.lib.cs_regex_blob(re, C.byref(words), C.byref(n)))
    arr = np.array(words[: n.value], dtype=np.int32)
    _lib.lib.cs_regex_destroy(re)
    return arr


def reference_blob(pattern):
    """The reference compiler's program: live (oracle/_ref) when built, else the
    committed fixture, else None."""
    if cpulibs.ref_regcomp() is not None:
        try:
            return cpulibs.ref_blob(pattern)
        except Exception:
            pass
    return golden_programs().get(pattern)


class OracleEngine:
    name = "oracle"

    def __init__(self):
        self.o = cpulibs.Oracle()

    def _blob(self, pat):
        blob = reference_blob(pat)
        if blob is None:  # product compiler: pinned word-for-word to the reference's in test_regex_compile.py
            blob = product_blob(pat)
        return np.ascontiguousarray(blob, dtype=np.int32)

    def lower(self, s):
        return self.o.lower(Col.from_list(s)).to_list()

    def upper(self, s):
        return self.o.upper(Col.from_list(s)).to_list()

    def strip(self, s, to_strip=None, side=0):
        return self.o.strip(Col.from_list(s), to_strip, side).to_list()

    def find(self, s, sub, start=0, end=-1):
        out, n = self.o.find(Col.from_list(s), sub, start, end)
        return out.tolist(), n

    def contains(self, s, pat):
        out, n = self.o.contains(Col.from_list(s), pat)
        return [bool(x) for x in out], n

    def replace(self, s, pat, repl, n=-1):
        return self.o.replace(Col.from_list(s), pat, repl, n).to_list()

    def split(self, s, delimiter=None, n=-1):
        return [c.to_list() for c in self.o.split(Col.from_list(s), delimiter, n)]

    def rsplit(self, s, delimiter=None, n=-1):
        return [c.to_list() for c in self.o.rsplit(Col.from_list(s), delimiter, n)]

    def contains_re(self, s, pat):
        out, n = self.o.contains_re(Col.from_list(s), self._blob(pat), 0)
        return [bool(x) for x in out], n

    def match(self, s, pat):
        out, n = self.o.contains_re(Col.from_list(s), self._blob(pat), 1)
        return [bool(x) for x in out], n

    def count_re(self, s, pat):
        out, n = self.o.count_re(Col.from_list(s), self._blob(pat))
        return out.tolist(), n

    def replace_re(self, s, pat, repl, n=-1):
        if pat == "":
            raise ValueError("empty pattern")
        return self.o.replace_re(Col.from_list(s), self._blob(pat), repl, n).to_list()

    def replace_with_backrefs(self, s, pat, repl):
        return self.o.replace_with_backrefs(Col.from_list(s), self._blob(pat), repl).to_list()

    def extract(self, s, pat):
        return [c.to_list() for c in self.o.extract(Col.from_list(s), self._blob(pat))]

    def findall(self, s, pat):
        return [c.to_list() for c in self.o.findall(Col.from_list(s), self._blob(pat))]

    def category(self, s):
        k, v = self.o.category(Col.from_list(s))
        return k.to_list(), v.tolist()

    def tokenize(self, s, delimiter=None):
        return self.o.tokenize(Col.from_list(s), delimiter).to_list()

    def ngrams(self, s, N=2, sep="_"):
        return self.o.ngrams(Col.from_list(s), N, sep).to_list()


class EmuEngine:
    """Product row logic (row_ops.h / regex_vm.h / regex_compile.cpp) on the host."""

    name = "rowemu"

    def __init__(self):
        self.e = cpulibs.RowEmu()

    def lower(self, s):
        return self.e.lower(Col.from_list(s)).to_list()

    def upper(self, s):
        return self.e.upper(Col.from_list(s)).to_list()

    def strip(self, s, to_strip=None, side=0):
        return self.e.strip(Col.from_list(s), to_strip, side).to_list()

    def find(self, s, sub, start=0, end=-1):
        out, n = self.e.find(Col.from_list(s), sub, start, end)
        return out.tolist(), n

    def contains(self, s, pat):
        out, n = self.e.contains(Col.from_list(s), pat)
        return [bool(x) for x in out], n

    def replace(self, s, pat, repl, n=-1):
        return self.e.replace(Col.from_list(s), pat, repl, n).to_list()

    def split(self, s, delimiter=None, n=-1):
        return [c.to_list() for c in self.e.split(Col.from_list(s), delimiter, n)]

    def rsplit(self, s, delimiter=None, n=-1):
        return [c.to_list() for c in self.e.rsplit(Col.from_list(s), delimiter, n)]

    def _re(self, pat):
        return self.e.compile(pat)

    def contains_re(self, s, pat):
        re = self._re(pat)
        out, n = self.e.contains_re(Col.from_list(s), re, 0)
        self.e._regex_free(re)
        return [bool(x) for x in out], n

    def match(self, s, pat):
        re = self._re(pat)
        out, n = self.e.contains_re(Col.from_list(s), re, 1)
        self.e._regex_free(re)
        return [bool(x) for x in out], n

    def count_re(self, s, pat):
        re = self._re(pat)
        out, n = self.e.count_re(Col.from_list(s), re)
        self.e._regex_free(re)
        return out.tolist(), n

    def replace_re(self, s, pat, repl, n=-1):
        re = self._re(pat)
        try:
            return self.e.replace_re(Col.from_list(s), re, repl, n).to_list()
        finally:
            self.e._regex_free(re)

    def replace_with_backrefs(self, s, pat, repl):
        re = self._re(pat)
        try:
            return self.e.replace_with_backrefs(Col.from_list(s), re, repl).to_list()
        finally:
            self.e._regex_free(re)

    def extract(self, s, pat):
        re = self._re(pat)
        try:
            return [c.to_list() for c in self.e.extract(Col.from_list(s), re)]
        finally:
            self.e._regex_free(re)

    def findall(self, s, pat):
        re = self._re(pat)
        try:
            return [c.to_list() for c in self.e.findall(Col.from_list(s), re)]
        finally:
            self.e._regex_free(re)

    def tokenize(self, s, delimiter=None):
        return self.e.tokenize(Col.from_list(s), delimiter).to_list()


class GpuEngine:
    """The product, through the C ABI (raw results) and the nvstrings mirror."""

    name = "gpu"

    def __init__(self):
        import custrings_amd
        from custrings_amd import _lib

        self.nvs = custrings_amd.nvstrings
        self.nvc = custrings_amd.nvcategory
        self.nvt = custrings_amd.nvtext
        self.L = _lib
        _lib.ensure_init()

    def col(self, s):
        return self.nvs.to_device(s)

    def lower(self, s):
        return self.col(s).lower().to_host()

    def upper(self, s):
        return self.col(s).upper().to_host()

    def strip(self, s, to_strip=None, side=0):
        c = self.col(s)
        return (c.strip(to_strip) if side == 0 else c.lstrip(to_strip) if side == 1 else c.rstrip(to_strip)).to_host()

    def find(self, s, sub, start=0, end=-1):
        c = self.col(s)
        res = np.zeros(max(len(s), 1), dtype=np.int32)
        found = C.c_int64()
        self.L.check(self.L.lib.cs_find(c.m_cptr, sub.encode("utf8"), start, end, res.ctypes.data, 0, None, C.byref(found)))
        return res[: len(s)].tolist(), found.value

    def _bools(self, fn, c, n, *args):
        res = np.zeros(max(n, 1), dtype=np.uint8)
        found = C.c_int64()
        self.L.check(fn(c.m_cptr, *args, res.ctypes.data, 0, None, C.byref(found)))
        return [bool(x) for x in res[:n]], found.value

    def contains(self, s, pat):
        return self._bools(self.L.lib.cs_contains, self.col(s), len(s), pat.encode("utf8"))

    def _re(self, pat):
        re = C.c_void_p()
        self.L.check(self.L.lib.cs_regex_compile(pat.encode("utf8"), C.byref(re)))
        return re

    def contains_re(self, s, pat):
        re = self._re(pat)
        try:
            return self._bools(self.L.lib.cs_contains_re, self.col(s), len(s), re)
        finally:
            self.L.lib.cs_regex_destroy(re)

    def match(self, s, pat):
        re = self._re(pat)
        try:
            return self._bools(self.L.lib.cs_match_re, self.col(s), len(s), re)
        finally:
            self.L.lib.cs_regex_destroy(re)

    def count_re(self, s, pat):
        re = self._re(pat)
        c = self.col(s)
        res = np.zeros(max(len(s), 1), dtype=np.int32)
        found = C.c_int64()
        try:
            self.L.check(self.L.lib.cs_count_re(c.m_cptr, re, res.ctypes.data, 0, None, C.byref(found)))
        finally:
            self.L.lib.cs_regex_destroy(re)
        return res[: len(s)].tolist(), found.value

    def replace(self, s, pat, repl, n=-1):
        return self.col(s).replace(pat, repl, n, regex=False).to_host()

    def replace_re(self, s, pat, repl, n=-1):
        return self.col(s).replace(pat, repl, n, regex=True).to_host()

    def split(self, s, delimiter=None, n=-1):
        return [c.to_host() for c in self.col(s).split(delimiter, n)]

    def rsplit(self, s, delimiter=None, n=-1):
        return [c.to_host() for c in self.col(s).rsplit(delimiter, n)]

    def replace_with_backrefs(self, s, pat, repl):
        return self.col(s).replace_with_backrefs(pat, repl).to_host()

    def extract(self, s, pat):
        return [c.to_host() for c in self.col(s).extract(pat)]

    def findall(self, s, pat):
        return [c.to_host() for c in self.col(s).findall(pat)]

    def category(self, s):
        cat = self.nvc.from_strings(self.col(s))
        return cat.keys().to_host(), cat.values()

    def tokenize(self, s, delimiter=None):
        return self.nvt.tokenize(self.col(s), delimiter).to_host()

    def ngrams(self, s, N=2, sep="_"):
        return self.nvt.ngrams(self.col(s), N, sep).to_host()


# ------------------------------------------------------------------ golden ----
def load_cases(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def run_case(eng, case):
    """Runs a golden case; returns the result at the level the case's expectation uses."""
    op, s, a = case["op"], case["input"], case["args"]
    py = case.get("level") == "py"
    nulls = [x is None for x in s]

    def lift(vals):
        return [None if (py and nulls[i]) else v for i, v in enumerate(vals)]

    if op in ("lower", "upper"):
        return getattr(eng, op)(s)
    if op in ("strip", "lstrip", "rstrip"):
        return eng.strip(s, a.get("to_strip"), {"strip": 0, "lstrip": 1, "rstrip": 2}[op])
    if op == "find":
        vals, _ = eng.find(s, a["sub"], a["start"], a["end"])
        return [None if (py and v < -1) else v for v in vals]
    if op == "contains":
        return lift(eng.contains(s, a["pat"])[0])
    if op in ("contains_re", "match", "count_re"):
        return lift(getattr(eng, op)(s, a["pat"])[0])
    if op == "replace":
        return eng.replace(s, a["pat"], a["repl"], a["n"])
    if op == "replace_re":
        return eng.replace_re(s, a["pat"], a["repl"], a["n"])
    if op == "split":
        return eng.split(s, a["delimiter"], a["n"])
    if op == "rsplit":
        return eng.rsplit(s, a["delimiter"], a["n"])
    if op == "extract":
        return eng.extract(s, a["pat"])
    if op == "replace_with_backrefs":
        return eng.replace_with_backrefs(s, a["pat"], a["repl"])
    if op == "findall":
        return eng.findall(s, a["pat"])
    if op == "category":
        k, v = eng.category(s)
        return {"keys": k, "values": v}
    if op == "tokenize":
        return eng.tokenize(s, a["delimiter"])
    if op == "ngrams":
        return eng.ngrams(s, a["N"], a["sep"])
    if op == "tokenize_ngrams":
        return eng.ngrams(eng.tokenize(s, None), a["N"], a["sep"])
    raise KeyError(op)
