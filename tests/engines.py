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

    # the rest of the find family: (results, count) as the reference's C++ methods return them
    def rfind(self, s, sub, start=0, end=-1):
        out, n = self.o.rfind(Col.from_list(s), sub, start, end)
        return out.tolist(), n

    def find_from(self, s, sub, starts=None, ends=None):
        out, n = self.o.find_from(Col.from_list(s), sub, starts, ends)
        return out.tolist(), n

    def find_multiple(self, s, targets):
        out, n = self.o.find_multiple(Col.from_list(s), Col.from_list(targets))
        return out.tolist(), n

    def compare(self, s, sub):
        out, n = self.o.compare(Col.from_list(s), sub)
        return out.tolist(), n

    def match_strings(self, s, t):
        out, n = self.o.match_strings(Col.from_list(s), Col.from_list(t))
        return [bool(x) for x in out], n

    def startswith(self, s, sub):
        out, n = self.o.startswith(Col.from_list(s), sub)
        return [bool(x) for x in out], n

    def endswith(self, s, sub):
        out, n = self.o.endswith(Col.from_list(s), sub)
        return [bool(x) for x in out], n

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

    # record forms (extract_record.cu:47-152, findall_record.cu:39-151): the same per-row find / extract as the
    # column-major forms, one list per row -- every group of the row for extract, the row's matches for findall
    def extract_record(self, s, pat):
        cols = self.extract(s, pat)
        return [[c[i] for c in cols] for i in range(len(s))]

    def findall_record(self, s, pat):
        cols = self.findall(s, pat)
        return [[c[i] for c in cols if c[i] is not None] for i in range(len(s))]

    def category(self, s):
        k, v = self.o.category(Col.from_list(s))
        return k.to_list(), v.tolist()

    def tokenize(self, s, delimiter=None):
        return self.o.tokenize(Col.from_list(s), delimiter).to_list()

    def ngrams(self, s, N=2, sep="_"):
        return self.o.ngrams(Col.from_list(s), N, sep).to_list()



    # ---- second part (oracle_round2.inc); category keys / values travel as python lists here ----
    def len(self, s):
        out, _ = self.o.len(Col.from_list(s))
        return [None if v < 0 else int(v) for v in out]

    def gather(self, s, pos):
        return self.o.gather(Col.from_list(s), pos).to_list()

    def sublist(self, s, start, end, step):
        return self.o.sublist(Col.from_list(s), start, end, step).to_list()

    def sort(self, s, stype=2, asc=True, nullfirst=True):
        return self.o.sort(Col.from_list(s), stype, asc, nullfirst).to_list()

    def order(self, s, stype=2, asc=True, nullfirst=True):
        return self.o.order(Col.from_list(s), stype, asc, nullfirst).tolist()

    def scatter(self, s, strs, pos):
        return self.o.scatter(Col.from_list(s), Col.from_list(strs), pos).to_list()

    def scalar_scatter(self, s, one, pos):
        return self.o.scatter(Col.from_list(s), one, pos).to_list()

    def cat(self, s, others, sep=None, narep=None):
        return self.o.cat(Col.from_list(s), [Col.from_list(x) for x in others], sep, narep).to_list()

    def join(self, s, sep="", narep=None):
        return self.o.join(Col.from_list(s), sep, narep).to_list()

    @staticmethod
    def _rows_of(flat, lst, s):
        f = flat.to_list()
        return [None if s[r] is None else f[lst[r] : lst[r + 1]] for r in range(len(s))]

    def split_record(self, s, delimiter=None, n=-1):
        f, lst = self.o.split_record(Col.from_list(s), delimiter, n)
        return self._rows_of(f, lst, s)

    def rsplit_record(self, s, delimiter=None, n=-1):
        f, lst = self.o.rsplit_record(Col.from_list(s), delimiter, n)
        return self._rows_of(f, lst, s)

    def partition(self, s, delimiter=" ", from_right=False):
        f = self.o.partition(Col.from_list(s), delimiter, from_right).to_list()
        return [f[3 * r : 3 * r + 3] for r in range(len(s))]

    def replace_multi(self, s, pats, repls):
        return self.o.replace_multi(Col.from_list(s), [self._blob(p) for p in pats], Col.from_list(repls)).to_list()

    def token_count(self, s, delimiter=None):
        return self.o.token_count(Col.from_list(s), delimiter).tolist()

    def unique_tokens(self, s, delimiter=None):
        return self.o.unique_tokens(Col.from_list(s), delimiter).to_list()

    def tokens_counts(self, s, tkns, delimiter=None):
        return self.o.tokens_counts(Col.from_list(s), Col.from_list(tkns), delimiter).tolist()

    def replace_tokens(self, s, tgts, repls, delimiter=None):
        return self.o.replace_tokens(Col.from_list(s), Col.from_list(tgts), Col.from_list(repls), delimiter).to_list()

    def normalize_spaces(self, s):
        return self.o.normalize_spaces(Col.from_list(s)).to_list()

    def tokenize_multi(self, s, delims):
        return self.o.tokenize_multi(Col.from_list(s), Col.from_list(delims)).to_list()

    # category family: (keys, values) of category(s), then the member function
    def cat_to_strings(self, s):
        k, v = self.o.category(Col.from_list(s))
        return self.o.cat_gather_strings(k, v, strict=False).to_list()  # NVCategory.cu:977-1009

    def cat_gather_strings(self, s, pos):
        k, _ = self.o.category(Col.from_list(s))
        return self.o.cat_gather_strings(k, pos, strict=True).to_list()

    def cat_gather(self, s, pos):  # NVCategory.cu:1142-1170: same keys, the positions as values (-1 allowed)
        k, _ = self.o.category(Col.from_list(s))
        if any(p < -1 or p >= k.rows for p in pos):
            raise IndexError("out of range")
        return k.to_list(), list(pos)

    def cat_gather_and_remap(self, s, pos):
        k, _ = self.o.category(Col.from_list(s))
        nk, v = self.o.cat_gather_and_remap(k, pos)
        return nk.to_list(), v.tolist()

    def cat_add_strings(self, s, t):  # NVCategory.cu:926-940: category of (rows ++ strs)
        return self.category(self.cat_to_strings(s) + list(t))

    def cat_remove_strings(self, s, t):  # NVCategory.cu:942-975: rows equal to any of strs (null == null) go away
        gone = set(t)
        return self.category([x for x in self.cat_to_strings(s) if x not in gone])

    def cat_add_keys(self, s, t):
        k, v = self.o.category(Col.from_list(s))
        nk, nv = self.o.cat_add_keys_and_remap(k, v, Col.from_list(t))
        return nk.to_list(), nv.tolist()

    @staticmethod
    def _sorted_keys(keys):
        """null first, then unsigned bytewise order (custring.inl:240-261)"""
        return sorted(keys, key=lambda x: (x is not None, b"" if x is None else x.encode("utf8")))

    def _remap(self, keys, values, new_keys):
        where = {k: i for i, k in enumerate(new_keys)}
        return new_keys, [v if v < 0 else where.get(keys[v], -1) for v in values]

    def cat_remove_keys(self, s, t):  # NVCategory.cu:1482-1565
        keys, values = self.category(s)
        if not keys or not t:
            return keys, values
        gone = set(t)
        return self._remap(keys, values, [k for k in keys if k not in gone])

    def cat_remove_unused_keys(self, keys, values):  # NVCategory.cu:1567-1706 (on a given category)
        used = {v for v in values if v >= 0}
        return self._remap(keys, values, [k for i, k in enumerate(keys) if i in used])

    def cat_set_keys(self, s, t):  # NVCategory.cu:1708-1822
        keys, values = self.category(s)
        if not t:
            return [], ([-1] * len(values) if keys else [])
        new_keys = self._sorted_keys(set(t))
        if not keys:
            return new_keys, values
        return self._remap(keys, values, new_keys)

    def cat_merge_category(self, s, t):  # NVCategory.cu:1223-1337: the new keys of cat2 go behind the keys of cat1
        k1, v1 = self.category(s)
        k2, v2 = self.category(t)
        if not k1 or not k2:
            return (k2, v2) if not k1 else (k1, v1)
        have = set(k1)
        merged = k1 + [k for k in k2 if k not in have]
        where = {k: i for i, k in enumerate(merged)}
        return merged, v1 + [v if v < 0 else where[k2[v]] for v in v2]

    def cat_merge_and_remap(self, s, t):  # NVCategory.cu:1339-1345 == create_from_categories
        k1, v1 = self.category(s)
        k2, v2 = self.category(t)
        merged = self._sorted_keys(set(k1) | set(k2))
        where = {k: i for i, k in enumerate(merged)}
        return merged, [where[k1[v]] for v in v1] + [where[k2[v]] for v in v2]


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

    # the rest of the find family: (results, count) as the reference's C++ methods return them
    def rfind(self, s, sub, start=0, end=-1):
        out, n = self.e.rfind(Col.from_list(s), sub, start, end)
        return out.tolist(), n

    def find_from(self, s, sub, starts=None, ends=None):
        out, n = self.e.find_from(Col.from_list(s), sub, starts, ends)
        return out.tolist(), n

    def find_multiple(self, s, targets):
        out, n = self.e.find_multiple(Col.from_list(s), Col.from_list(targets))
        return out.tolist(), n

    def compare(self, s, sub):
        out, n = self.e.compare(Col.from_list(s), sub)
        return out.tolist(), n

    def match_strings(self, s, t):
        out, n = self.e.match_strings(Col.from_list(s), Col.from_list(t))
        return [bool(x) for x in out], n

    def startswith(self, s, sub):
        out, n = self.e.startswith(Col.from_list(s), sub)
        return [bool(x) for x in out], n

    def endswith(self, s, sub):
        out, n = self.e.endswith(Col.from_list(s), sub)
        return [bool(x) for x in out], n

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

    # record forms (extract_record.cu:47-152, findall_record.cu:39-151): the same per-row find / extract as the
    # column-major forms, one list per row -- every group of the row for extract, the row's matches for findall
    def extract_record(self, s, pat):
        cols = self.extract(s, pat)
        return [[c[i] for c in cols] for i in range(len(s))]

    def findall_record(self, s, pat):
        cols = self.findall(s, pat)
        return [[c[i] for c in cols if c[i] is not None] for i in range(len(s))]

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

    # the rest of the find family through the C ABI (host results; (values, count) like the reference's C++ methods)
    def _i32(self, n, call):
        res = np.zeros(max(n, 1), dtype=np.int32)
        found = C.c_int64()
        self.L.check(call(res.ctypes.data, C.byref(found)))
        return res[:n].tolist(), found.value

    def _u8(self, n, call):
        res = np.zeros(max(n, 1), dtype=np.uint8)
        found = C.c_int64()
        self.L.check(call(res.ctypes.data, C.byref(found)))
        return [bool(x) for x in res[:n]], found.value

    def rfind(self, s, sub, start=0, end=-1):
        c = self.col(s)
        return self._i32(len(s), lambda out, f: self.L.lib.cs_rfind(c.m_cptr, sub.encode("utf8"), start, end, out, 0, None, f))

    def find_from(self, s, sub, starts=None, ends=None):
        c = self.col(s)
        st = None if starts is None else np.ascontiguousarray(starts, dtype=np.int32)
        en = None if ends is None else np.ascontiguousarray(ends, dtype=np.int32)
        return self._i32(len(s), lambda out, f: self.L.lib.cs_find_from(c.m_cptr, sub.encode("utf8"), None if st is None else st.ctypes.data,
                                                                     None if en is None else en.ctypes.data, 0, out, 0, None, f))

    def find_multiple(self, s, targets):
        c, t = self.col(s), self.col(targets)
        return self._i32(len(s) * len(targets), lambda out, f: self.L.lib.cs_find_multiple(c.m_cptr, t.m_cptr, out, 0, None, f))

    def compare(self, s, sub):
        c = self.col(s)
        return self._i32(len(s), lambda out, f: self.L.lib.cs_compare(c.m_cptr, sub.encode("utf8"), out, 0, None, f))

    def match_strings(self, s, t):
        c, o = self.col(s), self.col(t)
        if len(s) != len(t):
            raise ValueError("sizes must match")
        return self._u8(len(s), lambda out, f: self.L.lib.cs_match_strings(c.m_cptr, o.m_cptr, out, 0, None, f))

    def startswith(self, s, sub):
        c = self.col(s)
        return self._u8(len(s), lambda out, f: self.L.lib.cs_startswith(c.m_cptr, sub.encode("utf8"), out, 0, None, f))

    def endswith(self, s, sub):
        c = self.col(s)
        return self._u8(len(s), lambda out, f: self.L.lib.cs_endswith(c.m_cptr, sub.encode("utf8"), out, 0, None, f))

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

    def extract_record(self, s, pat):
        return [r.to_host() for r in self.col(s).extract_record(pat)]

    def findall_record(self, s, pat):
        return [r.to_host() for r in self.col(s).findall_record(pat)]

    def category(self, s):
        cat = self.nvc.from_strings(self.col(s))
        return cat.keys().to_host(), cat.values()

    def tokenize(self, s, delimiter=None):
        return self.nvt.tokenize(self.col(s), delimiter).to_host()

    def ngrams(self, s, N=2, sep="_"):
        return self.nvt.ngrams(self.col(s), N, sep).to_host()



    # ---- second part: array / combine / records / multi replace / category family / text counters ----
    def len(self, s):
        return self.col(s).len()

    def gather(self, s, pos):
        return self.col(s).gather(list(pos)).to_host()

    def sublist(self, s, start, end, step):
        return self.col(s).sublist(start, end, step).to_host()

    def sort(self, s, stype=2, asc=True, nullfirst=True):
        return self.col(s).sort(stype, asc, nullfirst).to_host()

    def order(self, s, stype=2, asc=True, nullfirst=True):
        return self.col(s).order(stype, asc, nullfirst)

    def scatter(self, s, strs, pos):
        return self.col(s).scatter(self.col(strs), list(pos)).to_host()

    def scalar_scatter(self, s, one, pos):
        return self.col(s).scalar_scatter(one, list(pos), len(pos)).to_host()

    def cat(self, s, others, sep=None, narep=None):
        return self.col(s).cat([self.col(o) for o in others], sep, narep).to_host()

    def join(self, s, sep="", narep=None):
        return self.col(s).cat(None, sep, narep).to_host()

    def split_record(self, s, delimiter=None, n=-1):
        return [None if r is None else r.to_host() for r in self.col(s).split_record(delimiter, n)]

    def rsplit_record(self, s, delimiter=None, n=-1):
        return [None if r is None else r.to_host() for r in self.col(s).rsplit_record(delimiter, n)]

    def partition(self, s, delimiter=" ", from_right=False):
        c = self.col(s)
        return [r.to_host() for r in (c.rpartition(delimiter) if from_right else c.partition(delimiter))]

    def replace_multi(self, s, pats, repls):
        return self.col(s).replace_multi(list(pats), list(repls)).to_host()

    def token_count(self, s, delimiter=None):
        return self.nvt.token_count(self.col(s), delimiter)

    def unique_tokens(self, s, delimiter=None):
        return self.nvt.unique_tokens(self.col(s), delimiter).to_host()

    def tokens_counts(self, s, tkns, delimiter=None):
        return self.nvt.tokens_counts(self.col(s), self.col(tkns), delimiter)

    def replace_tokens(self, s, tgts, repls, delimiter=None):
        r = self.nvt.replace_tokens(self.col(s), self.col(tgts), self.col(repls), delimiter)
        return None if r is None else r.to_host()

    def normalize_spaces(self, s):
        r = self.nvt.normalize_spaces(self.col(s))
        return None if r is None else r.to_host()

    def tokenize_multi(self, s, delims):
        return self.nvt.tokenize(self.col(s), list(delims)).to_host()

    def _kv(self, cat):
        return cat.keys().to_host(), cat.values()

    def _cat(self, s):
        return self.nvc.from_strings(self.col(s))

    def cat_to_strings(self, s):
        return self._cat(s).to_strings().to_host()

    def cat_gather_strings(self, s, pos):
        return self._cat(s).gather_strings(list(pos)).to_host()

    def cat_gather(self, s, pos):
        return self._kv(self._cat(s).gather(list(pos)))

    def cat_gather_and_remap(self, s, pos):
        return self._kv(self._cat(s).gather_and_remap(list(pos)))

    def cat_add_strings(self, s, t):
        return self._kv(self._cat(s).add_strings(self.col(t)))

    def cat_remove_strings(self, s, t):
        return self._kv(self._cat(s).remove_strings(self.col(t)))

    def cat_add_keys(self, s, t):
        return self._kv(self._cat(s).add_keys(self.col(t)))

    def cat_remove_keys(self, s, t):
        return self._kv(self._cat(s).remove_keys(self.col(t)))

    def cat_set_keys(self, s, t):
        return self._kv(self._cat(s).set_keys(self.col(t)))

    def cat_set_keys_then_remove_unused(self, s, t):
        return self._kv(self._cat(s).set_keys(self.col(t)).remove_unused_keys())

    def cat_merge_category(self, s, t):
        return self._kv(self._cat(s).merge_category(self._cat(t)))

    def cat_merge_and_remap(self, s, t):
        return self._kv(self._cat(s).merge_and_remap(self._cat(t)))


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
    if op in ("rfind", "find_from"):
        vals, _ = eng.rfind(s, a["sub"], a["start"], a["end"]) if op == "rfind" else eng.find_from(s, a["sub"], a.get("starts"), a.get("ends"))
        return [None if (py and v < -1) else v for v in vals]
    if op == "find_multiple":
        vals, _ = eng.find_multiple(s, a["targets"])
        tc = len(a["targets"])
        flat = [None if (py and v < -1) else v for v in vals]
        return [flat[r * tc:(r + 1) * tc] for r in range(len(s))] if py else flat
    if op == "compare":
        return lift(eng.compare(s, a["sub"])[0])
    if op == "match_strings":
        return eng.match_strings(s, a["other"])[0]
    if op in ("startswith", "endswith"):
        return lift(getattr(eng, op)(s, a["sub"])[0])
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
    if op == "findall_col0":
        return eng.findall(s, a["pat"])[0]
    if op == "extract_record":
        return eng.extract_record(s, a["pat"])
    if op == "findall_record":
        return eng.findall_record(s, a["pat"])
    if op == "category":
        k, v = eng.category(s)
        return {"keys": k, "values": v}
    if op == "tokenize":
        return eng.tokenize(s, a["delimiter"])
    if op == "ngrams":
        return eng.ngrams(s, a["N"], a["sep"])
    if op == "tokenize_ngrams":
        return eng.ngrams(eng.tokenize(s, None), a["N"], a["sep"])
    # second part
    if op == "len":
        return eng.len(s)
    if op == "gather":
        return eng.gather(s, a["pos"])
    if op == "sublist":
        return eng.sublist(s, a["start"], a["end"], a["step"])
    if op == "sort":
        return eng.sort(s, a["stype"], a["asc"], a["nullfirst"])
    if op == "order":
        return eng.order(s, a["stype"], a["asc"], a["nullfirst"])
    if op == "scatter":
        return eng.scatter(s, a["strs"], a["pos"])
    if op == "scalar_scatter":
        return eng.scalar_scatter(s, a["str"], a["pos"])
    if op == "cat":
        return eng.cat(s, a["others"], a["sep"], a["narep"])
    if op == "join":
        return eng.join(s, a["sep"], a["narep"])
    if op == "split_record":
        return eng.split_record(s, a["delimiter"], a["n"])
    if op == "rsplit_record":
        return eng.rsplit_record(s, a["delimiter"], a["n"])
    if op == "partition":
        return eng.partition(s, a["delimiter"], False)
    if op == "rpartition":
        return eng.partition(s, a["delimiter"], True)
    if op == "replace_multi":
        return eng.replace_multi(s, a["pats"], a["repls"])
    if op == "token_count":
        return eng.token_count(s, a["delimiter"])
    if op == "unique_tokens":
        return eng.unique_tokens(s, a["delimiter"])
    if op == "tokens_counts":
        return eng.tokens_counts(s, a["tokens"], a["delimiter"])
    if op == "replace_tokens":
        return eng.replace_tokens(s, a["tgts"], a["repls"], a["delimiter"])
    if op == "normalize_spaces":
        return eng.normalize_spaces(s)
    if op == "tokenize_multi":
        return eng.tokenize_multi(s, a["delimiters"])
    if op.startswith("cat_"):
        fn = getattr(eng, op)
        if op == "cat_to_strings":
            return fn(s)
        res = fn(s, a["arg"])
        if isinstance(res, tuple):
            return {"keys": res[0], "values": list(res[1])}
        return res
    raise KeyError(op)
