"""ctypes wrappers for the two CPU-side test libraries:

  * oracle/liboracle.so      -- the oracle (reference-formulation restatement)
  * tests/rowemu/librowemu.so -- the product's per-row device logic run on the host

Both expose the same tiny API (prefix `orc_` / `emu_`), wrapped here by `CpuLib`
working on `Col` values (python lists of bytes / None).  Test infrastructure only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# CS_SANITIZE=1 (tests/test_sanitizers.py, in a child interpreter with libasan preloaded): the ASan + UBSan builds
SANITIZE = bool(os.environ.get("CS_SANITIZE"))
_SUFFIX = "_asan.so" if SANITIZE else ".so"


def _build(target_dir):
    if os.environ.get("CS_CPULIBS_PREBUILT"):  # bench.py's worker processes: the parent built it already
        return
    # one make at a time per directory (pytest-xdist workers start together; a second make would relink a library
    # while the first worker is loading it)
    import fcntl

    with open(os.path.join(target_dir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.run(["make", "-s", "-C", target_dir] + (["asan"] if SANITIZE else []), check=True)


class Col:
    """Host-side strings column: numpy chars/offsets + validity bitmask (or None)."""

    def __init__(self, chars, offsets, validity=None):
        self.chars = np.ascontiguousarray(chars, dtype=np.uint8)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.rows = len(self.offsets) - 1
        if validity is not None:
            validity = np.ascontiguousarray(validity, dtype=np.uint8)
            if validity.size and np.all(np.unpackbits(validity, bitorder="little")[: self.rows] == 1):
                validity = None
        self.validity = validity

    @staticmethod
    def from_list(items):
        """items: list of str | bytes | None"""
        bs = [None if x is None else (x.encode("utf8") if isinstance(x, str) else bytes(x)) for x in items]
        lens = np.array([0 if b is None else len(b) for b in bs], dtype=np.int64)
        offsets = np.zeros(len(bs) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        chars = np.frombuffer(b"".join(b for b in bs if b is not None), dtype=np.uint8)
        validity = None
        if any(b is None for b in bs):
            bits = np.array([0 if b is None else 1 for b in bs], dtype=np.uint8)
            validity = np.packbits(bits, bitorder="little")
        return Col(chars, offsets, validity)

    def valid_bits(self):
        if self.validity is None:
            return np.ones(self.rows, dtype=bool)
        return np.unpackbits(self.validity, bitorder="little")[: self.rows].astype(bool)

    def bitmask(self):
        return np.packbits(self.valid_bits().astype(np.uint8), bitorder="little")

    def to_bytes_list(self):
        v = self.valid_bits()
        data = self.chars.tobytes()
        o = self.offsets
        return [data[o[i] : o[i + 1]] if v[i] else None for i in range(self.rows)]

    def to_list(self):
        return [None if b is None else b.decode("utf8", "surrogateescape") for b in self.to_bytes_list()]

    def same_as(self, other):
        return (
            self.rows == other.rows
            and np.array_equal(self.offsets, other.offsets)
            and np.array_equal(self.chars, other.chars)
            and np.array_equal(self.bitmask(), other.bitmask())
        )


class CpuLib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.p = prefix
        L = self.lib
        vp = C.c_void_p
        self._f("col_create", vp, [C.c_int64, vp, vp, vp])
        self._f("col_free", None, [vp])
        self._f("col_rows", C.c_int64, [vp])
        self._f("col_nbytes", C.c_int64, [vp])
        self._f("col_offsets", vp, [vp])
        self._f("col_chars", vp, [vp])
        self._f("col_bitmask", None, [vp, vp])
        self._f("free", None, [vp])
        self._f("lower", vp, [vp])
        self._f("upper", vp, [vp])
        self._f("strip", vp, [vp, C.c_char_p, C.c_int])
        self._f("find", C.c_int64, [vp, C.c_char_p, C.c_int, C.c_int, vp])
        self._f("contains", C.c_int64, [vp, C.c_char_p, vp])
        self._f("rfind", C.c_int64, [vp, C.c_char_p, C.c_int, C.c_int, vp])
        self._f("find_from", C.c_int64, [vp, C.c_char_p, vp, vp, vp])
        self._f("find_multiple", C.c_int64, [vp, vp, vp])
        self._f("compare", C.c_int64, [vp, C.c_char_p, vp])
        self._f("match_strings", C.c_int64, [vp, vp, vp])
        self._f("startswith", C.c_int64, [vp, C.c_char_p, vp])
        self._f("endswith", C.c_int64, [vp, C.c_char_p, vp])
        self._f("replace", vp, [vp, C.c_char_p, C.c_char_p, C.c_int])
        self._f("split", C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.POINTER(vp))])
        self._f("rsplit", C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.POINTER(vp))])
        self._f("tokenize", vp, [vp, C.c_char_p])
        del L

    def _f(self, name, res, args):
        fn = getattr(self.lib, self.p + name)
        fn.restype = res
        fn.argtypes = args
        setattr(self, "_" + name, fn)

    # -- marshalling
    def put(self, col):
        v = col.validity
        return self._col_create(
            col.rows,
            col.offsets.ctypes.data,
            col.chars.ctypes.data if col.chars.size else None,
            v.ctypes.data if v is not None else None,
        )

    def take(self, h):
        rows = self._col_rows(h)
        nbytes = self._col_nbytes(h)
        off = np.ctypeslib.as_array(C.cast(self._col_offsets(h), C.POINTER(C.c_int64)), shape=(rows + 1,)).copy()
        if nbytes:
            chars = np.ctypeslib.as_array(C.cast(self._col_chars(h), C.POINTER(C.c_uint8)), shape=(nbytes,)).copy()
        else:
            chars = np.zeros(0, dtype=np.uint8)
        bm = np.zeros((rows + 7) // 8, dtype=np.uint8)
        if rows:
            self._col_bitmask(h, bm.ctypes.data)
        self._col_free(h)
        return Col(chars, off, bm)

    def _unary(self, fn, col, *args):
        h = self.put(col)
        try:
            o = fn(h, *args)
            if not o:
                raise ValueError("invalid argument")
            return self.take(o)
        finally:
            self._col_free(h)

    @staticmethod
    def _b(s):
        return None if s is None else (s.encode("utf8") if isinstance(s, str) else s)

    # -- ops
    def lower(self, col):
        return self._unary(self._lower, col)

    def upper(self, col):
        return self._unary(self._upper, col)

    def strip(self, col, to_strip=None, side=0):
        return self._unary(self._strip, col, self._b(to_strip), side)

    def replace(self, col, s, repl, maxrepl=-1):
        return self._unary(self._replace, col, self._b(s), self._b(repl), maxrepl)

    def tokenize(self, col, delim=None):
        return self._unary(self._tokenize, col, self._b(delim))

    def find(self, col, s, start=0, end=-1):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        n = self._find(h, self._b(s), start, end, out.ctypes.data)
        self._col_free(h)
        return out, n

    def contains(self, col, s):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint8)
        n = self._contains(h, self._b(s), out.ctypes.data)
        self._col_free(h)
        return out, n

    # the rest of the find family (find.cu): results + the count the reference returns
    def rfind(self, col, s, start=0, end=-1):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        n = self._rfind(h, self._b(s), start, end, out.ctypes.data)
        self._col_free(h)
        return out, n

    def find_from(self, col, s, starts=None, ends=None):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        st = None if starts is None else np.ascontiguousarray(starts, dtype=np.int32)
        en = None if ends is None else np.ascontiguousarray(ends, dtype=np.int32)
        n = self._find_from(h, self._b(s), None if st is None else st.ctypes.data, None if en is None else en.ctypes.data, out.ctypes.data)
        self._col_free(h)
        return out, n

    def find_multiple(self, col, targets):
        h, t = self.put(col), self.put(targets)
        out = np.zeros(col.rows * targets.rows, dtype=np.int32)
        n = self._find_multiple(h, t, out.ctypes.data)
        self._col_free(h)
        self._col_free(t)
        return out, n

    def compare(self, col, s):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        n = self._compare(h, self._b(s), out.ctypes.data)
        self._col_free(h)
        return out, n

    def match_strings(self, col, other):
        h, t = self.put(col), self.put(other)
        out = np.zeros(col.rows, dtype=np.uint8)
        n = self._match_strings(h, t, out.ctypes.data)
        self._col_free(h)
        self._col_free(t)
        if n == -2:
            raise ValueError("sizes must match")
        return out, n

    def startswith(self, col, s):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint8)
        n = self._startswith(h, self._b(s), out.ctypes.data)
        self._col_free(h)
        return out, n

    def endswith(self, col, s):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint8)
        n = self._endswith(h, self._b(s), out.ctypes.data)
        self._col_free(h)
        return out, n

    def rsplit(self, col, delim=None, maxsplit=-1):
        return self.split(col, delim, maxsplit, fn=self._rsplit)

    def split(self, col, delim=None, maxsplit=-1, fn=None):
        h = self.put(col)
        arr = C.POINTER(C.c_void_p)()
        n = (fn or self._split)(h, self._b(delim), maxsplit, C.byref(arr))
        cols = [self.take(arr[i]) for i in range(n)]
        self._free(arr)
        self._col_free(h)
        return cols


class Oracle(CpuLib):
    def __init__(self):
        _build(os.path.join(ROOT, "oracle"))
        super().__init__(os.path.join(ROOT, "oracle", "liboracle" + _SUFFIX), "orc_")
        vp = C.c_void_p
        self._f("contains_re", C.c_int64, [vp, vp, C.c_int, vp])
        self._f("count_re", C.c_int64, [vp, vp, vp])
        self._f("replace_re", vp, [vp, vp, C.c_char_p, C.c_int])
        self._f("extract", C.c_int, [vp, vp, C.POINTER(C.POINTER(vp))])
        self._f("findall", C.c_int, [vp, vp, C.POINTER(C.POINTER(vp))])
        self._f("replace_with_backrefs", vp, [vp, vp, C.c_char_p])
        self._f("category", vp, [vp, vp])
        self._f("ngrams", vp, [vp, C.c_uint, C.c_char_p])
        self._f("synth", vp, [C.c_int, C.c_int64, C.c_int64, C.c_uint64, C.c_int64])
        self._f("digest", C.c_uint64, [vp])
        # second part (oracle_round2.inc)
        i32, i64 = C.c_int, C.c_int64
        self._f("len", i64, [vp, vp])
        self._f("gather", vp, [vp, vp, i64])
        self._f("sublist", vp, [vp, C.c_uint, C.c_uint, i32])
        self._f("order", None, [vp, i32, i32, i32, vp])
        self._f("sort", vp, [vp, i32, i32, i32])
        self._f("scatter", vp, [vp, vp, C.c_char_p, i32, vp, i64])
        self._f("cat", vp, [vp, vp, i32, C.c_char_p, C.c_char_p])
        self._f("join", vp, [vp, C.c_char_p, C.c_char_p])
        self._f("split_record", vp, [vp, C.c_char_p, i32, vp])
        self._f("rsplit_record", vp, [vp, C.c_char_p, i32, vp])
        self._f("partition", vp, [vp, C.c_char_p, i32])
        self._f("replace_multi", vp, [vp, vp, i32, vp])
        self._f("cat_gather_strings", vp, [vp, vp, i64, i32])
        self._f("cat_gather_and_remap", vp, [vp, vp, i64, vp])
        self._f("cat_add_keys_and_remap", vp, [vp, vp, i64, vp, vp])
        self._f("token_count", None, [vp, C.c_char_p, vp])
        self._f("unique_tokens", vp, [vp, C.c_char_p])
        self._f("tokens_counts", None, [vp, vp, C.c_char_p, vp])
        self._f("replace_tokens", vp, [vp, vp, vp, C.c_char_p])
        self._f("normalize_spaces", vp, [vp])
        self._f("tokenize_multi", vp, [vp, vp])

    # ---- second part: array / combine / records / multi-pattern replace / category remap / text counters
    def _call(self, fn, cols, *args, err=ValueError):
        """fn(handles of `cols`..., *args) -> column; a NULL result is the reference's exception"""
        hs = [self.put(c) for c in cols]
        try:
            o = fn(*hs, *args)
            if not o:
                raise err("the reference raises here")
            return self.take(o)
        finally:
            for h in hs:
                self._col_free(h)

    def len(self, col):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        total = self._len(h, out.ctypes.data)
        self._col_free(h)
        return out, total

    def gather(self, col, pos):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        return self._call(self._gather, [col], pos.ctypes.data, len(pos), err=IndexError)

    def sublist(self, col, start, end, step):
        return self._call(self._sublist, [col], start, end, step)

    def order(self, col, stype, ascending=True, nullfirst=True):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint32)
        self._order(h, stype, int(ascending), int(nullfirst), out.ctypes.data)
        self._col_free(h)
        return out

    def sort(self, col, stype, ascending=True, nullfirst=True):
        return self._call(self._sort, [col], stype, int(ascending), int(nullfirst))

    def scatter(self, col, strs, pos):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        if isinstance(strs, Col):
            hs = self.put(strs)
            try:
                return self._call(self._scatter, [col], hs, None, 0, pos.ctypes.data, len(pos))
            finally:
                self._col_free(hs)
        return self._call(self._scatter, [col], None, self._b(strs) or b"", 1 if strs is None else 0, pos.ctypes.data, len(pos))

    def cat(self, col, others, sep=None, narep=None):
        hs = [self.put(o) for o in others]
        arr = (C.c_void_p * max(len(hs), 1))(*hs)
        try:
            return self._call(self._cat, [col], arr, len(hs), self._b(sep), self._b(narep))
        finally:
            for h in hs:
                self._col_free(h)

    def join(self, col, delim, narep=None):
        return self._call(self._join, [col], self._b(delim), self._b(narep))

    def _records(self, fn, col, delim, maxsplit):
        h = self.put(col)
        lst = np.zeros(col.rows + 1, dtype=np.int64)
        o = fn(h, self._b(delim), maxsplit, lst.ctypes.data)
        self._col_free(h)
        return self.take(o), lst

    def split_record(self, col, delim=None, maxsplit=-1):
        return self._records(self._split_record, col, delim, maxsplit)

    def rsplit_record(self, col, delim=None, maxsplit=-1):
        return self._records(self._rsplit_record, col, delim, maxsplit)

    def partition(self, col, delim, from_right=False):
        return self._call(self._partition, [col], self._b(delim), int(from_right))

    def replace_multi(self, col, blobs, repls):
        blobs = [np.ascontiguousarray(b, dtype=np.int32) for b in blobs]
        arr = (C.c_void_p * len(blobs))(*[b.ctypes.data for b in blobs])
        hr = self.put(repls)
        try:
            return self._call(self._replace_multi, [col], arr, len(blobs), hr)
        finally:
            self._col_free(hr)

    def cat_gather_strings(self, keys, pos, strict=True):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        return self._call(self._cat_gather_strings, [keys], pos.ctypes.data, len(pos), int(strict), err=IndexError)

    def cat_gather_and_remap(self, keys, pos):
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        out = np.zeros(len(pos), dtype=np.int32)
        k = self._call(self._cat_gather_and_remap, [keys], pos.ctypes.data, len(pos), out.ctypes.data, err=IndexError)
        return k, out

    def cat_add_keys_and_remap(self, keys, values, strs):
        values = np.ascontiguousarray(values, dtype=np.int32)
        out = np.zeros(len(values), dtype=np.int32)
        hs = self.put(strs)
        try:
            k = self._call(self._cat_add_keys_and_remap, [keys], values.ctypes.data, len(values), hs, out.ctypes.data)
        finally:
            self._col_free(hs)
        return k, out

    def token_count(self, col, delim=None):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint32)
        self._token_count(h, self._b(delim), out.ctypes.data)
        self._col_free(h)
        return out

    def unique_tokens(self, col, delim=None):
        return self._call(self._unique_tokens, [col], self._b(delim))

    def tokens_counts(self, col, tkns, delim=None):
        h, ht = self.put(col), self.put(tkns)
        out = np.zeros((col.rows, tkns.rows), dtype=np.uint32)
        self._tokens_counts(h, ht, self._b(delim), out.ctypes.data)
        self._col_free(h)
        self._col_free(ht)
        return out

    def replace_tokens(self, col, tgts, repls, delim=None):
        return self._call(self._replace_tokens, [col, tgts, repls], self._b(delim))

    def normalize_spaces(self, col):
        return self._call(self._normalize_spaces, [col])

    def tokenize_multi(self, col, delims):
        return self._call(self._tokenize_multi, [col, delims])

    # regex entry points take a compiled program blob (int32 numpy array)
    def contains_re(self, col, blob, mode=0):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint8)
        n = self._contains_re(h, blob.ctypes.data, mode, out.ctypes.data)
        self._col_free(h)
        return out, n

    def count_re(self, col, blob):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        n = self._count_re(h, blob.ctypes.data, out.ctypes.data)
        self._col_free(h)
        return out, n

    def replace_re(self, col, blob, repl, maxrepl=-1):
        return self._unary(self._replace_re, col, blob.ctypes.data, self._b(repl), maxrepl)

    def replace_with_backrefs(self, col, blob, repl):
        return self._unary(self._replace_with_backrefs, col, blob.ctypes.data, self._b(repl))

    def _columns(self, fn, col, blob):
        h = self.put(col)
        arr = C.POINTER(C.c_void_p)()
        n = fn(h, blob.ctypes.data_as(C.c_void_p), C.byref(arr))
        cols = [self.take(arr[i]) for i in range(n)]
        if n:
            self._free(arr)
        self._col_free(h)
        return cols

    def extract(self, col, blob):
        return self._columns(self._extract, col, blob)

    def findall(self, col, blob):
        return self._columns(self._findall, col, blob)

    def category(self, col):
        h = self.put(col)
        vals = np.zeros(col.rows, dtype=np.int32)
        k = self._category(h, vals.ctypes.data)
        self._col_free(h)
        return self.take(k), vals

    def ngrams(self, col, n=2, sep="_"):
        return self._unary(self._ngrams, col, n, self._b(sep))

    def synth(self, kind, first_row, rows, seed=20240607, param=0):
        return self.take(self._synth(kind, first_row, rows, seed, param))

    def digest(self, col):
        """include/cs_synth_spec.h: the column digest (sum over rows of cs_digest_row) -- what cs_column_digest computes on the device"""
        h = self.put(col)
        d = int(self._digest(h))
        self._col_free(h)
        return d


class RowEmu(CpuLib):
    def __init__(self):
        _build(os.path.join(ROOT, "oracle"))  # generated unicode tables
        _build(os.path.join(ROOT, "tests", "rowemu"))
        super().__init__(os.path.join(ROOT, "tests", "rowemu", "librowemu" + _SUFFIX), "emu_")
        vp = C.c_void_p
        self._f("regex_compile", vp, [C.c_char_p])
        self._f("regex_free", None, [vp])
        self._f("regex_blob", C.c_int, [vp, C.POINTER(vp)])
        self._f("contains_re", C.c_int64, [vp, vp, C.c_int, vp])
        self._f("count_re", C.c_int64, [vp, vp, vp])
        self._f("replace_re", vp, [vp, vp, C.c_char_p, C.c_int])
        self._f("extract", C.c_int, [vp, vp, C.POINTER(C.POINTER(vp))])
        self._f("findall", C.c_int, [vp, vp, C.POINTER(C.POINTER(vp))])
        self._f("replace_with_backrefs", vp, [vp, vp, C.c_char_p])
        self._f("set_engine", None, [C.c_int])
        self._f("regex_tdfa_info", None, [vp, C.POINTER(C.c_int)])
        self._f("regex_units", C.c_int, [vp])
        self._f("regex_chain", C.c_int, [vp])
        self._f("regex_chain_sfx", C.c_int, [vp])
        self._f("regex_chain_rep", C.c_uint64, [vp])
        self._f("set_chain", None, [C.c_int])
        self._f("set_bits", None, [C.c_int])
        self._f("regex_bits_info", None, [vp, C.POINTER(C.c_int)])

    def set_engine(self, e):
        """0 = list simulator (Pike VM) only, 1 = tagged DFA when the program converts"""
        self._set_engine(e)

    def units(self, pattern):
        """(offered, x byte or None, x required) -- the unit decomposition of the replace kernels (regex_tdfa.cpp)"""
        re = self.compile(pattern)
        w = self._regex_units(re)
        self._regex_free(re)
        return bool(w & 1), (chr((w >> 8) & 127) if (w >> 8) & 127 else None), bool((w >> 16) & 1)

    def chain(self, pattern):
        """The chain form (regex_tdfa.h: chain_match) as a string of items, 'R' / 'x' with '+' when repeated or '{m,n}' when
        counted, '\\b' in front / behind when the chain has one, then '|' and the literal suffix when the chain has one -- or None"""
        re = self.compile(pattern)
        w = self._regex_chain(re)
        sfx = self._regex_chain_sfx(re) & 0xFFFFFFFF
        rep = self._regex_chain_rep(re)
        self._regex_free(re)
        n = (w >> 16) & 15
        if not n:
            return None

        def count(k):
            least, most = (rep >> (8 * k)) & 15, (rep >> (8 * k + 4)) & 15
            if (least, most) == (1, 1):
                return ""
            if (least, most) == (1, 0):
                return "+"
            return "{%d,%s}" % (least, most or "") if least != most else "{%d}" % least

        items = "".join(("x" if (w >> (2 * k)) & 1 else "R") + count(k) for k in range(n))
        items = ("\\b" if (w >> 24) & 1 else "") + items + ("\\b" if (w >> 25) & 1 else "")
        sl = (w >> 20) & 7
        if sl:
            items += "|" + "".join(chr((sfx >> (8 * k)) & 255) for k in range(sl))
        return items

    def bits(self, pattern):
        """The bit-parallel form (regex_bits.h) of a pattern: (classes, alternatives, flags) or None when it does not convert"""
        re = self.compile(pattern)
        out = (C.c_int * 4)()
        self._regex_bits_info(re, out)
        self._regex_free(re)
        return (out[1], out[3], out[2]) if out[0] else None

    def set_bits(self, on):
        """1: rows the bit-parallel form takes (plain ASCII, at most 95 bytes) go through it in contains_re / match / count_re / replace_re"""
        self._set_bits(int(on))

    def set_chain(self, on):
        """0: chain patterns keep the unit route in the host emulation of replace_re"""
        self._set_chain(int(on))

    def tdfa_info(self, re):
        out = (C.c_int * 6)()
        self._regex_tdfa_info(re, out)
        return list(out)

    def compile(self, pattern):
        return self._regex_compile(self._b(pattern))

    def blob(self, re):
        p = C.c_void_p()
        n = self._regex_blob(re, C.byref(p))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n,)).copy()

    def contains_re(self, col, re, mode=0):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.uint8)
        n = self._contains_re(h, re, mode, out.ctypes.data)
        self._col_free(h)
        return out, n

    def count_re(self, col, re):
        h = self.put(col)
        out = np.zeros(col.rows, dtype=np.int32)
        n = self._count_re(h, re, out.ctypes.data)
        self._col_free(h)
        return out, n

    def replace_re(self, col, re, repl, maxrepl=-1):
        return self._unary(self._replace_re, col, re, self._b(repl), maxrepl)

    def replace_with_backrefs(self, col, re, repl):
        return self._unary(self._replace_with_backrefs, col, re, self._b(repl))

    def _columns(self, fn, col, re):
        h = self.put(col)
        arr = C.POINTER(C.c_void_p)()
        n = fn(h, re, C.byref(arr))
        cols = [self.take(arr[i]) for i in range(n)]
        if n:
            self._free(arr)
        self._col_free(h)
        return cols

    def extract(self, col, re):
        return self._columns(self._extract, col, re)

    def findall(self, col, re):
        return self._columns(self._findall, col, re)


_REF = None


def ref_regcomp():
    """The real reference regex compiler (oracle/_ref), or None when not built."""
    global _REF
    if _REF is None:
        path = os.path.join(ROOT, "oracle", "_ref", "libref_regcomp.so")
        if not os.path.exists(path) and os.path.isdir("/root/reference"):
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=False)
        _REF = C.CDLL(path) if os.path.exists(path) else False
    return _REF or None


def pack_pattern(s):
    b = s.encode("utf8") if isinstance(s, str) else s
    out, i = [], 0
    while i < len(b):
        c = b[i]
        w = 1 + ((c & 0xF0) == 0xF0) + ((c & 0xE0) == 0xE0) + ((c & 0xC0) == 0xC0) - ((c & 0xC0) == 0x80)
        w = max(w, 1)
        v = 0
        for k in range(w):
            v = (v << 8) | (b[i + k] if i + k < len(b) else 0)
        out.append(v)
        i += w
    out.append(0)
    return (C.c_uint32 * len(out))(*out)


def ref_blob(pattern):
    lib = ref_regcomp()
    p = C.POINTER(C.c_int32)()
    n = lib.ref_regcomp_blob(pack_pattern(pattern), C.byref(p))
    arr = np.array(p[:n], dtype=np.int32)
    lib.ref_free(p)
    return arr
