"""`nvstrings` -- host-side mirror of the reference's Python module for the hot path.

Same function / method names, argument defaults and None<->null mapping as
/root/reference/python/nvstrings.py (cited per method), implemented over the
C ABI in include/custrings_amd.h via ctypes.  Every method is one C call on the
MI355X back-end; there is no Python or CPU implementation of any string op
here.  Reference methods outside the hot path raise NotImplementedError.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import lib, check, b, addr

__all__ = ["to_device", "from_strings", "from_offsets", "free", "bind_cpointer", "create_from_ipc", "nvstrings"]


def to_device(strs):
    """nvstrings.py:4-24 -- list of str/None -> device column (NVStrings::create_from_array)."""
    _lib.ensure_init()
    if isinstance(strs, str) or strs is None:
        strs = [strs]
    strs = list(strs)
    n = len(strs)
    enc = [None if s is None else (s.encode("utf8") if isinstance(s, str) else bytes(s)) for s in strs]
    if any(e is not None and b"\0" in e for e in enc):
        # a NUL inside a string: the C-string ingest would cut it there; the offsets ingest is binary safe
        lens = np.array([0 if e is None else len(e) for e in enc], dtype=np.int64)
        offs = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=offs[1:])
        chars = np.frombuffer(b"".join(e for e in enc if e is not None), dtype=np.uint8)
        valid = np.packbits(np.array([e is not None for e in enc], dtype=np.uint8), bitorder="little")
        valid = np.concatenate([valid, np.zeros(8, dtype=np.uint8)])
        return from_offsets64(chars if chars.size else np.zeros(1, dtype=np.uint8), offs, n, valid)
    arr = (C.c_char_p * max(n, 1))(*enc)
    out = C.c_void_p()
    check(lib.cs_column_from_host_strings(arr, n, None, C.byref(out)))
    return nvstrings(out.value)


def from_strings(*args):
    """nvstrings.py:27-56 -- concatenate instances / lists of instances into one (row-wise)."""
    cols = []
    for a in args:
        if isinstance(a, (list, tuple)):
            cols.extend(a)
        else:
            cols.append(a)
    hosts = []
    for c in cols:
        hosts.extend(c.to_host())
    return to_device(hosts)


def from_offsets(sbuf, obuf, scount, nbuf=None, ncount=0, bdevmem=False):
    """nvstrings.py:103-150 -- Arrow chars + int32 offsets (+ bitmask) -> instance
    (NVStrings::create_from_offsets, NVStrings.h:116)."""
    _lib.ensure_init()
    pc, k1 = addr(sbuf)
    po, k2 = addr(obuf)
    pn, k3 = addr(nbuf)
    out = C.c_void_p()
    check(lib.cs_column_from_offsets32(pc, int(scount), po, pn, 1 if bdevmem else 0, None, C.byref(out)))
    del k1, k2, k3
    return nvstrings(out.value)


def from_offsets64(chars, offsets, rows, validity=None, bdevmem=False, copy=True):
    """Native ingest (int64 offsets); copy=False wraps caller-owned device buffers."""
    _lib.ensure_init()
    pc, k1 = addr(chars)
    po, k2 = addr(offsets)
    pn, k3 = addr(validity)
    out = C.c_void_p()
    check(lib.cs_column_from_offsets64(pc, int(rows), po, pn, 1 if bdevmem else 0, 1 if copy else 0, None, C.byref(out)))
    r = nvstrings(out.value)
    if not copy:
        r._keep = (k1, k2, k3, chars, offsets)
    return r


IPC_COLUMN_BYTES = 3 * 64 + 3 * 8 + 4 * 4  # sizeof(cs_ipc_column), include/custrings_amd.h


def create_from_ipc(ipc_data):
    """nvstrings.py:348-360 -- an instance over the buffers another process of this node exported with
    get_ipc_data() (NVStrings::create_from_ipc, NVStrings.h:132): mapped through HIP IPC, no copy.  The
    exporting instance must stay alive while this one is in use."""
    _lib.ensure_init()
    rec = C.create_string_buffer(bytes(ipc_data), IPC_COLUMN_BYTES)
    out = C.c_void_p()
    check(lib.cs_column_ipc_import(rec, C.byref(out)))
    return nvstrings(out.value)


def free(dstrs):
    """nvstrings.py:363-367."""
    if dstrs is not None:
        dstrs._destroy()


def bind_cpointer(cptr, own=True):
    """nvstrings.py:370-377 -- wrap an existing cs_column* handle."""
    if cptr == 0 or cptr is None:
        return None
    return nvstrings(cptr, own)


_NOT_BUILT = (
    "compare hash stoi stol stof stod htoi to_booleans ip2int timestamp2int "
    "get repeat pad ljust center rjust zfill wrap slice slice_from "
    "slice_replace insert fillna capitalize swapcase title index rindex "
    "find_from rfind match_strings startswith endswith isalnum "
    "isalpha isdigit isspace isdecimal isnumeric islower isupper is_empty translate "
    "find_multiple url_encode url_decode get_ipc_data"
).split()


def _int_array(values, count=0):
    """list of ints | numpy array | device address (+ count) -> (address, n, on_device, keepalive)"""
    if isinstance(values, int):
        return values, int(count), 1, None
    if hasattr(values, "data_ptr"):  # torch tensor on the device
        return values.data_ptr(), int(count) or values.numel(), 1, values
    a = np.ascontiguousarray(values, dtype=np.int32)
    return a.ctypes.data, len(a), 0, a


def _escape_literal(text):
    """A literal as a pattern of the reference's regex dialect (regcomp.cpp:314-539): metacharacters quoted."""
    return "".join("\\" + ch if ch in "\\^$.|?*+()[]{}" else ch for ch in text)


class nvstrings:
    """Immutable device strings column (reference class: nvstrings.py:380-402).

    Holds a `cs_column*`; every transforming method returns a new instance.
    """

    def __init__(self, cptr, own=True):
        self.m_cptr = cptr
        self._own = own
        self._keep = None

    _cs_abi = True  # m_cptr is a cs_column* (the pyni glue wraps it on demand, see host/pyni_common.h)
    _nv_cptr = None

    def get_ipc_data(self):
        """nvstrings.py:447-462 -- the record another process hands to create_from_ipc (NVStrings::create_ipc_transfer,
        NVStrings.h:214): the HIP IPC handles of this column's buffers, as bytes."""
        rec = C.create_string_buffer(IPC_COLUMN_BYTES)
        check(lib.cs_column_ipc_export(self.m_cptr, rec))
        return rec.raw

    def _destroy(self):
        if self._nv_cptr:  # the C++ instance the glue wrapped around this handle
            import pyniNVStrings

            pyniNVStrings.n_dropWrapper(self._nv_cptr)
            self._nv_cptr = None
        if getattr(self, "m_cptr", None) and self._own:
            lib.cs_column_destroy(self.m_cptr)
        self.m_cptr = 0

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def __getattr__(self, name):
        if name in _NOT_BUILT:
            raise NotImplementedError("nvstrings.%s is outside the accelerated hot path (SURVEY.md section 8)" % name)
        raise AttributeError(name)

    def __str__(self):
        return str(self.to_host())

    def __repr__(self):
        return "<nvstrings count={}>".format(self.size())

    def __len__(self):
        return self.size()

    def __iter__(self):
        raise TypeError("iterable not supported by nvstrings")

    def get_cpointer(self):
        return self.m_cptr

    # ---- egress -------------------------------------------------------------
    def _export64(self):
        rows = self.size()
        nbytes = lib.cs_column_nbytes(self.m_cptr)
        chars = np.empty(max(nbytes, 1), dtype=np.uint8)
        offs = np.zeros(rows + 1, dtype=np.int64)
        valid = np.zeros((rows + 7) // 8 + 1, dtype=np.uint8)
        if rows:
            check(lib.cs_column_export_offsets64(self.m_cptr, chars.ctypes.data, offs.ctypes.data, valid.ctypes.data, 0, None))
        return chars[:nbytes], offs, valid[: (rows + 7) // 8]

    def to_host(self):
        """nvstrings.py:464-482 -- list of str, None for null rows."""
        chars, offs, valid = self._export64()
        rows = len(offs) - 1
        bits = np.unpackbits(valid, bitorder="little")[:rows] if rows else np.zeros(0, dtype=np.uint8)
        data = chars.tobytes()
        o = offs.tolist()
        return [data[o[i] : o[i + 1]].decode("utf8", "surrogateescape") if bits[i] else None for i in range(rows)]

    def to_offsets(self, sbuf, obuf, nbuf=0, bdevmem=False):
        """nvstrings.py:484-517 -- NVStrings::create_offsets (int32 offsets, LSB-first bitmask)."""
        pc, k1 = addr(sbuf)
        po, k2 = addr(obuf)
        pn, k3 = addr(nbuf)
        check(lib.cs_column_export_offsets32(self.m_cptr, pc, po, pn, 1 if bdevmem else 0, None))
        del k1, k2, k3

    def sublist(self, start, end=0, step=0):
        """NVStrings::sublist(start, end, step) (NVStrings.h:261; array.cu:238-260): rows start, start+step, ...
        before `end`.  A list as the first argument is the Python module's form (nvstrings.py:2390): gather."""
        if not isinstance(start, int):
            return self.gather(start, end)
        out = C.c_void_p()
        check(lib.cs_sublist(self.m_cptr, int(start), int(end), int(step), None, C.byref(out)))
        return nvstrings(out.value)

    # ---- re-arrangement (nvstrings.py:2326-2540; array.cu) --------------------------------------------
    def gather(self, indexes, count=0):
        """nvstrings.py:2394-2418 -- rows at the given positions (ints) or where the mask is true (bools)."""
        out = C.c_void_p()
        if not isinstance(indexes, int) and not hasattr(indexes, "data_ptr") and len(indexes) and isinstance(indexes[0], (bool, np.bool_)):
            m = np.ascontiguousarray(indexes, dtype=np.uint8)
            if len(m) != self.size():
                raise ValueError("gather: the mask must have one entry per string")
            check(lib.cs_gather_mask(self.m_cptr, m.ctypes.data, 0, None, C.byref(out)))
            return nvstrings(out.value)
        p, n, dev, keep = _int_array(indexes, count)
        st = lib.cs_gather(self.m_cptr, p, n, dev, None, C.byref(out))
        if st == _lib.CS_ERR_RANGE:
            raise IndexError(_lib.last_error())  # std::out_of_range in the reference
        check(st)
        del keep
        return nvstrings(out.value)

    def scatter(self, strs, indexes):
        """nvstrings.py:2420-2448 -- a copy with row indexes[j] replaced by row j of strs."""
        p, n, dev, keep = _int_array(indexes, strs.size())
        if n != strs.size():
            raise ValueError("scatter: one index per string of strs")
        out = C.c_void_p()
        check(lib.cs_scatter(self.m_cptr, strs.m_cptr, p, dev, None, C.byref(out)))
        del keep
        return nvstrings(out.value)

    def scalar_scatter(self, str, indexes, count):
        """nvstrings.py:2450-2478 -- a copy with the given rows replaced by one string."""
        p, n, dev, keep = _int_array(indexes, count)
        out = C.c_void_p()
        check(lib.cs_scatter_scalar(self.m_cptr, b(str), p, n if not dev else int(count), dev, None, C.byref(out)))
        del keep
        return nvstrings(out.value)

    def remove_strings(self, indexes, count=0):
        """nvstrings.py:2480-2505 -- the rows NOT named by indexes (array.cu:262-300)."""
        p, n, dev, keep = _int_array(indexes, count)
        if dev:
            raise NotImplementedError("remove_strings with a device pointer")
        mask = np.ones(self.size(), dtype=np.uint8)
        idx = np.asarray(keep, dtype=np.int64)
        mask[idx[(idx >= 0) & (idx < self.size())]] = 0
        out = C.c_void_p()
        check(lib.cs_gather_mask(self.m_cptr, mask.ctypes.data, 0, None, C.byref(out)))
        return nvstrings(out.value)

    def add_strings(self, strs):
        """nvstrings.py:2507-2530 -- these rows followed by those of strs."""
        arr = (C.c_void_p * 2)(self.m_cptr, strs.m_cptr)
        out = C.c_void_p()
        check(lib.cs_column_concat(arr, 2, None, C.byref(out)))
        return nvstrings(out.value)

    def copy(self):
        """nvstrings.py:2532-2540."""
        return self.sublist(0, self.size(), 1) if self.size() else to_device([])

    def sort(self, stype=2, asc=True, nullfirst=True):
        """nvstrings.py:2326-2355 -- by name (2), byte length (1) or both (3)."""
        out = C.c_void_p()
        check(lib.cs_sort(self.m_cptr, int(stype), 1 if asc else 0, 1 if nullfirst else 0, None, C.byref(out)))
        return nvstrings(out.value)

    def order(self, stype=2, asc=True, nullfirst=True, devptr=0):
        """nvstrings.py:2357-2388 -- the row indexes in sorted order."""
        rows = self.size()
        if devptr:
            check(lib.cs_order(self.m_cptr, int(stype), 1 if asc else 0, 1 if nullfirst else 0, devptr, 1, None))
            return devptr
        res = np.zeros(max(rows, 1), dtype=np.uint32)
        check(lib.cs_order(self.m_cptr, int(stype), 1 if asc else 0, 1 if nullfirst else 0, res.ctypes.data, 0, None))
        return [int(v) for v in res[:rows]]

    def len(self, devptr=0):
        """nvstrings.py:538-565 -- characters per row; None for null rows in the host list."""
        rows = self.size()
        total = C.c_int64()
        if devptr:
            check(lib.cs_len(self.m_cptr, devptr, 1, None, C.byref(total)))
            return devptr
        res = np.zeros(max(rows, 1), dtype=np.int32)
        check(lib.cs_len(self.m_cptr, res.ctypes.data, 0, None, C.byref(total)))
        return [None if v < 0 else int(v) for v in res[:rows]]

    # ---- combine (nvstrings.py:881-934; combine.cu) ---------------------------------------------------
    def cat(self, others=None, sep=None, na_rep=None):
        """nvstrings.py:881-911 -- row-wise concatenation with one or several other instances; without `others`
        all rows are joined into one string (pystrings.cpp n_cat: join(sep or "", na_rep))."""
        out = C.c_void_p()
        if others is None:
            check(lib.cs_join(self.m_cptr, b(sep if sep is not None else ""), b(na_rep), None, C.byref(out)))
            return nvstrings(out.value)
        if not isinstance(others, (list, tuple)):
            others = [others]
        arr = (C.c_void_p * max(len(others), 1))(*[o.m_cptr for o in others])
        check(lib.cs_cat(self.m_cptr, arr, len(others), b(sep), b(na_rep), None, C.byref(out)))
        return nvstrings(out.value)

    def join(self, sep=""):
        """nvstrings.py:913-934 -- all rows joined into a single string (null rows contribute nothing)."""
        out = C.c_void_p()
        check(lib.cs_join(self.m_cptr, b(sep), None, None, C.byref(out)))
        return nvstrings(out.value)

    def _export_window(self, first, rows):
        return self.sublist(first, first + rows)._export64()

    def size(self):
        """nvstrings.py:519-536."""
        return int(lib.cs_column_rows(self.m_cptr))

    def byte_count(self, vals=0, bdevmem=False):
        """nvstrings.py:567-596 -- per-row byte length (-1 null); returns the total."""
        pv, k = addr(vals)
        total = C.c_int64()
        check(lib.cs_column_byte_count(self.m_cptr, pv, 1 if bdevmem else 0, None, C.byref(total)))
        del k
        return total.value

    def set_null_bitmask(self, nbuf, bdevmem=False):
        """nvstrings.py:598-620 -- returns the number of nulls."""
        pn, k = addr(nbuf)
        cnt = C.c_int64()
        check(lib.cs_column_null_bitarray(self.m_cptr, pn, 0, 1 if bdevmem else 0, None, C.byref(cnt)))
        del k
        return cnt.value

    def null_count(self, emptyisnull=False):
        """nvstrings.py:622-645."""
        rows = self.size()
        tmp = np.zeros((rows + 7) // 8 + 1, dtype=np.uint8)
        cnt = C.c_int64()
        if rows == 0:
            return 0
        check(lib.cs_column_null_bitarray(self.m_cptr, tmp.ctypes.data, 1 if emptyisnull else 0, 0, None, C.byref(cnt)))
        return cnt.value

    def _null_flags(self):
        rows = self.size()
        tmp = np.zeros((rows + 7) // 8 + 1, dtype=np.uint8)
        cnt = C.c_int64()
        if rows:
            check(lib.cs_column_null_bitarray(self.m_cptr, tmp.ctypes.data, 0, 0, None, C.byref(cnt)))
        return np.unpackbits(tmp, bitorder="little")[:rows] == 0

    def device_memory(self):
        """nvstrings.py:2628 -- bytes of device memory held by this column."""
        rows = self.size()
        return int(lib.cs_column_nbytes(self.m_cptr)) + 8 * (rows + 1) + (rows + 7) // 8

    # ---- split ----------------------------------------------------------------
    def split(self, delimiter=None, n=-1):
        """nvstrings.py:1069-1097 -- column-major split; list of nvstrings."""
        arr = C.POINTER(C.c_void_p)()
        ncols = C.c_int()
        check(lib.cs_split(self.m_cptr, b(delimiter), int(n), None, C.byref(arr), C.byref(ncols)))
        out = [nvstrings(arr[i]) for i in range(ncols.value)]
        lib.cs_free(arr)
        return out

    def rsplit(self, delimiter=None, n=-1):
        """nvstrings.py:1099-1127 -- column-major split with the tokens located from the right."""
        arr = C.POINTER(C.c_void_p)()
        ncols = C.c_int()
        check(lib.cs_rsplit(self.m_cptr, b(delimiter), int(n), None, C.byref(arr), C.byref(ncols)))
        out = [nvstrings(arr[i]) for i in range(ncols.value)]
        lib.cs_free(arr)
        return out

    def _record_call(self, fn, delimiter, n, flat):
        rows = self.size()
        loff = np.zeros(rows + 1, dtype=np.int64)
        out = C.c_void_p()
        check(fn(self.m_cptr, b(delimiter), int(n), loff.ctypes.data, 0, None, C.byref(out)))
        f = nvstrings(out.value)
        if flat:
            return f, loff
        nulls = self._null_flags() if rows else []
        return [None if nulls[r] else f.sublist(int(loff[r]), int(loff[r + 1]), 1) for r in range(rows)]

    def split_record(self, delimiter=None, n=-1, flat=False):
        """nvstrings.py:936-967 -- one nvstrings per row holding that row's tokens (None for a null row).
        flat=True returns the native form: (all tokens in row-major order, rows+1 list offsets)."""
        return self._record_call(lib.cs_split_record, delimiter, n, flat)

    def rsplit_record(self, delimiter=None, n=-1, flat=False):
        """nvstrings.py:969-1001 -- split_record with the tokens located from the right."""
        return self._record_call(lib.cs_rsplit_record, delimiter, n, flat)

    def _partition(self, delimiter, from_right, flat):
        out = C.c_void_p()
        check(lib.cs_partition(self.m_cptr, b(delimiter), from_right, None, C.byref(out)))
        if not out.value:
            return []
        f = nvstrings(out.value)
        if flat:
            return f
        return [f.sublist(3 * r, 3 * r + 3, 1) for r in range(self.size())]

    def partition(self, delimiter=" ", flat=False):
        """nvstrings.py:1003-1034 -- per row (head, delimiter, tail) around the first occurrence."""
        return self._partition(delimiter, 0, flat)

    def rpartition(self, delimiter=" ", flat=False):
        """nvstrings.py:1036-1067 -- per row (head, delimiter, tail) around the last occurrence."""
        return self._partition(delimiter, 1, flat)

    # ---- replace ----------------------------------------------------------------
    def replace_multi(self, pats, repls, regex=True):
        """nvstrings.py:1487-1530 -- several patterns (list of str; regex) or literals (list / nvstrings) at once:
        at each position the first that matches is replaced by its replacement (or the single one)."""
        if regex:
            if not isinstance(pats, list):
                raise ValueError("pats must be list of str")
            pats = list(pats)
        else:
            pats = [None if p is None else _escape_literal(p) for p in (pats.to_host() if isinstance(pats, nvstrings) else pats)]
        if isinstance(repls, str):
            repls = to_device([repls])
        if isinstance(repls, list):
            repls = to_device(repls)
        res = [_compile(p) if p is not None else None for p in pats]
        arr = (C.c_void_p * max(len(res), 1))(*[r for r in res])
        out = C.c_void_p()
        try:
            check(lib.cs_replace_re_multi(self.m_cptr, arr, len(res), repls.m_cptr, None, C.byref(out)))
        finally:
            for r in res:
                if r is not None:
                    lib.cs_regex_destroy(r)
        return nvstrings(out.value)

    def replace(self, pat, repl, n=-1, regex=True):
        """nvstrings.py:1460-1485 -- replace_re when regex (default) else literal replace."""
        out = C.c_void_p()
        if regex:
            re = _compile(pat)
            try:
                check(lib.cs_replace_re(self.m_cptr, re, b(repl), int(n), None, C.byref(out)))
            finally:
                lib.cs_regex_destroy(re)
        else:
            check(lib.cs_replace(self.m_cptr, b(pat), b(repl), int(n), None, C.byref(out)))
        return nvstrings(out.value)

    def replace_with_backrefs(self, pat, repl):
        """nvstrings.py:1532-1557 -- regex replace with \\N in repl standing for capture group N of the match."""
        if not pat:
            raise ValueError("nvstrings::replace_with_backrefs parameter cannot be null or empty")
        re = _compile(pat)
        out = C.c_void_p()
        try:
            check(lib.cs_replace_with_backrefs(self.m_cptr, re, b(repl), None, C.byref(out)))
        finally:
            lib.cs_regex_destroy(re)
        return nvstrings(out.value)

    # ---- extract ------------------------------------------------------------------
    def extract(self, pat):
        """nvstrings.py:2127-2157 -- one nvstrings per capture group (column-major)."""
        re = _compile(pat)
        arr = C.POINTER(C.c_void_p)()
        ncols = C.c_int()
        try:
            check(lib.cs_extract(self.m_cptr, re, None, C.byref(arr), C.byref(ncols)))
        finally:
            lib.cs_regex_destroy(re)
        out = [nvstrings(arr[i]) for i in range(ncols.value)]
        if ncols.value:
            lib.cs_free(arr)
        return out

    def findall(self, pat):
        """nvstrings.py:1921-1948 -- column k holds every row's k-th match (column-major)."""
        re = _compile(pat)
        arr = C.POINTER(C.c_void_p)()
        ncols = C.c_int()
        try:
            check(lib.cs_findall(self.m_cptr, re, None, C.byref(arr), C.byref(ncols)))
        finally:
            lib.cs_regex_destroy(re)
        out = [nvstrings(arr[i]) for i in range(ncols.value)]
        if ncols.value:
            lib.cs_free(arr)
        return out

    # ---- record (row-major) forms ---------------------------------------------------
    @staticmethod
    def _records(cols, ragged):
        """Column-major result -> (flat nvstrings in row-major order, list offsets as a host numpy int64 array):
        the native record form (cs_records_from_columns)."""
        import numpy as np

        rows = cols[0].size() if cols else 0
        arr = (C.c_void_p * max(len(cols), 1))(*[c.m_cptr for c in cols])
        loff = np.zeros(rows + 1, dtype=np.int64)
        out = C.c_void_p()
        check(lib.cs_records_from_columns(arr, len(cols), int(ragged), loff.ctypes.data, 0, None, C.byref(out)))
        return nvstrings(out.value), loff

    @staticmethod
    def _rows_of(flat, loff):
        return [flat.sublist(int(loff[r]), int(loff[r + 1])) for r in range(len(loff) - 1)]

    def extract_record(self, pat, flat=False):
        """nvstrings.py:2097-2125 -- per row the capture groups of its first match (null where there is none).
        flat=True returns the native form instead: (one nvstrings in row-major order, list offsets)."""
        cols = self.extract(pat)
        if not cols:
            return []
        f, loff = self._records(cols, False)
        return (f, loff) if flat else self._rows_of(f, loff)

    def findall_record(self, pat, flat=False):
        """nvstrings.py:1891-1919 -- per row all its matches (an empty instance where there is none)."""
        f, loff = self._records(self.findall(pat), True)
        return (f, loff) if flat else self._rows_of(f, loff)

    # ---- strip --------------------------------------------------------------------
    def _strip(self, to_strip, side):
        out = C.c_void_p()
        check(lib.cs_strip(self.m_cptr, b(to_strip), side, None, C.byref(out)))
        return nvstrings(out.value)

    def lstrip(self, to_strip=None):
        """nvstrings.py:1583-1603."""
        return self._strip(to_strip, 1)

    def strip(self, to_strip=None):
        """nvstrings.py:1605-1625."""
        return self._strip(to_strip, 0)

    def rstrip(self, to_strip=None):
        """nvstrings.py:1627-1647."""
        return self._strip(to_strip, 2)

    # ---- case -----------------------------------------------------------------------
    def lower(self):
        """nvstrings.py:1649-1665."""
        out = C.c_void_p()
        check(lib.cs_lower(self.m_cptr, None, C.byref(out)))
        return nvstrings(out.value)

    def upper(self):
        """nvstrings.py:1667-1683."""
        out = C.c_void_p()
        check(lib.cs_upper(self.m_cptr, None, C.byref(out)))
        return nvstrings(out.value)

    # ---- find / contains ----------------------------------------------------------------
    def find(self, sub, start=0, end=None, devptr=0):
        """nvstrings.py:1796-1823 -- char position, -1 miss; host list has None for null rows
        (pystrings.cpp n_find: values < -1 become None)."""
        rows = self.size()
        end = -1 if end is None else int(end)
        found = C.c_int64()
        if devptr:
            check(lib.cs_find(self.m_cptr, b(sub), int(start), end, devptr, 1, None, C.byref(found)))
            return devptr
        res = np.zeros(max(rows, 1), dtype=np.int32)
        check(lib.cs_find(self.m_cptr, b(sub), int(start), end, res.ctypes.data, 0, None, C.byref(found)))
        return [None if v < -1 else int(v) for v in res[:rows]]

    # ---- the rest of the find family (nvstrings.py:647, 1825-1890, 2005-2095, 2550; find.cu) -----------------------
    def _positions(self, call, devptr, n=None):
        rows = self.size() if n is None else n
        found = C.c_int64()
        if devptr:
            check(call(devptr, 1, C.byref(found)))
            return devptr
        res = np.zeros(max(rows, 1), dtype=np.int32)
        check(call(res.ctypes.data, 0, C.byref(found)))
        return res[:rows]

    def rfind(self, sub, start=0, end=None, devptr=0):
        """nvstrings.py:1861-1890 -- the LAST occurrence inside characters [start, end); None for null rows."""
        end = -1 if end is None else int(end)
        res = self._positions(lambda out, dev, f: lib.cs_rfind(self.m_cptr, b(sub), int(start), end, out, dev, None, f), devptr)
        return res if devptr else [None if v < -1 else int(v) for v in res]

    def find_from(self, sub, starts=0, ends=0, devptr=0):
        """nvstrings.py:1825-1859 -- `starts` / `ends`: device pointers to one int32 per row (0 = from the start / to the end)."""
        res = self._positions(lambda out, dev, f: lib.cs_find_from(self.m_cptr, b(sub), starts or None, ends or None, 1, out, dev, None, f), devptr)
        return res if devptr else [None if v < -1 else int(v) for v in res]

    def find_multiple(self, strs, devptr=0):
        """nvstrings.py:2550-2580 -- a row of positions per string, one per target."""
        if strs is None:
            raise ValueError("nvstrings.find_multiple: parameter required")
        targets = strs if isinstance(strs, nvstrings) else to_device(list(strs))
        tc = targets.size()
        if tc == 0:
            raise ValueError("nvstrings.find_multiple empty argument list")
        rows = self.size()
        res = self._positions(lambda out, dev, f: lib.cs_find_multiple(self.m_cptr, targets.m_cptr, out, dev, None, f), devptr, rows * tc)
        if devptr:
            return res
        return [[None if v < -1 else int(v) for v in res[r * tc:(r + 1) * tc]] for r in range(rows)]

    def compare(self, str, devptr=0):
        """nvstrings.py:647-672 -- bytewise difference to `str` (0 = equal); None for null rows in the host list."""
        rows = self.size()
        res = self._positions(lambda out, dev, f: lib.cs_compare(self.m_cptr, b(str), out, dev, None, f), devptr)
        if devptr:
            return res
        nulls = self._null_flags()
        return [None if nulls[i] else int(res[i]) for i in range(rows)]

    def match_strings(self, strs, devptr=0):
        """nvstrings.py:2005-2047 -- row-wise equality with another instance (or list) of the same size."""
        if strs is None:
            raise ValueError("nvstrings.match_strings: parameter required")
        other = strs if isinstance(strs, nvstrings) else to_device(list(strs))
        if other.size() != self.size():
            raise ValueError("nvstrings.match_strings list size must match")
        rows = self.size()
        found = C.c_int64()
        if devptr:
            check(lib.cs_match_strings(self.m_cptr, other.m_cptr, devptr, 1, None, C.byref(found)))
            return devptr
        if rows == 0:
            return []
        res = np.zeros(rows, dtype=np.uint8)
        check(lib.cs_match_strings(self.m_cptr, other.m_cptr, res.ctypes.data, 0, None, C.byref(found)))
        return [bool(v) for v in res]

    def startswith(self, pat, devptr=0):
        """nvstrings.py:2049-2071; null rows -> None in the host list."""
        return self._bools(lambda out, dev, f: lib.cs_startswith(self.m_cptr, b(pat), out, dev, None, f), devptr)

    def endswith(self, pat, devptr=0):
        """nvstrings.py:2073-2095."""
        return self._bools(lambda out, dev, f: lib.cs_endswith(self.m_cptr, b(pat), out, dev, None, f), devptr)

    def _bools(self, call, devptr):
        rows = self.size()
        found = C.c_int64()
        if devptr:
            check(call(devptr, 1, C.byref(found)))
            return devptr
        if rows == 0:
            return []
        res = np.zeros(rows, dtype=np.uint8)
        check(call(res.ctypes.data, 0, C.byref(found)))
        nulls = self._null_flags()
        return [None if nulls[i] else bool(res[i]) for i in range(rows)]

    def contains(self, pat, regex=True, devptr=0):
        """nvstrings.py:1951-1978; null rows -> None in the host list (pystrings.cpp:2654-2662)."""
        if regex:
            re = _compile(pat)
            try:
                return self._bools(lambda p, d, f: lib.cs_contains_re(self.m_cptr, re, p, d, None, f), devptr)
            finally:
                lib.cs_regex_destroy(re)
        return self._bools(lambda p, d, f: lib.cs_contains(self.m_cptr, b(pat), p, d, None, f), devptr)

    def match(self, pat, devptr=0):
        """nvstrings.py:1980-2003."""
        re = _compile(pat)
        try:
            return self._bools(lambda p, d, f: lib.cs_match_re(self.m_cptr, re, p, d, None, f), devptr)
        finally:
            lib.cs_regex_destroy(re)

    def count(self, pat, devptr=0):
        """nvstrings.py:2033-2047 -- occurrences of the pattern per row."""
        rows = self.size()
        found = C.c_int64()
        re = _compile(pat)
        try:
            if devptr:
                check(lib.cs_count_re(self.m_cptr, re, devptr, 1, None, C.byref(found)))
                return devptr
            if rows == 0:
                return []
            res = np.zeros(rows, dtype=np.int32)
            check(lib.cs_count_re(self.m_cptr, re, res.ctypes.data, 0, None, C.byref(found)))
        finally:
            lib.cs_regex_destroy(re)
        nulls = self._null_flags()
        return [None if nulls[i] else int(res[i]) for i in range(rows)]

    # ---- parity helper ---------------------------------------------------------------------
    def digest(self):
        d = C.c_uint64()
        check(lib.cs_column_digest(self.m_cptr, None, C.byref(d)))
        return d.value


def _compile(pat):
    if pat is None:
        raise ValueError("pattern cannot be null")
    re = C.c_void_p()
    check(lib.cs_regex_compile(b(pat), C.byref(re)))
    return re
