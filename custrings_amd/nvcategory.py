"""`nvcategory` -- host-side mirror of /root/reference/python/nvcategory.py for the
hot path (dictionary encoding: sorted unique keys + int32 values), over the C ABI.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import nvstrings as _nvs
from ._lib import lib, check

__all__ = ["to_device", "from_offsets", "from_strings", "from_strings_list", "bind_cpointer", "nvcategory"]


def _build(col_ptr):
    out = C.c_void_p()
    check(lib.cs_category_build(col_ptr, None, C.byref(out)))
    return nvcategory(out.value)


def to_device(strs):
    """nvcategory.py:5-25 -- category straight from a host list."""
    s = _nvs.to_device(strs)
    return _build(s.m_cptr)


def from_offsets(sbuf, obuf, scount, nbuf=None, ncount=0, bdevmem=False):
    """nvcategory.py:28-75 (NVCategory::create_from_offsets, NVCategory.h:101)."""
    s = _nvs.from_offsets(sbuf, obuf, scount, nbuf, ncount, bdevmem)
    return _build(s.m_cptr)


def from_strings(*args):
    """nvcategory.py:78-103 -- one category over the rows of 1..n nvstrings, in order
    (NVCategory::create_from_strings, NVCategory.h:107,114)."""
    return from_strings_list(list(args))


def from_strings_list(list):
    """nvcategory.py:106-128."""
    _lib.ensure_init()
    if len(list) == 1:
        return _build(list[0].m_cptr)
    arr = (C.c_void_p * max(len(list), 1))(*[s.m_cptr for s in list])
    out = C.c_void_p()
    check(lib.cs_column_concat(arr, len(list), None, C.byref(out)))
    allrows = _nvs.nvstrings(out.value)
    return _build(allrows.m_cptr)


def from_categories(cats):
    """NVCategory::create_from_categories (NVCategory.h:121; NVCategory.cu:430-514):
    merged key set, concatenated remapped values."""
    arr = (C.c_void_p * max(len(cats), 1))(*[c.m_cptr for c in cats])
    out = C.c_void_p()
    check(lib.cs_category_merge(arr, len(cats), None, C.byref(out)))
    return nvcategory(out.value)


IPC_CATEGORY_BYTES = (3 * 64 + 3 * 8 + 4 * 4) + 64 + 8  # sizeof(cs_ipc_category)


def create_from_ipc(ipc_data):
    """A category over the buffers another process exported with get_ipc_data() (NVCategory::create_from_ipc,
    NVCategory.h:128; the reference exposes it in C++ only)."""
    _lib.ensure_init()
    rec = C.create_string_buffer(bytes(ipc_data), IPC_CATEGORY_BYTES)
    out = C.c_void_p()
    check(lib.cs_category_ipc_import(rec, C.byref(out)))
    return nvcategory(out.value)


def bind_cpointer(cptr, own=True):
    """nvcategory.py:157-163."""
    if not cptr:
        return None
    return nvcategory(cptr, own)


_NOT_BUILT = "keys_type to_numbers gather_numbers".split()


def _ints(values, count=0):
    if isinstance(values, int):
        return values, int(count), 1, None
    if hasattr(values, "data_ptr"):
        return values.data_ptr(), int(count) or values.numel(), 1, values
    a = np.ascontiguousarray(values, dtype=np.int32)
    return a.ctypes.data, len(a), 0, a


def _checked(status):
    if status == _lib.CS_ERR_RANGE:
        raise IndexError(_lib.last_error())  # std::out_of_range in the reference
    check(status)


class nvcategory:
    """Reference class: nvcategory.py:166-192."""

    def __init__(self, cptr, own=True):
        self.m_cptr = cptr
        self._own = own

    def get_ipc_data(self):
        """The record for create_from_ipc in another process (NVCategory::create_ipc_transfer, NVCategory.h:176)."""
        rec = C.create_string_buffer(IPC_CATEGORY_BYTES)
        check(lib.cs_category_ipc_export(self.m_cptr, rec))
        return rec.raw

    _cs_abi = True  # m_cptr is a cs_category* (the pyni glue wraps it on demand, see host/pyni_common.h)
    _nv_cptr = None

    def __del__(self):
        try:
            if self._nv_cptr:
                import pyniNVCategory

                pyniNVCategory.n_dropWrapper(self._nv_cptr)
                self._nv_cptr = None
            if self.m_cptr and self._own:
                lib.cs_category_destroy(self.m_cptr)
            self.m_cptr = 0
        except Exception:
            pass

    def __getattr__(self, name):
        if name in _NOT_BUILT:
            raise NotImplementedError("nvcategory.%s is outside the accelerated hot path (SURVEY.md section 8)" % name)
        raise AttributeError(name)

    def __str__(self):
        return "keys: " + str(self.keys()) + "\nvalues: " + str(self.values())

    def __repr__(self):
        return "<nvcategory keys={},values={}>".format(self.keys_size(), self.size())

    def get_cpointer(self):
        return self.m_cptr

    def size(self):
        """nvcategory.py:200-219."""
        return int(lib.cs_category_size(self.m_cptr))

    def keys_size(self):
        """nvcategory.py:221-240."""
        return int(lib.cs_category_keys_size(self.m_cptr))

    def keys(self, narr=None):
        """nvcategory.py:242-274 -- the sorted unique keys as an nvstrings."""
        out = C.c_void_p()
        check(lib.cs_category_keys(self.m_cptr, C.byref(out)))
        return _nvs.nvstrings(out.value)

    def values(self, devptr=0, bdevmem=None):
        """nvcategory.py:364-389 -- int32 key index per row.  `devptr`: an int address or a tensor with
        data_ptr() is device memory, a numpy array host memory (bdevmem overrides)."""
        if devptr is not None and not (isinstance(devptr, int) and devptr == 0):
            p, keep = _lib.addr(devptr)
            on_device = 1 if (isinstance(devptr, int) or hasattr(devptr, "data_ptr") or hasattr(devptr, "__cuda_array_interface__")) else 0
            if bdevmem is not None:
                on_device = 1 if bdevmem else 0
            check(lib.cs_category_get_values(self.m_cptr, p, on_device, None))
            return devptr
        n = self.size()
        res = np.zeros(max(n, 1), dtype=np.int32)
        if n:
            check(lib.cs_category_get_values(self.m_cptr, res.ctypes.data, 0, None))
        return res[:n].tolist()

    def values_cpointer(self):
        """nvcategory.py:391-396."""
        return lib.cs_category_values_ptr(self.m_cptr)

    def value_for_index(self, idx):
        """nvcategory.py:324-341."""
        return self.values()[idx]

    def value(self, str):
        """nvcategory.py:343-362 -- index of a key, -1 when absent."""
        k = self.keys().to_host()
        return k.index(str) if str in k else -1

    def indexes_for_key(self, str, devptr=0):
        """nvcategory.py:276-322 -- the rows whose value is the given key."""
        k = self.value(str)
        res = [i for i, v in enumerate(self.values()) if v == k] if k >= 0 else []
        return res

    def _cat_call(self, fn, *args):
        out = C.c_void_p()
        _checked(fn(self.m_cptr, *args, None, C.byref(out)))
        return nvcategory(out.value)

    def to_strings(self):
        """nvcategory.py:419-436 -- the original strings back (NVCategory::to_strings)."""
        out = C.c_void_p()
        check(lib.cs_category_to_strings(self.m_cptr, None, C.byref(out)))
        return _nvs.nvstrings(out.value) if out.value else None

    def gather_strings(self, indexes, count=0):
        """nvcategory.py:438-470 -- keys[indexes[i]] as strings; an index outside the keys raises."""
        p, n, dev, keep = _ints(indexes, count)
        out = C.c_void_p()
        _checked(lib.cs_category_gather_strings(self.m_cptr, p, n, dev, None, C.byref(out)))
        return _nvs.nvstrings(out.value)

    def gather(self, indexes, count=0):
        """nvcategory.py:510-545 -- same keys, the given indexes as values."""
        p, n, dev, keep = _ints(indexes, count)
        return self._cat_call(lib.cs_category_gather, p, n, dev)

    def gather_and_remap(self, indexes, count=0):
        """nvcategory.py:472-508 -- only the keys the indexes name, values renumbered."""
        p, n, dev, keep = _ints(indexes, count)
        return self._cat_call(lib.cs_category_gather_and_remap, p, n, dev)

    def add_strings(self, nvs):
        """nvcategory.py:547-574."""
        return self._cat_call(lib.cs_category_add_strings, nvs.m_cptr)

    def remove_strings(self, nvs):
        """nvcategory.py:576-603."""
        return self._cat_call(lib.cs_category_remove_strings, nvs.m_cptr)

    def merge_category(self, nvcat):
        """nvcategory.py:669-685 -- the other category's new keys are appended behind these keys (NVCategory.cu:1223-1337)."""
        return self._cat_call(lib.cs_category_merge_category, nvcat.m_cptr)

    def merge_and_remap(self, nvcat):
        """nvcategory.py:687-715 -- merged sorted key set, both value lists renumbered."""
        return from_categories([self, nvcat])

    def add_keys(self, strs):
        """nvcategory.py:605-624 (NVCategory::add_keys_and_remap)."""
        return self._cat_call(lib.cs_category_add_keys, strs.m_cptr)

    def remove_keys(self, strs):
        """nvcategory.py:626-645 (NVCategory::remove_keys_and_remap)."""
        return self._cat_call(lib.cs_category_remove_keys, strs.m_cptr)

    def remove_unused_keys(self):
        """nvcategory.py:717-733 (NVCategory::remove_unused_keys_and_remap)."""
        return self._cat_call(lib.cs_category_remove_unused_keys)

    def set_keys(self, strs):
        """nvcategory.py:647-667 (NVCategory::set_keys_and_remap)."""
        return self._cat_call(lib.cs_category_set_keys, strs.m_cptr)
