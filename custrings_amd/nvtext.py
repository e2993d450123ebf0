"""`nvtext` -- host-side mirror of /root/reference/python/nvtext.py for the hot path
(tokenize + n-grams), over the C ABI."""
import ctypes as C

import numpy as np

from . import nvstrings as _nvs
from ._lib import lib, check, b

__all__ = ["tokenize", "ngrams", "unique_tokens", "token_count", "tokens_counts", "replace_tokens", "normalize_spaces"]


def tokenize(strs, delimiter=None):
    """nvtext.py:7-43 -- every token of every row, in row order, as one column.
    delimiter None = whitespace; otherwise ANY character of `delimiter` separates
    (NVText::tokenize, NVText.h:40; tokens.cu:45-50)."""
    out = C.c_void_p()
    if isinstance(delimiter, (list, _nvs.nvstrings)):  # nvtext.py:38-40: several whole-string delimiters
        d = _nvs.to_device(delimiter) if isinstance(delimiter, list) else delimiter
        check(lib.cs_tokenize_multi(strs.m_cptr, d.m_cptr, None, C.byref(out)))
        return _nvs.nvstrings(out.value)
    check(lib.cs_tokenize(strs.m_cptr, b(delimiter), None, C.byref(out)))
    return _nvs.nvstrings(out.value)


def ngrams(tokens, N=2, sep="_"):
    """nvtext.py:290-319 -- n-grams over the whole token column
    (NVText::create_ngrams, NVText.h:153; ngram.cu:32-110)."""
    out = C.c_void_p()
    check(lib.cs_ngrams(tokens.m_cptr, int(N), b(sep), None, C.byref(out)))
    return _nvs.nvstrings(out.value)


def unique_tokens(strs, delimiter=" "):
    """nvtext.py:46-73 -- the sorted distinct tokens of all rows (NVText::unique_tokens, tokens.cu:262-304)."""
    out = C.c_void_p()
    check(lib.cs_unique_tokens(strs.m_cptr, b(delimiter), None, C.byref(out)))
    return _nvs.nvstrings(out.value)


def token_count(strs, delimiter=" ", devptr=0):
    """nvtext.py:76-101 -- tokens per row (0 for a null row)."""
    rows = strs.size()
    if devptr:
        check(lib.cs_token_count(strs.m_cptr, b(delimiter), devptr, 1, None))
        return devptr
    res = np.zeros(max(rows, 1), dtype=np.uint32)
    check(lib.cs_token_count(strs.m_cptr, b(delimiter), res.ctypes.data, 0, None))
    return [int(v) for v in res[:rows]]


def tokens_counts(strs, tgts, delimiter=" ", devptr=0):
    """nvtext.py:162-191 -- per row, how many of its tokens equal each of tgts (a list per row)."""
    if isinstance(tgts, list):
        tgts = _nvs.to_device(tgts)
    rows, tc = strs.size(), tgts.size()
    if devptr:
        check(lib.cs_tokens_counts(strs.m_cptr, tgts.m_cptr, b(delimiter), devptr, 1, None))
        return devptr
    res = np.zeros(max(rows * tc, 1), dtype=np.uint32)
    check(lib.cs_tokens_counts(strs.m_cptr, tgts.m_cptr, b(delimiter), res.ctypes.data, 0, None))
    return [[int(v) for v in res[r * tc : (r + 1) * tc]] for r in range(rows)]


def replace_tokens(strs, tgts, repls, delimiter=None):
    """nvtext.py:194-233 -- every token equal to one of tgts is replaced by the matching repl (or the single one)."""
    if isinstance(repls, str):
        repls = _nvs.to_device([repls])
    if isinstance(repls, list):
        repls = _nvs.to_device(repls)
    if isinstance(tgts, list):
        tgts = _nvs.to_device(tgts)
    out = C.c_void_p()
    check(lib.cs_replace_tokens(strs.m_cptr, tgts.m_cptr, repls.m_cptr, b(delimiter), None, C.byref(out)))
    return _nvs.nvstrings(out.value) if out.value else None


def normalize_spaces(strs):
    """nvtext.py:236-258 -- tokens of each row joined by single spaces."""
    out = C.c_void_p()
    check(lib.cs_normalize_spaces(strs.m_cptr, None, C.byref(out)))
    return _nvs.nvstrings(out.value) if out.value else None


def __getattr__(name):
    if name in ("contains_strings", "strings_counts", "edit_distance", "scatter_count"):
        raise NotImplementedError("nvtext.%s is outside the accelerated hot path (SURVEY.md section 8)" % name)
    raise AttributeError(name)
