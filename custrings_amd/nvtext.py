"""`nvtext` -- host-side mirror of /root/reference/python/nvtext.py for the hot path
(tokenize + n-grams), over the C ABI."""
import ctypes as C

from . import nvstrings as _nvs
from ._lib import lib, check, b

__all__ = ["tokenize", "ngrams"]


def tokenize(strs, delimiter=None):
    """nvtext.py:7-43 -- every token of every row, in row order, as one column.
    delimiter None = whitespace; otherwise ANY character of `delimiter` separates
    (NVText::tokenize, NVText.h:40; tokens.cu:45-50)."""
    out = C.c_void_p()
    check(lib.cs_tokenize(strs.m_cptr, b(delimiter), None, C.byref(out)))
    return _nvs.nvstrings(out.value)


def ngrams(tokens, N=2, sep="_"):
    """nvtext.py:290-319 -- n-grams over the whole token column
    (NVText::create_ngrams, NVText.h:153; ngram.cu:32-110)."""
    out = C.c_void_p()
    check(lib.cs_ngrams(tokens.m_cptr, int(N), b(sep), None, C.byref(out)))
    return _nvs.nvstrings(out.value)


def __getattr__(name):
    if name in ("unique_tokens", "token_count", "contains_strings", "strings_counts", "tokens_counts",
                "replace_tokens", "normalize_spaces", "edit_distance", "scatter_count"):
        raise NotImplementedError("nvtext.%s is outside the accelerated hot path (SURVEY.md section 8)" % name)
    raise AttributeError(name)
