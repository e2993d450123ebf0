"""Helpers for the -m gpu tests: device columns <-> host Col, synthetic columns."""
import ctypes as C

import numpy as np

import cpulibs

SEED = 20240607


def lib():
    from custrings_amd import _lib

    _lib.ensure_init()
    return _lib


def synth(kind, first_row, rows, param=0, seed=SEED):
    from custrings_amd import nvstrings

    L = lib()
    out = C.c_void_p()
    L.check(L.lib.cs_synth_column(kind, first_row, rows, seed, param, None, C.byref(out)))
    return nvstrings.nvstrings(out.value)


def to_col(g):
    chars, offs, valid = g._export64()
    return cpulibs.Col(chars, offs, valid)


def from_col(col):
    from custrings_amd import nvstrings

    return nvstrings.from_offsets64(col.chars, col.offsets, col.rows, col.validity)


def assert_same(g, ocol, what=""):
    got = to_col(g)
    assert got.rows == ocol.rows, what
    assert np.array_equal(got.bitmask(), ocol.bitmask()), what + ": validity differs"
    assert np.array_equal(got.offsets, ocol.offsets), what + ": offsets differ"
    assert np.array_equal(got.chars, ocol.chars), what + ": chars differ"


def bools(g, fn_name, pat_or_re):
    L = lib()
    rows = g.size()
    res = np.zeros(max(rows, 1), dtype=np.uint8)
    found = C.c_int64()
    L.check(getattr(L.lib, fn_name)(g.m_cptr, pat_or_re, res.ctypes.data, 0, None, C.byref(found)))
    return res[:rows], found.value


def compile_re(pat):
    from custrings_amd import nvstrings

    return nvstrings._compile(pat)
