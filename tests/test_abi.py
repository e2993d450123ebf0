"""The C-ABI library loads without a GPU, exports every symbol that
include/custrings_amd.h declares, and refuses compute without a device (there is
no CPU fallback in the product)."""
import ctypes as C
import os
import re

import pytest

import cpulibs

HEADER = os.path.join(cpulibs.ROOT, "include", "custrings_amd.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cs_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported():
    from custrings_amd import _lib

    names = declared_symbols()
    assert len(names) >= 45
    for n in names:
        assert hasattr(_lib.lib, n), "missing export: " + n
    # and the ctypes prototype table covers all of them
    assert set(names) <= set(_lib._PROTOS), set(names) - set(_lib._PROTOS)


def test_regex_compile_is_host_only():
    from custrings_amd import _lib

    re_ = C.c_void_p()
    assert _lib.lib.cs_regex_compile(b"\\d+\\.\\d+", C.byref(re_)) == 0
    assert _lib.lib.cs_regex_inst_count(re_) > 0
    _lib.lib.cs_regex_destroy(re_)


def test_compiled_patterns_are_kept_and_counted(monkeypatch):
    """cs_regex_compile keeps the last 32 compiled patterns: the same pattern gives the same (counted) object, a handle stays
    valid after its pattern has dropped off the list, and CS_REGEX_NO_CACHE compiles afresh."""
    from custrings_amd import _lib

    L = _lib.lib

    def compile_(p):
        h = C.c_void_p()
        assert L.cs_regex_compile(p, C.byref(h)) == 0
        return h

    a, b, c = compile_(b"ke+pt [0-9]"), compile_(b"ke+pt [0-9]"), compile_(b"other")
    assert a.value == b.value and a.value != c.value
    n = L.cs_regex_inst_count(a)
    L.cs_regex_destroy(b)
    others = [compile_(b"p%d+" % i) for i in range(40)]  # pushes the first pattern off the list
    assert L.cs_regex_inst_count(a) == n  # (still alive: this handle holds it)
    d = compile_(b"ke+pt [0-9]")
    assert d.value != a.value and L.cs_regex_inst_count(d) == n
    for h in [a, c, d] + others:
        L.cs_regex_destroy(h)
    monkeypatch.setenv("CS_REGEX_NO_CACHE", "1")
    e, f = compile_(b"fresh"), compile_(b"fresh")
    assert e.value != f.value
    L.cs_regex_destroy(e)
    L.cs_regex_destroy(f)


def test_compute_fails_loudly_without_a_device():
    from custrings_amd import _lib

    if _lib.lib.cs_device_count() > 0:
        pytest.skip("a GPU is visible")
    import custrings_amd

    with pytest.raises(RuntimeError):
        custrings_amd.nvstrings.to_device(["a"])
    out = C.c_void_p()
    st = _lib.lib.cs_synth_column(3, 0, 10, 1, 0, None, C.byref(out))
    assert st == _lib.CS_ERR_NO_DEVICE
    assert "no CPU fallback" in _lib.last_error()


def test_product_does_not_reference_the_oracle():
    root = os.path.join(cpulibs.ROOT, "custrings_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for bad in ("liboracle", "import oracle", "from oracle", "oracle.cpp", "-loracle", "orc_"):
                    assert bad not in txt, (f, bad)
                for line in txt.splitlines():
                    if line.lstrip().startswith("#include"):
                        assert "oracle" not in line, (f, line)
