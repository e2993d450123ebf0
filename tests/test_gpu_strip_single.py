"""-m gpu: the single-pass strip (cs_rows.hip: k_strip_stream, CS_STRIP_SINGLE=1 -- opt-in, measured slower than the two
passes) gives the two-pass kernels' column bit for bit and the oracle's strings."""
import contextlib
import os
import random

import pytest

import gpuutil



def _has_experiments():
    from custrings_amd import _lib

    return bool(_lib.lib.cs_has_experiments())


# (the one-pass kernels are in the experiments build only -- `make -C custrings_amd/csrc exp`, CS_LIB_PATH=.../libcustrings_amd_exp.so)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _has_experiments(), reason="the product library is built without the experiments (make exp)")]


@contextlib.contextmanager
def env(**kv):
    # (the library reads its switches once: cs_config_set changes them at run time)
    L = gpuutil.lib().lib
    for k, v in kv.items():
        L.cs_config_set(k.encode(), str(v).encode())
    try:
        yield
    finally:
        for k in kv:
            old = os.environ.get(k)
            L.cs_config_set(k.encode(), None if old is None else old.encode())


@pytest.mark.parametrize("kind,rows", [(2, 1_000_000), (3, 300_000), (2, 777), (2, 64), (2, 65), (2, 1), (2, 5000)])
def test_single_pass_strip_equals_two_pass(kind, rows):
    g = gpuutil.synth(kind, 0, rows)
    L = gpuutil.lib()
    for name, arg in (("strip", None), ("lstrip", None), ("rstrip", None), ("strip", " e"), ("strip", "aeiou 0123456789.")):
        want = getattr(g, name)(arg).digest()
        before = int(L.lib.cs_fallback_count())
        with env(CS_STRIP_SINGLE=1):
            got = getattr(g, name)(arg).digest()
        assert got == want, (name, arg)
        assert int(L.lib.cs_fallback_count()) == before, (name, arg)


def test_single_pass_strip_against_the_oracle(gpu_engine, oracle_engine):
    """nulls, empty rows, rows that strip to nothing, a long row among short ones (its tile goes straight from memory),
    non-ASCII members of the set"""
    rnd = random.Random(5)
    s = []
    for i in range(20000):
        k = rnd.randrange(12)
        if k == 0:
            s.append(None)
        elif k == 1:
            s.append("")
        elif k == 2:
            s.append(" \t " * rnd.randrange(1, 9))
        else:
            s.append(" " * rnd.randrange(4) + "".join(rnd.choice("ab é\tz") for _ in range(rnd.randrange(40))) + "\t" * rnd.randrange(3))
    s[7000] = "  " + "x" * 9000 + "   "
    s[7001] = " " * 700 + "y" + " " * 300
    with env(CS_STRIP_SINGLE=1):
        for side in (0, 1, 2):
            for chars in (None, " é", "ab"):
                assert gpu_engine.strip(s, chars, side) == oracle_engine.strip(s, chars, side), (side, chars)
