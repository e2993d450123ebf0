"""-m gpu: the single-pass split (cs_split1.hip, CS_SPLIT_SINGLE=1) gives the two-pass kernels' columns bit for bit
(digest over offsets, chars, validity), on the sampled-estimate route too, and hands over to them when it gives up."""
import contextlib
import os

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


def digests(cols):
    return [c.digest() for c in cols]


CALLS = [
    ("split", (" ",), {}),
    ("split", (" ", 3), {}),
    ("split", (" ", 1), {}),
    ("rsplit", (" ", 3), {}),
    ("split", (None,), {}),
    ("split", (None, 4), {}),
    ("split", ("/",), {}),
    ("split", (". ",), {}),
    ("split", ("e",), {}),
]


@pytest.mark.parametrize("kind,rows", [(3, 300_000), (2, 200_000), (3, 777), (3, 64), (3, 65), (3, 1)])
def test_single_pass_equals_two_pass(kind, rows):
    g = gpuutil.synth(kind, 0, rows)
    L = gpuutil.lib()
    for name, args, kw in CALLS:
        want = digests(getattr(g, name)(*args, **kw))
        before = int(L.lib.cs_fallback_count())
        with env(CS_SPLIT_SINGLE=1):
            got = digests(getattr(g, name)(*args, **kw))
        assert got == want, (name, args)
        assert int(L.lib.cs_fallback_count()) == before, (name, args)


def test_single_pass_on_the_sampled_estimate():
    """one sub-tile in 37 sampled: buffers from mean + 8 sigma, the column count from what the sample saw (rows with more
    tokens raise it while the kernel runs; k_split_fixup writes the null rows of the columns that appeared late)"""
    g = gpuutil.synth(3, 0, 400_000)
    want, want5, wantw = digests(g.split(" ")), digests(g.split(" ", 5)), digests(g.split(None))
    with env(CS_SPLIT_SINGLE=1, CS_SPLIT1_SAMPLE=169):
        assert digests(g.split(" ")) == want
        assert digests(g.split(" ", 5)) == want5
        assert digests(g.split(None)) == wantw


def test_single_pass_gives_up_and_the_two_passes_take_over():
    g = gpuutil.synth(3, 0, 300_000)
    L = gpuutil.lib()
    want = digests(g.split(" "))
    before = int(L.lib.cs_fallback_count())
    with env(CS_SPLIT_SINGLE=1, CS_SPLIT1_SHRINK=4):
        assert digests(g.split(" ")) == want
    assert int(L.lib.cs_fallback_count()) > before


def test_single_pass_full_size_c3():
    g = gpuutil.synth(3, 0, 100_000_000)
    want = digests(g.split(" "))
    with env(CS_SPLIT_SINGLE=1):
        got = digests(g.split(" "))
    assert got == want
