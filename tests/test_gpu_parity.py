"""-m gpu: the HIP path, called through the C ABI (libcustrings_amd.so), against
(1) the committed golden vectors, (2) the oracle on seeded random columns,
(3) the oracle on the synthetic benchmark columns, and (4) size-independent
properties at BASELINE.json's full sizes.  Bit-exact: offsets, chars, validity,
bools, ints."""
import ctypes as C

import numpy as np
import pytest

import cpulibs
import engines
import fuzzdata
import gpuutil

pytestmark = pytest.mark.gpu

REF = engines.load_cases("reference_tests.json")
APX = engines.load_cases("survey_appendix_a.json")
IPV4 = r"\d+\.\d+\.\d+\.\d+"
IPV4B = r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b"


@pytest.fixture(scope="module")
def orc():
    return cpulibs.Oracle()


def test_native_library_is_loaded_and_on_gfx950():
    L = gpuutil.lib()
    assert L.lib.cs_device_count() >= 1
    import os

    maps = open("/proc/self/maps").read()
    assert "libcustrings_amd.so" in maps
    assert os.path.exists(L._PATH)


@pytest.mark.parametrize("case", REF + APX, ids=[c["id"] for c in REF + APX])
def test_gpu_golden(gpu_engine, case):
    assert engines.run_case(gpu_engine, case) == case["expect"], case["src"]


def test_gpu_replace_re_rejects_empty_pattern(gpu_engine):
    with pytest.raises(ValueError):
        gpu_engine.replace_re(["a"], "", "x")
    with pytest.raises(ValueError):
        gpu_engine.replace(["a"], "", "x")


@pytest.mark.parametrize("seed", [1, 2])
def test_gpu_vs_oracle_string_ops(gpu_engine, oracle_engine, seed):
    s = fuzzdata.rows(seed, 1500, max_len=40)
    o, g = oracle_engine, gpu_engine
    assert g.lower(s) == o.lower(s)
    assert g.upper(s) == o.upper(s)
    for ts in (None, " ", "ab ", "é ", "\n\t x"):
        for side in (0, 1, 2):
            assert g.strip(s, ts, side) == o.strip(s, ts, side)
    for sub in ("a", "é", "ab", " ", "", "bc", "😀"):
        for st, en in ((0, -1), (1, 5), (3, 2), (2, 100)):
            assert g.find(s, sub, st, en) == o.find(s, sub, st, en), (sub, st, en)
        assert g.contains(s, sub) == o.contains(s, sub)
    for pat, repl in (("a", "xx"), ("é", ""), ("ab", "é"), (" ", "__")):
        for n in (-1, 0, 1, 2):
            assert g.replace(s, pat, repl, n) == o.replace(s, pat, repl, n), (pat, repl, n)
    for d in (None, " ", "a", "é", "ab", ","):
        for n in (-1, 1, 2):
            assert g.split(s, d, n) == o.split(s, d, n), (d, n)
    for d in (None, " ", "_-", "é "):
        assert g.tokenize(s, d) == o.tokenize(s, d)
        toks = o.tokenize(s, d)
        for N in (1, 2, 3):
            assert g.ngrams(toks, N, "_") == o.ngrams(toks, N, "_")
    assert g.category(s) == o.category(s)


PATTERNS = [IPV4, IPV4B, r"a*", r"x*", r"a|aa", r"aa|a", r"a+?", r"\w+", r"\W", r"[\W]", r"\s+", r"^a", r"a$", r"\bc",
            r"\B", r"[a-c]+[x-z]?", r"[^a-c ]+", r"é+", r"[é-ü]", r"(a|b)*c", r".*", r"^$", r"(ab|a)(bc|c)?",
            r"a{2,3}", r"(a|b|c){3}", "a" * 70, r"(a|b|c|d|e|f|g|h){8}"]


@pytest.mark.parametrize("pat", PATTERNS, ids=[repr(p)[:30] for p in PATTERNS])
def test_gpu_vs_oracle_regex(gpu_engine, oracle_engine, pat):
    s = fuzzdata.rows(11, 700, alphabet=list("aabbc xyz_.\n019") + ["é", "ü", "😀"]) + fuzzdata.log_rows(5, 700)
    s += ["a" * 80, "ab" * 50, "abcdefgh" * 3]
    o, g = oracle_engine, gpu_engine
    assert g.contains_re(s, pat) == o.contains_re(s, pat)
    assert g.match(s, pat) == o.match(s, pat)
    assert g.count_re(s, pat) == o.count_re(s, pat)
    for n in (-1, 1, 2):
        assert g.replace_re(s, pat, "<é>", n) == o.replace_re(s, pat, "<é>", n), n


def test_gpu_long_and_ragged_rows(gpu_engine, oracle_engine):
    s = ["", None, "x" * 5000 + " 1.2.3.4 " + "y" * 3000, " ".join(["tok"] * 700), "é" * 2000, None, "", "a b"]
    o, g = oracle_engine, gpu_engine
    assert g.lower(s) == o.lower(s)
    assert g.split(s, " ", -1) == o.split(s, " ", -1)
    assert g.split(s, None, 3) == o.split(s, None, 3)
    assert g.replace_re(s, IPV4, "<IP>", -1) == o.replace_re(s, IPV4, "<IP>", -1)
    assert g.tokenize(s, None) == o.tokenize(s, None)
    assert g.category(s) == o.category(s)


def test_gpu_empty_column(gpu_engine):
    g = gpu_engine
    assert g.lower([]) == [] and g.strip([]) == [] and g.replace_re([], "a", "b") == []
    assert g.tokenize([], None) == []
    assert [c for c in g.split([], " ")] == [[]]
    k, v = g.category([])
    assert k == [] and v == []


def test_gpu_arrow_roundtrip_int32():
    from custrings_amd import nvstrings

    gpuutil.lib()
    values = np.array([97, 112, 112, 108, 101, 112, 101, 97, 114], dtype=np.int8)
    offsets = np.array([0, 5, 5, 9], dtype=np.int32)
    bitmask = np.array([5], dtype=np.int8)
    s = nvstrings.from_offsets(values, offsets, 3, bitmask, 1)  # python/tests/test_offsets.py:36-47
    assert s.to_host() == ["apple", None, "pear"]
    s = nvstrings.from_offsets(values, offsets, 3)  # :18-24
    assert s.to_host() == ["apple", "", "pear"]
    s = nvstrings.to_device(["a", "p", "p", "l", "e"])  # :67-82
    v = np.empty(5, dtype=np.int8)
    o = np.empty(6, dtype=np.int32)
    n = np.empty(1, dtype=np.int8)
    s.to_offsets(v, o, n)
    assert v.tolist() == [97, 112, 112, 108, 101] and o.tolist() == [0, 1, 2, 3, 4, 5] and n.tolist() == [31]
    s = nvstrings.to_device(["a", None, "p", "l", "e"])
    nulls = np.zeros(1, dtype=np.uint8)
    assert s.set_null_bitmask(nulls) == 1 and nulls[0] == 0x1D  # nvstrings.py:611-616
    assert s.null_count() == 1
    assert nvstrings.to_device(["abc", "", None]).null_count(True) == 2
    lens = np.zeros(5, dtype=np.int32)
    assert s.byte_count(lens) == 4 and lens.tolist() == [1, -1, 1, 1, 1]


# ---- synthetic benchmark columns: generator and ops vs the oracle ----------------------
@pytest.mark.parametrize("kind,param", [(2, 0), (3, 0), (4, 1000), (4, 1 << 20), (5, 0)])
@pytest.mark.parametrize("first", [0, 99_000_000])
def test_gpu_synth_matches_spec(orc, kind, param, first):
    rows = 30_000
    gpuutil.assert_same(gpuutil.synth(kind, first, rows, param), orc.synth(kind, first, rows, param=param), "synth")


def test_gpu_vs_oracle_find_family(gpu_engine, oracle_engine):
    """rfind, find_from, find_multiple, compare, match_strings, startswith, endswith (find.cu:36-72, 123-236, 276-387)
    through the C ABI: values and the counts the reference returns; then the Python mirror's host lists."""
    from test_rowemu_parity import find_family_fuzz

    s = fuzzdata.rows(7, 2000, max_len=30) + ["", None, "a", "aa", "éé", "ab" * 20]
    find_family_fuzz(gpu_engine, oracle_engine, s)
    d = gpu_engine.col(["hello", "there", "world", "accéntéd", None, ""])
    assert d.rfind("d") == [-1, -1, 4, 7, None, -1] and d.find_from("r") == [-1, 3, 2, -1, None, -1]
    assert d.compare("there") == [-12, 0, 3, -19, None, -1]
    assert d.find_multiple(["e", "o", "d"]) == [[1, 4, -1], [2, -1, -1], [-1, 1, 4], [-1, -1, 7], [None, None, None], [-1, -1, -1]]
    assert d.startswith("he") == [True, False, False, False, None, False] and d.endswith("d") == [False, False, True, True, None, False]
    assert gpu_engine.col(["hello", "here", None, "accéntéd", None, ""]).match_strings(d) == [True, False, False, True, True, True]


def test_gpu_character_sets_of_any_size(gpu_engine, oracle_engine):
    """strip / tokenize / the NVText counters with character sets beyond 64 members (the reference walks a set of any
    length: custring_view.inl:93-105, text/tokens.cu:45-50)"""
    from test_rowemu_parity import big_sets

    s = fuzzdata.rows(5, 3000, max_len=40) + ["ΑΒΓ abc ωψχ", "жзи hello ЯЮЭ", "zzz", "", None, "ω", "Я" * 5 + "x" + "α" * 3, "~~~abc~~~"]
    g, o = gpu_engine, oracle_engine
    for ts in big_sets():
        for side in (0, 1, 2):
            assert g.strip(s, ts, side) == o.strip(s, ts, side), (len(ts), side)
        assert g.tokenize(s, ts) == o.tokenize(s, ts), len(ts)
        assert g.token_count(s, ts) == o.token_count(s, ts), len(ts)


def test_gpu_c2_lower_strip_split(orc):
    rows = 200_000
    g, o = gpuutil.synth(2, 0, rows), orc.synth(2, 0, rows)
    gl, ol = g.lower(), orc.lower(o)
    gpuutil.assert_same(gl, ol, "lower")
    gs, os_ = gl.strip(), orc.strip(ol)
    gpuutil.assert_same(gs, os_, "strip")
    gc, oc = gs.split(" "), orc.split(os_, " ")
    assert len(gc) == len(oc)
    for k, (a, b) in enumerate(zip(gc, oc)):
        gpuutil.assert_same(a, b, "split col %d" % k)
    gw, ow = g.split(None, 4), orc.split(o, None, 4)
    assert len(gw) == len(ow)
    for k, (a, b) in enumerate(zip(gw, ow)):
        gpuutil.assert_same(a, b, "wssplit col %d" % k)
    gpuutil.assert_same(g.upper(), orc.upper(o), "upper")
    f_g = np.zeros(rows, dtype=np.int32)
    L = gpuutil.lib()
    found = C.c_int64()
    L.check(L.lib.cs_find(g.m_cptr, "é".encode(), 0, -1, f_g.ctypes.data, 0, None, C.byref(found)))
    f_o, n_o = orc.find(o, "é", 0, -1)
    assert np.array_equal(f_g, f_o) and found.value == n_o


@pytest.mark.parametrize("pat", [IPV4, IPV4B])
@pytest.mark.parametrize("first", [0, 73_000_000])
def test_gpu_c3_regex_and_split(orc, pat, first):
    rows = 200_000
    g, o = gpuutil.synth(3, first, rows), orc.synth(3, first, rows)
    blob = np.ascontiguousarray(engines.reference_blob(pat))
    gpuutil.assert_same(g.replace(pat, "<IP>"), orc.replace_re(o, blob, "<IP>"), "replace_re")
    gpuutil.assert_same(g.replace(pat, "", 1), orc.replace_re(o, blob, "", 1), "replace_re n=1")
    re = gpuutil.compile_re(pat)
    got, n = gpuutil.bools(g, "cs_contains_re", re)
    exp, n_o = orc.contains_re(o, blob, 0)
    assert np.array_equal(got, exp) and n == n_o
    if pat == IPV4:
        gc, oc = g.split(" "), orc.split(o, " ")
        assert len(gc) == len(oc)
        for k, (a, b) in enumerate(zip(gc, oc)):
            gpuutil.assert_same(a, b, "split col %d" % k)
        gpuutil.assert_same(g.replace("0.", "#", regex=False), orc.replace(o, "0.", "#"), "replace")


@pytest.mark.parametrize("K", [1000, 1 << 20])
def test_gpu_c4_category(orc, K):
    from custrings_amd import nvcategory

    rows = 300_000
    g, o = gpuutil.synth(4, 0, rows, K), orc.synth(4, 0, rows, param=K)
    cat = nvcategory.from_strings(g)
    ok, ov = orc.category(o)
    gpuutil.assert_same(cat.keys(), ok, "keys")
    vals = np.zeros(rows, dtype=np.int32)
    cat.values(vals)
    assert np.array_equal(vals, ov)
    # merge of two shard categories == category of the concatenation (NVCategory.cu:430-514)
    g1, g2 = gpuutil.synth(4, 0, rows // 2, K), gpuutil.synth(4, rows // 2, rows - rows // 2, K)
    m = nvcategory.from_categories([nvcategory.from_strings(g1), nvcategory.from_strings(g2)])
    gpuutil.assert_same(m.keys(), ok, "merged keys")
    m.values(vals)
    assert np.array_equal(vals, ov)


def test_gpu_c5_tokenize_ngrams(orc):
    from custrings_amd import nvtext

    rows = 100_000
    g, o = gpuutil.synth(5, 0, rows), orc.synth(5, 0, rows)
    gt, ot = nvtext.tokenize(g), orc.tokenize(o)
    gpuutil.assert_same(gt, ot, "tokenize")
    gpuutil.assert_same(nvtext.ngrams(gt, 2, "_"), orc.ngrams(ot, 2, "_"), "bigrams")


# ---- full-size properties (BASELINE.json configs 2 and 3) -----------------------------------
def test_gpu_full_size_c2_properties():
    rows = 10_000_000
    g = gpuutil.synth(2, 0, rows)
    low = g.lower()
    assert low.lower().digest() == low.digest()  # idempotent
    st = low.strip()
    assert st.strip().digest() == st.digest()
    cols = st.split(" ")
    # every byte of a row is either in a token or one of the (tokens-1) delimiters
    tok_bytes = sum(int(gpuutil.lib().lib.cs_column_nbytes(c.m_cptr)) for c in cols)
    tokens = sum(rows - c.null_count() for c in cols)
    nonnull_rows = rows - cols[0].null_count()
    assert tok_bytes + (tokens - nonnull_rows) == int(gpuutil.lib().lib.cs_column_nbytes(st.m_cptr))
    assert g.null_count() == st.null_count() == cols[0].null_count()


def test_gpu_full_size_c3_properties(orc):
    rows = 100_000_000
    g = gpuutil.synth(3, 0, rows)
    L = gpuutil.lib()
    re = gpuutil.compile_re(IPV4)
    found = C.c_int64()
    flags = np.zeros(rows, dtype=np.uint8)
    L.check(L.lib.cs_contains_re(g.m_cptr, re, flags.ctypes.data, 0, None, C.byref(found)))
    n_hit = found.value
    assert int(flags.sum(dtype=np.int64)) == n_hit
    assert abs(n_hit / rows - 0.55) < 0.01  # 50 % one quad + 5 % two (cs_synth_spec.h)
    rep = g.replace(IPV4, "<IP>")
    L.check(L.lib.cs_contains_re(rep.m_cptr, re, flags.ctypes.data, 0, None, C.byref(found)))
    assert found.value == 0  # nothing left to replace
    # replacing again changes nothing (idempotence, checked by digest)
    assert rep.replace(IPV4, "<IP>").digest() == rep.digest()
    # the literal "<IP>" now appears once per replaced quad: 0.5 + 2*0.05 per row
    cnt = np.zeros(rows, dtype=np.int32)
    re2 = gpuutil.compile_re("<IP>")
    L.check(L.lib.cs_count_re(rep.m_cptr, re2, cnt.ctypes.data, 0, None, C.byref(found)))
    assert abs(int(cnt.sum(dtype=np.int64)) / rows - 0.60) < 0.01
    del flags, cnt
    # sampled windows of the full column agree with the oracle bit for bit
    blob = np.ascontiguousarray(engines.reference_blob(IPV4))
    for first in (0, 31_415_926, 99_950_000):
        w = 50_000
        o = orc.synth(3, first, w)
        exp = orc.replace_re(o, blob, "<IP>")
        chars, offs, valid = rep._export_window(first, w)
        got = cpulibs.Col(chars, offs, valid)
        assert got.same_as(exp), first
    # split: bytes are conserved (tokens + single-space delimiters)
    cols = g.split(" ")
    tok_bytes = sum(int(L.lib.cs_column_nbytes(c.m_cptr)) for c in cols)
    tokens = sum(rows - c.null_count() for c in cols)
    assert tok_bytes + (tokens - rows) == int(L.lib.cs_column_nbytes(g.m_cptr))


def test_gpu_full_size_c3_split_windows(orc):
    """split(' ') of the 100M-row headline column: sampled row windows of every output column equal
    the oracle's split of the same rows (columns the window's rows do not reach are all null)."""
    rows = 100_000_000
    g = gpuutil.synth(3, 0, rows)
    L = gpuutil.lib()
    cols = g.split(" ")
    assert all(int(L.lib.cs_column_offset_width(c.m_cptr)) == 4 for c in cols)  # each column is far below 2 GiB
    for first in (0, 27_182_818, 99_949_937):
        w = 50_000
        exp = orc.split(orc.synth(3, first, w), " ")
        assert len(exp) <= len(cols)
        for k, c in enumerate(cols):
            chars, offs, valid = c._export_window(first, w)
            got = cpulibs.Col(chars, offs, valid)
            if k < len(exp):
                assert got.same_as(exp[k]), (first, k)
            else:
                assert chars.size == 0 and not got.bitmask().any(), (first, k)
    # the same call with 64-bit offsets forced gives the same columns (digest covers offsets, chars, validity)
    L.check(L.lib.cs_config_set(b"CS_SPLIT_OFF64", b"1"))
    try:
        cols64 = g.split(" ")
    finally:
        L.check(L.lib.cs_config_set(b"CS_SPLIT_OFF64", None))
    assert all(int(L.lib.cs_column_offset_width(c.m_cptr)) == 8 for c in cols64)
    assert [c.digest() for c in cols64] == [c.digest() for c in cols]


def test_gpu_full_shard_c4_category(orc):
    """C4 at one GPU's shard of the 1B-row config (125M rows, K = 1M tokens): the key set equals the
    oracle's on a sample that covers it, keys are strictly ascending, and sampled windows of the values
    point at the rows' own strings."""
    from custrings_amd import nvcategory

    rows, K = 125_000_000, 1_000_000
    g = gpuutil.synth(4, 0, rows, K)
    cat = nvcategory.from_strings(g)
    assert cat.size() == rows
    keys = cat.keys()
    kchars, koffs, kvalid = keys._export64()
    nk = keys.size()
    assert keys.null_count() == 1 and not (kvalid[0] & 1)  # 0.1 % null rows: the null key sorts first
    kb = kchars.reshape(-1, 16)  # every non-null key is 16 bytes
    assert kb.shape[0] == nk - 1 and np.array_equal(koffs[1:], np.arange(nk) * 16)
    as_rows = kb.view(">u8")  # bytewise order == numeric order of the two big-endian halves
    order = np.lexsort((as_rows[:, 1], as_rows[:, 0]))
    assert np.array_equal(order, np.arange(nk - 1)), "keys not in ascending bytewise order"
    assert np.unique(as_rows, axis=0).shape[0] == nk - 1
    # a small-K column of the same generator: every key occurs in a 300k-row sample -> same key set as the oracle
    g2 = gpuutil.synth(4, 0, rows, 1000)
    ok, _ = orc.category(orc.synth(4, 0, 300_000, param=1000))
    cat2 = nvcategory.from_strings(g2)
    gpuutil.assert_same(cat2.keys(), ok, "K=1000 keys at 125M rows")
    # values: key[value[r]] is row r's string (null rows -> key 0)
    vals = np.zeros(rows, dtype=np.int32)
    cat.values(vals)
    assert vals.min() == 0 and vals.max() == nk - 1
    for first in (0, 62_500_001, 124_900_000):
        w = 100_000
        chars, offs, valid = g._export_window(first, w)
        isnull = np.unpackbits(valid, bitorder="little")[:w] == 0
        v = vals[first : first + w]
        assert np.array_equal(v == 0, isnull)
        rb = chars.reshape(-1, 16)
        assert np.array_equal(kb[v[~isnull] - 1], rb), first
    del vals


@pytest.mark.parametrize("K", [1 << 27, 1 << 40])
def test_gpu_category_with_keys_close_to_rows(orc, K, monkeypatch):
    """BASELINE.md section 3, C4 with K close to N (NVCategory.cu:246-304 sorts every distinct key): 60M rows of the
    log-uniform generator over 2^27 names (about a fifth of the rows are distinct keys) and over 2^40 names (nearly every
    row is its own key).  The distinct keys go through the radix sort on the 8-byte prefix with full compares among equal
    prefixes: keys strictly ascending bytewise (hence distinct), sampled windows of the values point at the rows' own
    strings, and on a prefix of the column the whole category equals the oracle's."""
    from custrings_amd import nvcategory

    rows = 60_000_000
    g = gpuutil.synth(4, 0, rows, K)
    cat = nvcategory.from_strings(g)
    keys = cat.keys()
    kchars, koffs, kvalid = keys._export64()
    nk = keys.size()
    assert nk > (8_000_000 if K == 1 << 27 else 20_000_000)
    assert keys.null_count() == 1 and not (kvalid[0] & 1)
    kb = kchars.reshape(-1, 16)
    assert kb.shape[0] == nk - 1
    hi, lo = kb[:, :8].copy().view(">u8").ravel(), kb[:, 8:].copy().view(">u8").ravel()
    assert np.all((hi[1:] > hi[:-1]) | ((hi[1:] == hi[:-1]) & (lo[1:] > lo[:-1]))), "keys not strictly ascending"
    vals = np.zeros(rows, dtype=np.int32)
    cat.values(vals)
    assert vals.min() == 0 and vals.max() == nk - 1
    for first in (0, 31_000_001, rows - 100_000):
        w = 100_000
        chars, offs, valid = g._export_window(first, w)
        isnull = np.unpackbits(valid, bitorder="little")[:w] == 0
        v = vals[first : first + w]
        assert np.array_equal(v == 0, isnull)
        assert np.array_equal(kb[v[~isnull] - 1], chars.reshape(-1, 16)), first
    del vals, cat, keys
    # the whole category of a 300k-row prefix against the oracle, through the same sort (forced: the key set is small)
    monkeypatch.setenv("CS_CAT_RADIX", "1")
    small = gpuutil.synth(4, 0, 300_000, K)
    ok, ov = orc.category(orc.synth(4, 0, 300_000, param=K))
    c2 = nvcategory.from_strings(small)
    gpuutil.assert_same(c2.keys(), ok, "keys")
    v2 = np.zeros(300_000, dtype=np.int32)
    c2.values(v2)
    assert np.array_equal(v2, ov)


def test_gpu_full_shard_c5_tokenize_ngrams(orc):
    """C5 at one GPU's shard of the 500M-row config (62.5M rows): token and byte conservation over
    the whole shard, sampled row windows of the flat token column and of its bigrams against the oracle."""
    from custrings_amd import nvtext

    rows = 62_500_000
    g = gpuutil.synth(5, 0, rows)
    L = gpuutil.lib()
    toks = nvtext.tokenize(g)
    ntok = toks.size()
    assert toks.null_count() == 0
    # a 64-bit-offset column: the shard's chars exceed 2 GiB
    assert int(L.lib.cs_column_nbytes(g.m_cptr)) > (1 << 31)
    bi = nvtext.ngrams(toks, 2, "_")
    assert bi.size() == ntok - 1
    # every bigram is token_i + '_' + token_i+1: bytes = 2 * token bytes - first - last + (ntok - 1)
    tb = int(L.lib.cs_column_nbytes(toks.m_cptr))
    first_len = toks.sublist(0, 1).byte_count()
    last_len = toks.sublist(ntok - 1, ntok).byte_count()
    assert int(L.lib.cs_column_nbytes(bi.m_cptr)) == 2 * tb - first_len - last_len + (ntok - 1)
    for first in (0, 31_250_000, rows - 40_000):
        w = 40_000
        before = nvtext.tokenize(g.sublist(0, first)).size() if first else 0
        o = orc.synth(5, first, w)
        ot = orc.tokenize(o)
        chars, offs, valid = toks._export_window(before, ot.rows)
        assert cpulibs.Col(chars, offs, valid).same_as(ot), first
        ob = orc.ngrams(ot, 2, "_")
        chars, offs, valid = bi._export_window(before, ob.rows)
        assert cpulibs.Col(chars, offs, valid).same_as(ob), first


def test_gpu_global_category_single_rank(orc):
    """dist.global_category with GpuOps on one rank == the plain category build."""
    from custrings_amd import dist as csd

    g, o = gpuutil.synth(4, 0, 50_000, 500), orc.synth(4, 0, 50_000, param=500)
    keys, values = csd.global_category(g)
    ok, ov = orc.category(o)
    gpuutil.assert_same(keys, ok, "keys")
    assert np.array_equal(values.cpu().numpy(), ov)
    # the merge path itself: two shards merged on one rank through GpuOps
    ops = csd.GpuOps()
    a, b = gpuutil.synth(4, 0, 25_000, 500), gpuutil.synth(4, 25_000, 25_000, 500)
    ca, (cha, ofa, na) = ops.category(a)
    cb, (chb, ofb, nb) = ops.category(b)
    mk, codes = ops.concat_category([ops.column(cha, ofa, na), ops.column(chb, ofb, nb)])
    gpuutil.assert_same(mk, ok, "merged keys")
    va = ops.remap(ca, codes[: ca.keys_size()].contiguous())
    vb = ops.remap(cb, codes[ca.keys_size() :].contiguous())
    assert np.array_equal(np.concatenate([va.cpu().numpy(), vb.cpu().numpy()]), ov)


def test_gpu_category_merge_gathered_c_abi(orc):
    """cs_category_merge_gathered (the merge step of the distributed build behind the C ABI): three shards of one column,
    each "rank" merges the gathered key sets and remaps its own codes -- merged keys and the concatenated codes equal the
    category of the whole column (oracle), for every rank; a shard that is all null and an empty shard take part."""
    from custrings_amd import dist as csd

    ops = csd.GpuOps()
    rows = 60_000
    ok, ov = orc.category(orc.synth(4, 0, rows, param=700))
    cuts = [0, 25_000, 25_000, 60_000]  # (the middle shard is empty)
    shards = [gpuutil.synth(4, cuts[i], cuts[i + 1] - cuts[i], 700) for i in range(3)]
    cats, keysets = [], []
    for sh in shards:
        cat, (ch, of, nn) = ops.category(sh)
        cats.append(cat)
        keysets.append(ops.column(ch, of, nn))
    got = []
    for rank in range(3):
        mk, vals = ops.merge_gathered(cats[rank], keysets, rank)
        gpuutil.assert_same(mk, ok, "merged keys on rank %d" % rank)
        got.append(vals.cpu().numpy())
    assert np.array_equal(np.concatenate(got), ov)


# ---- code paths of the persistent tile kernels (stream replace_re, emit2) -------------------
def _log_like(rnd, lo, hi, nonascii_every=0, idx=0):
    words = []
    total = 0
    target = rnd.randint(lo, hi)
    while total < target:
        k = rnd.random()
        if k < 0.15:
            w = ".".join(str(rnd.randint(0, 255)) for _ in range(rnd.choice([3, 4, 4, 4, 5])))
        elif k < 0.25:
            w = str(rnd.randint(0, 99999))
        elif k < 0.30:
            w = "x" * rnd.randint(17, 40)  # tokens longer than the 16-byte assembly path
        else:
            w = "".join(rnd.choice("abcdefgh/") for _ in range(rnd.randint(1, 9)))
        words.append(w)
        total += len(w) + 1
    s = " ".join(words)[:hi]
    if nonascii_every and idx % nonascii_every == 0:
        s = s[: len(s) // 2] + "é" + s[len(s) // 2 :]
    return s


@pytest.mark.parametrize("rows", [1, 63, 64, 65, 257, 4097, 20000])
def test_gpu_tile_kernels_row_counts(gpu_engine, oracle_engine, rows):
    import random

    rnd = random.Random(rows)
    s = [_log_like(rnd, 30, 90) for _ in range(rows)]
    for i in range(0, rows, 97):
        s[i] = None if i % 2 else ""
    o, g = oracle_engine, gpu_engine
    assert g.replace_re(s, IPV4, "<IP>", -1) == o.replace_re(s, IPV4, "<IP>", -1)
    assert g.split(s, " ", -1) == o.split(s, " ", -1)


@pytest.mark.parametrize("most", [33, 40, 64, 65])
def test_gpu_split_33_to_64_columns_on_the_tile_kernels(gpu_engine, oracle_engine, most):
    """Rows of up to 64 tokens (the C5 column splits into 38): the first-generation tile kernels, where lane k holds
    column k's destination; 65 and more take the generic kernels.  Short rows (second-generation measure pass first)
    and rows beyond the 96-byte masks."""
    import random
    from custrings_amd import _lib

    rnd = random.Random(most)
    o, g = oracle_engine, gpu_engine
    for wide in (False, True):
        s = []
        for i in range(3000):
            k = most if i % 50 == 7 else rnd.randint(1, most)
            s.append(" ".join("".join(rnd.choice("abcde") for _ in range(rnd.randint(0, 5 if wide else 1))) for _ in range(k)))
        s[5] = None
        s[6] = ""
        f0 = int(_lib.lib.cs_fallback_count())
        assert g.split(s, " ", -1) == o.split(s, " ", -1), (most, wide)
        assert g.split(s, " ", 40) == o.split(s, " ", 40), (most, wide)
        assert int(_lib.lib.cs_fallback_count()) == f0


@pytest.mark.parametrize("shape", ["fits", "nonascii", "long_rows", "wide_tiles"])
def test_gpu_tile_kernels_fallback_paths(gpu_engine, oracle_engine, shape):
    """fits: every sub-tile takes the lean scan / emit2; nonascii: some sub-tiles hold a
    non-ASCII row (generic scan inside the stream kernel); long_rows: rows beyond the 96-byte
    register masks; wide_tiles: 64-row spans beyond the prefetch registers (older kernels)."""
    import random

    rnd = random.Random(len(shape))
    if shape == "fits":
        s = [_log_like(rnd, 20, 93) for _ in range(6000)]
    elif shape == "nonascii":
        s = [_log_like(rnd, 20, 90, nonascii_every=150, idx=i) for i in range(6000)]
    elif shape == "long_rows":
        s = [_log_like(rnd, 20, 90) if i % 40 else _log_like(rnd, 100, 260) for i in range(6000)]
    else:
        s = [_log_like(rnd, 100, 300) for _ in range(3000)]
    o, g = oracle_engine, gpu_engine
    for pat, repl, n in ((IPV4, "<IP>", -1), (IPV4, "", 1), (IPV4B, "#", -1), (r"x*", "", -1), (r"\d+", "9", 2), (r"[a-c]+", "<long>", -1)):
        assert g.replace_re(s, pat, repl, n) == o.replace_re(s, pat, repl, n), (shape, pat, repl, n)
    for n in (-1, 1, 3):
        assert g.split(s, " ", n) == o.split(s, " ", n), (shape, n)
    assert g.split(s, ".", -1) == o.split(s, ".", -1)


@pytest.mark.parametrize("shape", ["short", "medium", "long", "huge"])
def test_gpu_tokenize_tile_kernels(gpu_engine, oracle_engine, shape):
    """Byte-parallel tokenize (cs_tokenize.hip): tiles of 64 / 32 / 16 rows by row length, the
    per-row fallback beyond that, whitespace and small ASCII delimiter sets, non-ASCII text,
    empty / null / all-delimiter rows, tokens across piece and tile boundaries."""
    import random

    rnd = random.Random(len(shape) * 7)
    lo, hi = {"short": (0, 30), "medium": (60, 150), "long": (200, 330), "huge": (500, 900)}[shape]
    words = ["a", "bc", "déf", "x" * 17, "12.5", "😀", "tab\tbed", "q" * 40]
    s = []
    for i in range(5000 if shape != "huge" else 600):
        if i % 53 == 0:
            s.append(None)
        elif i % 41 == 0:
            s.append("")
        elif i % 37 == 0:
            s.append(" \t  \n ")
        else:
            target = rnd.randint(lo, hi)
            t = rnd.choice(["", " ", "  "])
            while len(t) < target:
                t += rnd.choice(words) + rnd.choice([" ", "  ", "\n", "_", "-", " \t "])
            s.append(t[: max(target, 1)])
    o, g = oracle_engine, gpu_engine
    for d in (None, " ", "_-", " _-\n", "é ", "abcde"):
        assert g.tokenize(s, d) == o.tokenize(s, d), (shape, d)


@pytest.mark.parametrize("count", [3, 64, 65, 700, 9000])
def test_gpu_ngrams_tile_kernel(gpu_engine, oracle_engine, count):
    """cs_ngram.hip (no dropped rows: closed-form offsets) and the row-wise path (nulls / empties)."""
    import random

    rnd = random.Random(count)
    toks = ["".join(rnd.choice("abcdé😀xyz") for _ in range(rnd.choice([1, 2, 3, 5, 8, 15, 16, 17, 33]))) for _ in range(count)]
    o, g = oracle_engine, gpu_engine
    for N in (2, 3, 5):
        for sep in ("_", "", " - ", "12345678", "123456789"):
            assert g.ngrams(toks, N, sep) == o.ngrams(toks, N, sep), (count, N, sep)
    holes = list(toks)
    for i in range(0, count, 7):
        holes[i] = None if i % 2 else ""
    assert g.ngrams(holes, 2, "_") == o.ngrams(holes, 2, "_")


@pytest.mark.parametrize("rlen", [17, 21, 33, 64, 65, 200])
def test_gpu_long_replacements_on_the_stream_kernel(gpu_engine, oracle_engine, orc, rlen):
    """replace_re / replace with a replacement beyond the sixteen bytes the stream kernel keeps in registers: up to 64
    bytes it still takes the single-pass kernel (the text is read from memory at assembly), beyond that the two-pass
    kernels; either way the result is the oracle's -- rows with several matches, shrinking and growing patterns, a limit."""
    import random

    L = gpuutil.lib().lib
    rnd = random.Random(rlen)
    s = [_log_like(rnd, 20, 90) for _ in range(4000)]
    for i in range(0, len(s), 101):
        s[i] = None if i % 2 else ""
    repl = ("[REDACTED-IP-ADDRESS]" * 12)[:rlen]
    o, g = oracle_engine, gpu_engine
    before = int(L.cs_fallback_count())
    assert g.replace_re(s, IPV4, repl, -1) == o.replace_re(s, IPV4, repl, -1), rlen
    assert int(L.cs_fallback_count()) == before  # (single pass, whatever the replacement's length)
    # (one-byte matches with a replacement this long may outgrow what the single pass provisions: the host then repeats
    # with the two-pass kernels, counted -- the result is what matters here)
    for pat, n in ((r"\d+", -1), (r"[a-c]+", 2), (r"x{17,}", -1)):
        assert g.replace_re(s, pat, repl, n) == o.replace_re(s, pat, repl, n), (pat, n, rlen)
    assert g.replace(s, "cab", repl, -1) == o.replace(s, "cab", repl, -1)
    assert g.replace(s, ".", repl, 1) == o.replace(s, ".", repl, 1)
    before = int(L.cs_fallback_count())
    # the 100k-row C3 column (many sub-tiles, the persistent grid, the scanner wave)
    gc, oc = gpuutil.synth(3, 0, 100_000), orc.synth(3, 0, 100_000)
    blob = np.ascontiguousarray(engines.reference_blob(IPV4) if engines.reference_blob(IPV4) is not None else engines.product_blob(IPV4), dtype=np.int32)
    gpuutil.assert_same(gc.replace(IPV4, repl), orc.replace_re(oc, blob, repl), "C3 replace_re, %d-byte replacement" % rlen)
    assert int(L.cs_fallback_count()) == before


WIDE_PATTERNS = [r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}", r"\w{5}", r"[a-z]{3,8}@", r"[0-9a-f]{8}-[0-9a-f]{4}", r"(a|b|c){6}x", r"\w{5,7} ", r"[ab]{2,6}c|\d{5}"]


def test_gpu_unit_route_on_tiles_with_non_ascii_bytes(gpu_engine, oracle_engine):
    """Sub-tiles that hold bytes >= 0x80 and a pattern such bytes can only kill (ASCII classes and literals, no anchors, no
    \\b: regex_tdfa.cpp header word 31 bit 17): the unit route takes them, a unit's scan ending at the unit's end
    (cs_regex.hip: reclassify_high) -- replace_re / count_re / findall against the oracle, with the characters next to the
    matches, between units and at the row ends; \\d patterns (non-ASCII digits exist) keep the generic scan and must agree too."""
    import random

    rnd = random.Random(23)
    pieces = ["1.2.3.4", "10.20.30.40 ", "é", "ü", "€", "😀", " ", ".", "12", "abc", "a@b", "GET /x ", "7-8", "abc.com ", "@", "-", "b", "0", "٣"]
    s = []
    for _ in range(6000):
        row = "".join(rnd.choice(pieces) for _ in range(rnd.randint(0, 16)))
        s.append(row if len(row.encode()) <= 90 else row[:28])
    s += ["é1.2.3.4é", "1.2.3.4é5.6.7.8", "é", "éé1.2.3", "1.2.3.é4", "aaé", "ébé", "😀ab😀", "", None]
    o, g = oracle_engine, gpu_engine
    for pat in (r"[0-9]+\.[0-9]+\.[0-9]+\.[0-9]+", r"[0-9]+", r"[a-c]+", r"[a-c]+@[a-c]+", r"[a-z]+\.com", r"\d+\.\d+\.\d+\.\d+", r"\d+"):
        for repl in ("<IP>", "", "a-longer-one"):
            assert g.replace_re(s, pat, repl, -1) == o.replace_re(s, pat, repl, -1), (pat, repl)
        assert g.count_re(s, pat) == o.count_re(s, pat), pat
        assert g.findall(s, pat) == o.findall(s, pat), pat
        assert g.contains_re(s, pat) == o.contains_re(s, pat), pat


def _with_outliers(rnd, rows=6000):
    """short log-like rows with a few very long ones among them (one ASCII, one with two-byte characters, one at the end)"""
    s = [_log_like(rnd, 20, 90) for _ in range(rows)]
    s[rows // 3] = ("GET /Index.HTML 10.1.2.3 Mixed CASE words " * 3000)[:100_000]
    s[rows // 2] = ("Ünïcödé ÀÉÎ straße 192.168.0.1 " * 400)
    s[rows // 2 + 1] = None
    s[-1] = "tail row 8.8.8.8 " * 700
    return s


def test_gpu_one_long_row_among_short_ones_stays_on_the_tile_kernels(gpu_engine, oracle_engine):
    """A column whose largest 64-row tile does not fit the staging buffer because of a FEW long rows: the tile kernels
    take it and handle the oversize tiles a thread per row themselves (cs_internal.h: few_spans64_over) -- lower / upper
    against the oracle."""
    import random

    from custrings_amd import _lib

    s = _with_outliers(random.Random(5))
    o, g = oracle_engine, gpu_engine
    assert g.lower(s) == o.lower(s)
    assert g.upper(s) == o.upper(s)
    # replace_re: the stream kernel sizes and writes an oversize sub-tile's rows a thread each, inside the prefix chain
    f0 = int(_lib.lib.cs_fallback_count())
    for pat, repl, n in ((IPV4, "<IP>", -1), (IPV4, "", -1), (IPV4, "[a longer replacement]", -1), (r"\d+", "#", 2), (r"[A-Z]+", "x", -1), (r"x*", "-", -1),
                         (r"\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}", "<IP>", -1)):
        assert g.replace_re(s, pat, repl, n) == o.replace_re(s, pat, repl, n), (pat, repl, n)
    assert g.replace(s, "GET", "PUT") == o.replace(s, "GET", "PUT")
    assert int(_lib.lib.cs_fallback_count()) == f0
    for chars in (None, " GETtailrow8.", "Ü"):
        assert g.strip(s, chars) == o.strip(s, chars), chars
    # rows of hundreds of bytes throughout (no tile size fits any tile): the same paths for every tile
    wide = [_log_like(rnd2, 300, 600, nonascii_every=7, idx=i) for i, rnd2 in ((i, random.Random(i)) for i in range(700))] + [None, "", "Ü" * 300]
    assert g.lower(wide) == o.lower(wide) and g.upper(wide) == o.upper(wide)
    assert g.strip(wide, None) == o.strip(wide, None)
    assert g.tokenize(wide) == o.tokenize(wide)
    # split: the first-generation tile kernels on sub-tiles of 32 / 16 / 8 rows
    for col in (wide, [_log_like(random.Random(2000 + i), 900, 1300) for i in range(300)] + [None, ""]):
        for delim, n in ((" ", 8), (" ", 3), (".", -1), ("/", 5)):
            assert g.split(col, delim, n) == o.split(col, delim, n), (len(col), delim, n)
    # replace_re: tiles of eight / four rows through the stream kernel (rows beyond the sliding window scan generically)
    f1 = int(_lib.lib.cs_fallback_count())
    huge = [_log_like(random.Random(1000 + i), 900, 1300) for i in range(300)]
    for col in (wide, huge):
        for pat, repl in ((IPV4, "<IP>"), (r"[a-c]+", ""), (r"\d+", "<number>")):
            assert g.replace_re(col, pat, repl, -1) == o.replace_re(col, pat, repl, -1), (len(col), pat)
    assert int(_lib.lib.cs_fallback_count()) == f1
    # tokenize: an oversize tile is walked in segments of the staging size, state carried across (whitespace, a delimiter set)
    assert g.tokenize(s) == o.tokenize(s)
    assert g.tokenize(s, " /.") == o.tokenize(s, " /.")
    long_only = [("word%d " % i) * 3000 for i in range(70)] + [None, "", "x"]  # every tile oversize
    assert g.tokenize(long_only) == o.tokenize(long_only)
    # split: the first-generation tile kernels read an oversize sub-tile.s rows from memory and write its tokens straight to the columns
    for delim, n in ((" ", 5), (" ", 1), ("#", -1), (".", 3), (None, 4)):
        assert g.split(s, delim, n) == o.split(s, delim, n), (delim, n)
    assert g.rsplit(s, " ", 3) == o.rsplit(s, " ", 3)
    few = list(s)
    few[len(few) // 3] = "one-token-row" * 1000  # the long rows hold fewer tokens than the short ones
    few[len(few) // 2] = None
    few[-1] = ""
    assert g.split(few, " ", -1) == o.split(few, " ", -1)


@pytest.mark.parametrize("kind", ["url", "nested", "nul", "long", "dups"])
def test_gpu_category_keys_that_share_long_prefixes(gpu_engine, oracle_engine, kind):
    """Keys that tie on the sort's 8-byte prefix (URL-like columns: every key): the tied records are ordered by rounds of
    radix sorts on the next seven key bytes, group by group (cs_category.hip) -- keys that are prefixes of other keys,
    keys with NUL bytes (zero padding must not tie them with their extensions), prefixes far longer than a round, a
    column with few ties (the compare network in LDS) -- keys and codes against the oracle; sort / order ride on it."""
    import random

    rnd = random.Random(len(kind))
    n = 40000
    if kind == "url":
        s = ["https://example.com/%s/%d" % (rnd.choice(["a", "api/v1", "static/img"]), rnd.randrange(30000)) for _ in range(n)]
    elif kind == "nested":  # every key a prefix of the next ones
        s = ["prefix-prefix-" + "x" * rnd.randrange(60) for _ in range(n)] + ["prefix-prefix-" + "x" * i + "y" for i in range(40)]
    elif kind == "nul":
        alphabet = ["a", "b", "\0", "\0\0", "ab", ""]
        s = ["".join(rnd.choice(alphabet) for _ in range(rnd.randrange(14))) for _ in range(n)]
    elif kind == "long":
        s = ["z" * 200 + "%05d" % rnd.randrange(20000) + "q" * rnd.randrange(3) for _ in range(n)]
    else:  # few ties
        s = ["%08x" % rnd.randrange(1 << 30) for _ in range(n)] + ["samesame-%d" % (i % 50) for i in range(500)]
    s[17] = None
    s[18] = ""
    o, g = oracle_engine, gpu_engine
    assert g.category(s) == o.category(s)


@pytest.mark.parametrize("pat,repl", [("x*", "-"), ("a*", "<>"), ("\\d*", "#"), ("[a-c]*", "."), ("b*|c", "_"), ("\\b", "|"), ("$", "!"), ("^", ">"),
                                      ("x*", "<IP>"), ("\\d*", "<number>")])
def test_gpu_patterns_that_match_the_empty_string_on_the_stream_kernel(gpu_engine, oracle_engine, pat, repl):
    """replace_re with a pattern that matches the empty string and a replacement of one or two bytes (replace.cu:91-93,
    the zero-length repeat rule): bounded growth -- one replacement per character and one at the
    row's end -- so the single-pass stream kernel takes it (the out tile and the output sized for exactly that; up to
    eight bytes), with limits too; no fallback."""
    import random
    from custrings_amd import _lib

    rnd = random.Random(len(pat) * 31 + len(repl))
    s = [_log_like(rnd, 0, 93) for _ in range(5000)] + ["", None, "x", "xx", "axxb", "aaa", "abcabc", "é", "xéx", "12", " ", "\n", "a\nb"]
    o, g = oracle_engine, gpu_engine
    f0 = int(_lib.lib.cs_fallback_count())
    for n in (-1, 1, 3):
        assert g.replace_re(s, pat, repl, n) == o.replace_re(s, pat, repl, n), (pat, repl, n)
    long_rows = [_log_like(rnd, 100, 260) for _ in range(1500)]
    assert g.replace_re(long_rows, pat, repl, -1) == o.replace_re(long_rows, pat, repl, -1)
    assert int(_lib.lib.cs_fallback_count()) == f0


@pytest.mark.parametrize("rows", [1, 64, 65, 4097, 20000])
def test_gpu_findall_extract_from_packed_spans(gpu_engine, oracle_engine, rows, monkeypatch):
    """findall / extract on 64-row tiles: the scan stream kernel leaves one packed word per (column, row) and every tile's
    bytes per column, k_spans_write_tile2 turns lengths into offsets itself (no pass over the lengths).  Nulls, empty
    rows, rows without a match, more than four matches in a row (the exact-width second pass writes begins / lens),
    more than four capture groups (columns beyond the write kernel's prefetched four), the last partial tile; the
    unpacked route must agree."""
    import random

    rnd = random.Random(rows)
    s = [_log_like(rnd, 20, 90) for _ in range(rows)]
    for i in range(0, rows, 53):
        s[i] = None if i % 2 else ""
    if rows > 100:
        s[77] = "1.2.3.4 5.6.7.8 9.9.9.9 10.0.0.1 8.8.8.8 7.7.7.7"  # six matches in one row
    o, g = oracle_engine, gpu_engine
    pats_f = [IPV4, r"\d+", r"[a-c]+", r"/\S*"]
    pats_e = [r"(\d+)\.(\d+)\.\d+\.(\d+) ", r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"(\w)(\w)(\w)(\w)(\w)(\w)", r"(GET|POST) (/\S*)", r"(a)|(b)"]
    for pat in pats_f:
        want = o.findall(s, pat)
        assert g.findall(s, pat) == want, pat
    for pat in pats_e:
        want = o.extract(s, pat)
        assert g.extract(s, pat) == want, pat
    monkeypatch.setenv("CS_SPANS_UNPACKED", "1")
    assert g.findall(s, IPV4) == o.findall(s, IPV4)
    assert g.extract(s, pats_e[0]) == o.extract(s, pats_e[0])


@pytest.mark.parametrize("pat", WIDE_PATTERNS)
def test_gpu_patterns_with_five_to_eight_threads(gpu_engine, oracle_engine, orc, pat):
    """Counted repetitions keep five to eight threads alive: such programs run the tagged DFA with eight start offsets
    (regex_tdfa.h TdfaWide; cs_regex_engine reports the thread count) in contains_re / match / count_re (tile stream kernel)
    and replace_re (the stream kernel's WIDE forms for replacements of up to 16 bytes, two-pass beyond), the list simulator
    elsewhere -- all against the oracle."""
    import random

    from custrings_amd import nvstrings

    L = gpuutil.lib().lib
    re = nvstrings._compile(pat)
    e = int(L.cs_regex_engine(re))
    L.cs_regex_destroy(re)
    assert e & 1 and 5 <= ((e >> 8) & 15) <= 8, hex(e)
    rnd = random.Random(len(pat))
    s = [_log_like(rnd, 10, 120) for _ in range(3000)] + ["1.2.3.4", "1234.1.2.3", "1.22.333.4444x", "abcdefgh@", "deadbeef-cafe", "aaaaaax", "12345 ", "", None, "é1.2.3.4é"]
    o, g = oracle_engine, gpu_engine
    assert g.contains_re(s, pat) == o.contains_re(s, pat)
    assert g.match(s, pat) == o.match(s, pat)
    assert g.count_re(s, pat) == o.count_re(s, pat)
    from custrings_amd import _lib

    f0 = int(_lib.lib.cs_fallback_count())
    for repl, n in (("<>", -1), ("", -1), ("#", 2), ("<longer>", -1), ("[thirteen-b.]", -1), ("[thirteen-b.]", 1), ("a-replacement-of-more-than-16-bytes", -1)):
        assert g.replace_re(s, pat, repl, n) == o.replace_re(s, pat, repl, n), (repl, n)
    # (rows within the 96-byte masks: the stream kernel's WIDE forms without the sliding window)
    short = [_log_like(rnd, 10, 90) for _ in range(3000)] + ["1.2.3.4", "", None]
    for repl in ("<>", "<longer>", "[thirteen-b.]"):
        assert g.replace_re(short, pat, repl, -1) == o.replace_re(short, pat, repl, -1), repl
    assert int(_lib.lib.cs_fallback_count()) == f0
    assert g.findall(s, pat) == o.findall(s, pat)
    # a C3 window through the persistent grid
    gc, oc = gpuutil.synth(3, 0, 100_000), orc.synth(3, 0, 100_000)
    blob = np.ascontiguousarray(engines.reference_blob(pat) if engines.reference_blob(pat) is not None else engines.product_blob(pat), dtype=np.int32)
    want, nwant = orc.contains_re(oc, blob)
    got, ngot = gpuutil.bools(gc, "cs_contains_re", gpuutil.compile_re(pat))
    assert np.array_equal(got, want) and ngot == nwant
    gpuutil.assert_same(gc.replace(pat, "<IP>"), orc.replace_re(oc, blob, "<IP>"), "C3 replace_re")


def test_gpu_literal_replace_on_stream_kernel(gpu_engine, oracle_engine, orc):
    """Literal needles without metacharacters and a replacement no longer than the needle take
    the single-pass replace_re kernel; results must equal the literal replace of the oracle."""
    s = fuzzdata.rows(3, 3000, max_len=60) + fuzzdata.log_rows(9, 2000) + ["aaaa", "aaa", "abab ab", "", None]
    o, g = oracle_engine, gpu_engine
    for pat, repl in (("a", "b"), ("aa", "a"), ("ab", ""), ("b c", "_"), ("1", "#"), ("abc", "abc"), ("a", "")):
        for n in (-1, 0, 1, 2):
            assert g.replace(s, pat, repl, n) == o.replace(s, pat, repl, n), (pat, repl, n)
    rows = 100_000
    gc, oc = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    for pat, repl in (("GET", "G"), ("POST ", "P"), ("200", "OK"), ("e", "")):
        gpuutil.assert_same(gc.replace(pat, repl, regex=False), orc.replace(oc, pat, repl), "literal %r" % pat)


def test_gpu_growing_replace_with_many_matches(gpu_engine, oracle_engine, orc):
    """A replacement longer than the match, on rows with more matches than the single-pass kernel
    keeps in registers: the roomier launch rescans such rows while it assembles them."""
    s = fuzzdata.rows(4, 3000, max_len=70) + fuzzdata.log_rows(10, 2000) + ["aaaaaaaaaaaaaaaaaaaaaaaa", "a" * 90, "", None, "é" * 20 + "a"]
    o, g = oracle_engine, gpu_engine
    for pat, repl in (("a", "xx"), ("a", "xyz12"), (" ", "    "), ("ab", "abab"), ("1", "0123456789abcdef")):
        for n in (-1, 1, 5, 7):
            assert g.replace(s, pat, repl, n) == o.replace(s, pat, repl, n), (pat, repl, n)
    for pat, repl in ((r"\d", "<d>"), (r"[aeiou]", "<v>"), (r"\s", "__"), (r"\w+", "<word>"), (r"\d+", "<number-here>"), (r"[a-c]", "é")):
        for n in (-1, 6):
            assert g.replace_re(s, pat, repl, n) == o.replace_re(s, pat, repl, n), (pat, repl, n)
    rows = 100_000
    gc, oc = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    gpuutil.assert_same(gc.replace(" ", "  ", regex=False), orc.replace(oc, " ", "  "), "literal ' ' -> '  '")
    for pat, repl in ((r"\d", "##"), (r"[aeiou]", "<v>")):
        blob = np.ascontiguousarray(engines.reference_blob(pat))
        gpuutil.assert_same(gc.replace(pat, repl), orc.replace_re(oc, blob, repl), "%s -> %s" % (pat, repl))


def test_gpu_literal_replace_routing_on_odd_bytes(monkeypatch):
    """A literal needle is routed through the regex stream kernel only when a per-character scan
    and a per-byte scan agree on the column's bytes (no NUL, no lead byte announcing over an ASCII
    byte).  Columns built from valid sequences, stray continuation bytes and rows cut in the
    middle of a sequence qualify; a lead byte in front of ASCII does not.  Either way the result
    must equal the row-wise literal kernels."""
    rng = np.random.default_rng(21)
    tokens = [bytes([c]) for c in b"abab 12.x"] + ["é".encode(), "€".encode(), "😀".encode(), b"\x80", b"\xbf", b"\xf0\xe2\x80\x80"]
    for extra in ([], [b"\xe2"], [b"\x00"]):
        pool = tokens + extra
        data = b"".join(pool[i] for i in rng.integers(0, len(pool), 400_000))
        chars = np.frombuffer(data, dtype=np.uint8)
        rows = 12_000
        cuts = np.sort(rng.integers(0, len(chars), rows - 1))
        offs = np.concatenate([[0], cuts, [len(chars)]]).astype(np.int64)
        col = cpulibs.Col(chars, offs, None)
        g = gpuutil.from_col(col)
        fast = [gpuutil.to_col(g.replace(p, r, regex=False)) for p, r in (("ab", "x"), ("a", "xx"), (" ", ""), ("b", "b"))]
        monkeypatch.setenv("CS_REPLACE_ROWWISE", "1")
        slow = [gpuutil.to_col(g.replace(p, r, regex=False)) for p, r in (("ab", "x"), ("a", "xx"), (" ", ""), ("b", "b"))]
        monkeypatch.delenv("CS_REPLACE_ROWWISE")
        assert all(a.same_as(b) for a, b in zip(fast, slow)), extra


def test_gpu_category_long_keys_at_every_alignment(gpu_engine, oracle_engine):
    """Keys longer than the 32 bytes the probe keeps in registers, equal up to their last bytes
    or up to byte 31 / 32 / 33 / 63 / 64, of every length around the 16- and 32-byte piece
    boundaries, starting at every byte alignment (a filler row of 0..16 bytes in front)."""
    import random

    rnd = random.Random(8)
    stem = "".join(rnd.choice("abcdefgh") for _ in range(300))
    keys = []
    for n in list(range(0, 70)) + [95, 96, 97, 127, 128, 129, 200, 300]:
        keys.append(stem[:n])
        for cut in (0, 15, 16, 17, 31, 32, 33, 63, 64, n - 1):
            if 0 <= cut < n:
                keys.append(stem[:cut] + "Z" + stem[cut + 1:n])
    keys += ["é" * 17, "é" * 16 + "e", "", None]
    s = []
    for _ in range(6000):
        s.append("f" * rnd.randint(0, 16))  # shifts the alignment of the next row
        s.append(rnd.choice(keys))
    assert gpu_engine.category(s) == oracle_engine.category(s)


@pytest.mark.parametrize("shift", [0, 1, 7, 13])
def test_gpu_zero_copy_column_at_odd_address(orc, shift):
    """A borrowed (zero-copy) column whose chars start at an arbitrary byte address inside the
    caller's allocation: every tile kernel derives its 16-byte pieces from the absolute address,
    so the results must equal those of the copied column."""
    import torch
    from custrings_amd import nvstrings, nvtext, nvcategory

    rows = 50_000
    o = orc.synth(3, 0, rows)
    dev = torch.device("cuda:0")
    backing = torch.zeros(len(o.chars) + 64, dtype=torch.uint8, device=dev)
    backing[shift:shift + len(o.chars)] = torch.from_numpy(o.chars).to(dev)
    offs = torch.from_numpy(o.offsets.astype(np.int64)).to(dev)
    chars = backing[shift:shift + len(o.chars)]
    assert chars.data_ptr() % 16 == (backing.data_ptr() + shift) % 16
    g = nvstrings.from_offsets64(chars, offs, rows, None, bdevmem=True, copy=False)
    ref = gpuutil.from_col(o)

    def same(a, b, what):
        assert gpuutil.to_col(a).same_as(gpuutil.to_col(b)), what

    same(g.replace(IPV4, "<IP>"), ref.replace(IPV4, "<IP>"), "replace_re")
    same(g.replace(IPV4, "<redacted-ip>"), ref.replace(IPV4, "<redacted-ip>"), "growing replace_re")
    same(g.replace("e", "EE", regex=False), ref.replace("e", "EE", regex=False), "literal replace")
    for a, b in zip(g.split(" "), ref.split(" ")):
        same(a, b, "split")
    same(g.upper(), ref.upper(), "upper")
    same(g.strip("GETPOS "), ref.strip("GETPOS "), "strip")
    same(nvtext.tokenize(g), nvtext.tokenize(ref), "tokenize")
    same(nvtext.ngrams(nvtext.tokenize(g), 2, "_"), nvtext.ngrams(nvtext.tokenize(ref), 2, "_"), "ngrams")
    assert g.contains(IPV4) == ref.contains(IPV4)
    assert g.find("200") == ref.find("200")
    ca, cb = nvcategory.from_strings(g), nvcategory.from_strings(ref)
    same(ca.keys(), cb.keys(), "category keys")
    va, vb = np.zeros(rows, dtype=np.int32), np.zeros(rows, dtype=np.int32)
    ca.values(va), cb.values(vb)
    assert np.array_equal(va, vb)


@pytest.mark.parametrize("base", [0, 5])
@pytest.mark.parametrize("nulls", ["none", "empty", "holding_bytes"])
@pytest.mark.parametrize("width", [32, 64])
def test_gpu_ingest_makes_columns_canonical(base, nulls, width):
    """create_from_offsets (NVStringsImpl.cu:399-444): offsets may start above zero and Arrow allows
    bytes under a null row; the native column always has offsets[0] == 0 and empty null rows.  The
    already-canonical case keeps the ingested buffers, the others are re-packed: both must export
    the same rows."""
    import torch
    from custrings_amd import nvstrings

    rng = np.random.default_rng(3)
    rows = 5000
    lens = rng.integers(0, 40, rows)
    valid_bits = np.ones(rows, dtype=bool) if nulls == "none" else rng.random(rows) > 0.2
    if nulls == "empty":
        lens[~valid_bits] = 0
    offs = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    chars = rng.integers(97, 123, int(offs[-1]) + base, dtype=np.uint8)
    offs += base
    mask = None if nulls == "none" else np.packbits(valid_bits, bitorder="little")
    want = [chars[offs[i]:offs[i + 1]].tobytes().decode() if valid_bits[i] else None for i in range(rows)]
    dev = torch.device("cuda:0")
    for on_device in (False, True):
        c = torch.from_numpy(chars).to(dev) if on_device else chars
        o = offs.astype(np.int32) if width == 32 else offs
        o = torch.from_numpy(o).to(dev) if on_device else o
        m = None if mask is None else (torch.from_numpy(mask).to(dev) if on_device else mask)
        if width == 32:
            g = nvstrings.from_offsets(c, o, rows, m, 0, bdevmem=on_device)
        else:
            g = nvstrings.from_offsets64(c, o, rows, m, bdevmem=on_device)
        assert g.to_host() == want, (on_device,)
        ch, of, va = g._export64()
        assert of[0] == 0 and of[-1] == sum(len(w) for w in want if w is not None)
        assert g.byte_count() == of[-1]
        lens_out = np.zeros(rows, dtype=np.int32)
        g.byte_count(lens_out)
        assert lens_out.tolist() == [len(w) if w is not None else -1 for w in want]


@pytest.mark.parametrize("long_rows", [False, True])
def test_gpu_whitespace_split_tile_kernels(gpu_engine, oracle_engine, long_rows):
    """split(None, n) (split.cu:863-956) on the tile kernels: runs of spaces, tabs, newlines and
    other control bytes separate, non-ASCII bytes never do, the token that exhausts maxsplit keeps
    the rest of the row; with a row beyond the 96-bit masks the column takes the generic kernels."""
    import random

    rnd = random.Random(17)
    alphabet = list("abcXYZ09_.") + [" ", " ", "  ", "\t", "\n", "\x01", "\x1f", "é", "\u00a0", "\u2003", "😀"]
    s = []
    for _ in range(4000):
        s.append("".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 30))))
    s += ["", " ", "   ", "a", " a", "a ", " a b ", "\ta\nb  c", "a  b   c d", None, "é é"]
    if long_rows:
        s += ["w " * 80, "x" * 200]
    o, g = oracle_engine, gpu_engine
    for n in (-1, 0, 1, 2, 3, 7):
        assert g.split(s, None, n) == o.split(s, None, n), n


def test_gpu_find_windows_and_escaped_literal_replace(orc):
    """find(sub, start, end) on ASCII tiles takes the candidate bitmap with the window applied as a
    byte range; a literal needle made of regex metacharacters rides the stream kernel escaped."""
    rows = 100_000
    g, o = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    L = gpuutil.lib()
    for sub in ("200", " ", "GET /", "e"):
        for st, en in ((0, -1), (10, 60), (5, 5), (40, 20), (-3, 30), (70, 200), (0, 1), (47, 49)):
            f = np.zeros(rows, dtype=np.int32)
            found = C.c_int64()
            L.check(L.lib.cs_find(g.m_cptr, sub.encode(), st, en, f.ctypes.data, 0, None, C.byref(found)))
            f_o, n_o = orc.find(o, sub, st, en)
            assert np.array_equal(f, f_o) and found.value == n_o, (sub, st, en)
    for pat, repl in ((".", "_"), ("/", "//"), (". ", ""), ("(", "["), ("a.b", "x"), ("\\", "/"), ("$", "USD"), ("[", "]"), ("*", "")):
        gpuutil.assert_same(g.replace(pat, repl, regex=False), orc.replace(o, pat, repl), "literal %r" % pat)
    s = ["a.b.c", "..", "x*y+z?", "(a|b)", "^$", "[x]{2}", "back\\slash", "", None, "3.14"]
    from custrings_amd import nvstrings

    d = nvstrings.to_device(s)
    for pat, repl in ((".", "-"), ("*", "S"), ("+", "P"), ("?", "Q"), ("(", "<"), (")", ">"), ("|", "I"), ("^", "C"), ("$", "D"),
                      ("[", "L"), ("]", "R"), ("{", "B"), ("}", "E"), ("\\", "/"), ("a.b", "AB"), ("..", ":")):
        want = [None if x is None else x.replace(pat, repl) for x in s]
        assert d.replace(pat, repl, regex=False).to_host() == want, (pat, repl)


def test_gpu_multibyte_delimiter_split_tile_kernels(gpu_engine, oracle_engine, orc):
    """split on a delimiter of 2..8 ASCII bytes: occurrences are taken left to right without
    overlap (custring_view.inl:1223-1279), also when the delimiter overlaps itself."""
    import random

    rnd = random.Random(23)
    alphabet = list("aab.,;  x") + ["é", "ab", ", ", "aa"]
    s = ["".join(rnd.choice(alphabet) for _ in range(rnd.randint(0, 28))) for _ in range(4000)]
    s += ["", "aa", "aaa", "aaaa", "aaaaa", ", ", "a, ", ", a", "a, , b", None, "ab" * 30, ", " * 20 + "x"]
    o, g = oracle_engine, gpu_engine
    for d in ("aa", ", ", "ab", "a, ", ". ", "aaa", "b.,;", ";  x", "abababab"):
        for n in (-1, 1, 2, 5):
            assert g.split(s, d, n) == o.split(s, d, n), (d, n)
    rows = 100_000
    gc, oc = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    for d in (". ", " /", "00 "):
        got, want = gc.split(d), orc.split(oc, d)
        assert len(got) == len(want), d
        for a, b in zip(got, want):
            gpuutil.assert_same(a, b, "split %r" % d)


def test_gpu_category_table_growth(orc, monkeypatch):
    """The category build starts with a small hash table and retries with a larger one when a
    probe run gets long: force the retries with a tiny first table."""
    from custrings_amd import nvcategory

    rows = 200_000
    g, o = gpuutil.synth(4, 0, rows, 1 << 20), orc.synth(4, 0, rows, param=1 << 20)
    ok, ov = orc.category(o)
    for first in ("6", "12", "30"):
        monkeypatch.setenv("CS_CAT_FIRST_LOG2", first)
        cat = nvcategory.from_strings(g)
        gpuutil.assert_same(cat.keys(), ok, "keys (first table 2^%s)" % first)
        vals = np.zeros(rows, dtype=np.int32)
        cat.values(vals)
        assert np.array_equal(vals, ov)


def test_gpu_case_tile_kernel_on_arbitrary_bytes(monkeypatch):
    """lower()/upper() tile kernel against the row-wise kernels on rows of ARBITRARY bytes
    (stray / missing continuation bytes, truncated sequences at row ends): the fast path must
    hand every malformed row to the sequential routine, so both routes agree bit for bit."""
    rng = np.random.default_rng(5)
    rows = 20_000
    lens = rng.integers(0, 90, rows)
    offs = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    pool = np.array(list(b"abcXYZ 09.") + [0xC3, 0xA9, 0x89, 0xE2, 0x82, 0xAC, 0xF0, 0x9F, 0x98, 0x80, 0xC4, 0xB0, 0xFF, 0x80], dtype=np.uint8)
    chars = pool[rng.integers(0, len(pool), int(offs[-1]))]
    col = cpulibs.Col(chars, offs, None)
    g = gpuutil.from_col(col)
    fast_l, fast_u = gpuutil.to_col(g.lower()), gpuutil.to_col(g.upper())
    monkeypatch.setenv("CS_CASE_ROWWISE", "1")
    slow_l, slow_u = gpuutil.to_col(g.lower()), gpuutil.to_col(g.upper())
    assert fast_l.same_as(slow_l) and fast_u.same_as(slow_u)


def test_gpu_tile_kernels_on_arbitrary_bytes(monkeypatch):
    """Tile kernels against the row-wise kernels on rows of ARBITRARY bytes (NUL, invalid UTF-8,
    control characters): every fast path must either handle such bytes or hand the tile to the
    sequential code, so the two routes agree bit for bit."""
    rng = np.random.default_rng(11)
    rows = 30_000
    lens = rng.integers(0, 95, rows)
    offs = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    pool = np.array(list(b"ab1.2 3.4.5.6 x9_\t\n") + [0, 0xC3, 0xA9, 0xE2, 0x82, 0xFF, 0x80, 0x1F], dtype=np.uint8)
    chars = pool[rng.integers(0, len(pool), int(offs[-1]))]
    valid = np.packbits(rng.random(rows) > 0.02, bitorder="little")
    col = cpulibs.Col(chars, offs, valid)
    g = gpuutil.from_col(col)
    L = gpuutil.lib()

    def snapshot():
        out = {}
        out["replace"] = gpuutil.to_col(g.replace(IPV4, "<IP>"))
        out["replace1"] = gpuutil.to_col(g.replace(r"\d", "#", 2))
        out["lit"] = gpuutil.to_col(g.replace("a", "b", regex=False))
        out["split"] = [gpuutil.to_col(c) for c in g.split(" ")]
        out["split2"] = [gpuutil.to_col(c) for c in g.split(".", 2)]
        from custrings_amd import nvtext
        out["tok"] = gpuutil.to_col(nvtext.tokenize(g))
        out["tok2"] = gpuutil.to_col(nvtext.tokenize(g, " ."))
        out["strip"] = gpuutil.to_col(g.strip())
        re = gpuutil.compile_re(IPV4)
        out["contains"] = gpuutil.bools(g, "cs_contains_re", re)
        cnt = np.zeros(rows, dtype=np.int32)
        found = C.c_int64()
        L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
        out["count"] = (cnt, found.value)
        f = np.zeros(rows, dtype=np.int32)
        L.check(L.lib.cs_find(g.m_cptr, b"3.4", 0, -1, f.ctypes.data, 0, None, C.byref(found)))
        out["find"] = (f, found.value)
        return out

    fast = snapshot()
    for var in ("CS_REGEX_TWO_PASS", "CS_REGEX_ROWWISE", "CS_SPLIT_GENERIC", "CS_TOKENIZE_ROWWISE", "CS_STRIP_ROWWISE",
                "CS_FIND_ROWWISE", "CS_REPLACE_ROWWISE"):
        monkeypatch.setenv(var, "1")
    slow = snapshot()
    for k in fast:
        a, b = fast[k], slow[k]
        if isinstance(a, list):
            assert len(a) == len(b), k
            assert all(x.same_as(y) for x, y in zip(a, b)), k
        elif isinstance(a, tuple):
            assert np.array_equal(a[0], b[0]) and a[1] == b[1], k
        else:
            assert a.same_as(b), k


# ---- extract: capture groups through cs_extract ----------------------
GROUP_PATTERNS = [r"(\w+) (\w+)", r"(a|ab)(c|bcd)", r"(a|b)*c", r"((a)|(b))+", r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"(a*)(b*)", r"(a+?)(a*)",
                  r"(?:x)(y)?z", r"^(\w)(\w*)$", r"(é+)|(a)", r"((\w)\w*) ", r"(x?)(y?)(z?)", r"(.)(.)", r"(GET|POST) (/\S*)", r"(b)?a",
                  r"no_groups", r"()a", r"(a|b|c|d|e|f|g|h){8}(x)?", "(" + "a" * 70 + ")|(b)"]


@pytest.mark.parametrize("route", ["dfa", "lists"])
@pytest.mark.parametrize("pat", GROUP_PATTERNS, ids=[repr(p)[:30] for p in GROUP_PATTERNS])
def test_gpu_vs_oracle_extract(gpu_engine, oracle_engine, pat, route, monkeypatch):
    """route dfa: group ranges carried by the tagged DFA (k_extract_spans_dfa) where the program converts;
    route lists: the anchored list simulation (GroupVm) for every program."""
    if route == "lists":
        monkeypatch.setenv("CS_EXTRACT_LISTS", "1")
    s = fuzzdata.rows(12, 700, alphabet=list("aabbc xyz_.\n019") + ["é", "ü", "😀"]) + fuzzdata.log_rows(7, 700)
    s += ["a" * 80, "ab" * 50, "abcdefgh" * 3, None, ""]
    s += ["z" * 260 + " First Last 10.2.3.4 ab abcd xyz", "ab" * 150 + "c", "é" * 130 + " a b "]  # rows beyond the packed slots (255 bytes)
    assert gpu_engine.extract(s, pat) == oracle_engine.extract(s, pat)


@pytest.mark.parametrize("rows", [1, 63, 64, 65, 257, 4097])
def test_gpu_extract_row_counts(gpu_engine, oracle_engine, rows):
    s = fuzzdata.log_rows(40 + rows, rows)
    pat = r"(\w+) (/\S*) (\d+\.\d+\.\d+\.\d+)?"
    assert gpu_engine.extract(s, pat) == oracle_engine.extract(s, pat)


def test_gpu_extract_empty_column_and_no_groups(gpu_engine):
    assert gpu_engine.extract([], r"(a)") == []
    assert gpu_engine.extract(["a", None], r"a") == []
    assert gpu_engine.extract([None, None], r"(a)(b)") == [[None, None], [None, None]]


@pytest.mark.parametrize("first", [0, 73_000_000])
def test_gpu_c3_extract(orc, first):
    rows = 200_000
    g, o = gpuutil.synth(3, first, rows), orc.synth(3, first, rows)
    pat = r"(\d+)\.(\d+)\.\d+\.(\d+) "
    blob = np.ascontiguousarray(engines.reference_blob(pat))
    gc, oc = g.extract(pat), orc.extract(o, blob)
    assert len(gc) == len(oc) == 3
    for k, (a, b) in enumerate(zip(gc, oc)):
        gpuutil.assert_same(a, b, "extract group %d" % (k + 1))


# ---- findall -------------------------------------------------------
@pytest.mark.parametrize("pat", PATTERNS, ids=[repr(p)[:30] for p in PATTERNS])
def test_gpu_vs_oracle_findall(gpu_engine, oracle_engine, pat):
    s = fuzzdata.rows(13, 500, alphabet=list("aabbc xyz_.\n019") + ["é", "ü", "😀"]) + fuzzdata.log_rows(9, 500)
    s += ["a" * 80, "ab" * 50, None, ""]
    assert gpu_engine.findall(s, pat) == oracle_engine.findall(s, pat)


def test_gpu_findall_edges(gpu_engine):
    assert gpu_engine.findall([], "a") == []
    assert gpu_engine.findall(["xyz", None, ""], "a") == [[None, None, None]]
    assert gpu_engine.findall(["a1b22", None, "", "333"], r"\d+") == [["1", None, None, "333"], ["22", None, None, None]]


@pytest.mark.parametrize("first", [0, 73_000_000])
def test_gpu_c3_findall(orc, first):
    rows = 200_000
    g, o = gpuutil.synth(3, first, rows), orc.synth(3, first, rows)
    blob = np.ascontiguousarray(engines.reference_blob(IPV4))
    gc, oc = g.findall(IPV4), orc.findall(o, blob)
    assert len(gc) == len(oc) == 2
    for k, (a, b) in enumerate(zip(gc, oc)):
        gpuutil.assert_same(a, b, "findall column %d" % k)


# ---- rsplit ---------------------------------------------------------
@pytest.mark.parametrize("seed", [1, 2])
def test_gpu_vs_oracle_rsplit(gpu_engine, oracle_engine, seed):
    s = fuzzdata.rows(seed, 900) + ["a_b_c_d", "  a b  c ", "aaa", "_a_", "aaaa", "a  b", "  ", "x", None, ""]
    o, g = oracle_engine, gpu_engine
    for d in (None, " ", "a", "é", "ab", "éa", ",", "aa", "  ", "aba", "_"):
        for n in (-1, 0, 1, 2, 5):
            assert g.rsplit(s, d, n) == o.rsplit(s, d, n), (d, n)


def test_gpu_c3_rsplit(orc):
    rows = 200_000
    g, o = gpuutil.synth(3, 5_000_000, rows), orc.synth(3, 5_000_000, rows)
    for d, n in ((" ", -1), (" ", 3), (None, 2), (". ", -1)):
        gc, oc = g.rsplit(d, n), orc.rsplit(o, d, n)
        assert len(gc) == len(oc)
        for k, (a, b) in enumerate(zip(gc, oc)):
            gpuutil.assert_same(a, b, "rsplit(%r,%d) col %d" % (d, n, k))


# ---- replace_with_backrefs -----------------------------------------
BACKREF_CASES = [(r"(\w) (\w)", r"\1-\2"), (r"(\d+)\.(\d+)", r"<\2.\1>"), (r"(a|ab)(c|bcd)", r"[\2|\1|\0]"), (r"(a)|(b)", r"\1x\2"),
                 (r"(é+)", r"\1\1"), (r"b", r"\0\0"), (r"(a)(b)?", r"\2\9_"), (r"(GET|POST) (/\S*)", r"\2 \1"), (r"((a|b)(c|x))+", r"\3\2\1"),
                 (r"\b(\w)(\w*)", r"\2\1"), (r"x(?:y)(z)", r"\1"), (r"(.)", r"\1,"), (r"(a|b|c|d|e|f|g|h){8}(x)?", r"<\1\2>")]


@pytest.mark.parametrize("route", ["tiles", "lists", "rowwise"])
@pytest.mark.parametrize("pat,repl", BACKREF_CASES, ids=[repr(p)[:24] for p, _ in BACKREF_CASES])
def test_gpu_vs_oracle_replace_with_backrefs(gpu_engine, oracle_engine, pat, repl, route, monkeypatch):
    if route == "lists":
        monkeypatch.setenv("CS_REGEX_NO_TDFA", "1")
    if route == "rowwise":
        monkeypatch.setenv("CS_REGEX_ROWWISE", "1")
    s = fuzzdata.rows(14, 600, alphabet=list("aabbc xyz_.\n019") + ["é", "ü", "😀"]) + fuzzdata.log_rows(11, 600)
    s += ["a" * 80, "ab" * 50, "abcdefgh" * 3, None, ""]
    assert gpu_engine.replace_with_backrefs(s, pat, repl) == oracle_engine.replace_with_backrefs(s, pat, repl)


@pytest.mark.parametrize("route", ["dfa", "lists"])
def test_gpu_replace_with_backrefs_edges(gpu_engine, route, monkeypatch):
    if route == "lists":
        monkeypatch.setenv("CS_REGEX_NO_TDFA", "1")
    col = gpu_engine.col(["a1", None, ""])
    assert col.replace_with_backrefs(r"(\d)", None).to_host() == [None, None, None]
    with pytest.raises(ValueError):
        col.replace_with_backrefs("", "x")
    with pytest.raises(ValueError):
        col.replace_with_backrefs(r"(a*)", r"\1")  # matches the empty string: the reference would not terminate
    assert gpu_engine.col([]).replace_with_backrefs(r"(a)", r"\1").to_host() == []


def test_gpu_c3_replace_with_backrefs(orc):
    rows = 200_000
    g, o = gpuutil.synth(3, 9_000_000, rows), orc.synth(3, 9_000_000, rows)
    pat, repl = r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"
    blob = np.ascontiguousarray(engines.reference_blob(pat))
    gpuutil.assert_same(g.replace_with_backrefs(pat, repl), orc.replace_with_backrefs(o, blob, repl), "replace_with_backrefs")


# ---- record (row-major) forms of extract / findall --------------------------------------------------
def _transpose(cols, ragged):
    rows = len(cols[0]) if cols else 0
    out = []
    for r in range(rows):
        rec = [c[r] for c in cols]
        if ragged:
            n = 0
            while n < len(rec) and rec[n] is not None:
                n += 1
            rec = rec[:n]
        out.append(rec)
    return out


def test_gpu_record_forms_reference_vectors(gpu_engine):
    # cpp/tests/test_extract.cpp:27-52
    s = ["First Last", "Joe Schmoe", "John Smith", "Jane Smith", "Beyonce", "Sting", None, ""]
    got = [r.to_host() for r in gpu_engine.col(s).extract_record(r"(\w+) (\w+)")]
    assert got == [["First", "Last"], ["Joe", "Schmoe"], ["John", "Smith"], ["Jane", "Smith"]] + [[None, None]] * 4
    # python/tests/test_regex.py:129-136
    s = ["hello", "and héllo", "this was empty", "", "another"]
    got = [r.to_host() for r in gpu_engine.col(s).findall_record("[aA]")]
    assert got == [[], ["a"], ["a"], [], ["a"]]


@pytest.mark.parametrize("pat", [r"(\w+) (\w+)", r"(a)|(b)", r"(\d+)\.(\d+)", r"(x?)(y?)(z?)", r"\d+", r"a|b", r"\w+"])
def test_gpu_record_forms_vs_oracle(gpu_engine, oracle_engine, pat):
    s = fuzzdata.rows(15, 400, alphabet=list("aabbc xyz_.\n019") + ["é", "ü", "😀"]) + fuzzdata.log_rows(12, 300) + [None, ""]
    col = gpu_engine.col(s)
    want = oracle_engine.findall(s, pat)
    flat, loff = col.findall_record(pat, flat=True)
    recs = _transpose(want, True)
    assert flat.to_host() == [x for rec in recs for x in rec]
    assert loff.tolist() == [0] + list(np.cumsum([len(rec) for rec in recs]))
    want = oracle_engine.extract(s, pat)
    if want:
        flat, loff = col.extract_record(pat, flat=True)
        recs = _transpose(want, False)
        assert flat.to_host() == [x for rec in recs for x in rec]
        assert loff.tolist() == list(range(0, len(s) * len(want) + 1, len(want)))
    else:
        assert col.extract_record(pat) == []


def test_gpu_two_streams_share_the_buffer_cache(orc):
    """Ops issued on two HIP streams in turn: blocks released on one stream are taken by the other (the cache then makes
    the taker wait for the release EVENT on the device, cs_core.hip: CachedBlock).  Results equal the oracle's, and the
    interleaving does not stall the host per released block (a loose bound on the wall time against one stream)."""
    import ctypes as C
    import time

    import torch

    from custrings_amd import _lib, nvstrings

    L = _lib.lib
    rows = 200_000
    g, o = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    re = nvstrings._compile(r"\d+\.\d+\.\d+\.\d+")
    blob = np.ascontiguousarray(engines.reference_blob(r"\d+\.\d+\.\d+\.\d+") if engines.reference_blob(r"\d+\.\d+\.\d+\.\d+") is not None
                                else engines.product_blob(r"\d+\.\d+\.\d+\.\d+"), dtype=np.int32)
    want_low, want_rep = orc.lower(o), orc.replace_re(o, blob, "<IP>")
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def round_on(stream):
        h = C.c_void_p(stream.cuda_stream)
        a, b = C.c_void_p(), C.c_void_p()
        _lib.check(L.cs_lower(g.m_cptr, h, C.byref(a)))
        _lib.check(L.cs_replace_re(g.m_cptr, re, b"<IP>", -1, h, C.byref(b)))
        return nvstrings.nvstrings(a.value), nvstrings.nvstrings(b.value)

    def timed(seq, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            x, y = round_on(seq[i % len(seq)])
            del x, y
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    one = timed(streams[:1], 40)
    for i in range(6):
        low, rep = round_on(streams[i % 2])
        gpuutil.assert_same(low, want_low, "lower on stream %d" % (i % 2))
        gpuutil.assert_same(rep, want_rep, "replace_re on stream %d" % (i % 2))
        del low, rep
    two = timed(streams, 40)
    assert two < 5 * one + 0.05, (one, two)
    L.cs_regex_destroy(re)
