"""-m gpu: the SURVEY.md section 8(f) rows built in round 2 -- array / combine ops, record and
partition forms of split, replace_re with several patterns, the NVCategory remap family, the
NVText counters -- through the C ABI against the oracle on seeded random columns.  (The
reference's own test vectors for these ops run in test_gpu_parity.py::test_gpu_golden.)"""
import random

import numpy as np
import pytest

import fuzzdata

pytestmark = pytest.mark.gpu


def _rows(seed, n, **kw):
    return fuzzdata.rows(seed, n, **kw)


def _same(f, g, *args):
    """both engines give the same result, or both raise (the reference throws there)"""
    try:
        exp = f(*args)
    except (ValueError, IndexError):
        with pytest.raises((ValueError, IndexError)):
            g(*args)
        return
    assert g(*args) == exp, args


@pytest.mark.parametrize("seed", [3, 4])
def test_gpu_array_ops(gpu_engine, oracle_engine, seed):
    g, o = gpu_engine, oracle_engine
    s = _rows(seed, 1200, max_len=30)
    rnd = random.Random(seed)
    assert g.len(s) == o.len(s)
    pos = [rnd.randrange(len(s)) for _ in range(900)]
    assert g.gather(s, pos) == o.gather(s, pos)
    with pytest.raises(IndexError):
        g.gather(s, [0, len(s)])
    with pytest.raises(IndexError):
        g.gather(s, [-1])
    for start, end, step in ((0, len(s), 2), (5, 400, 3), (900, 10, -7), (len(s), 0, -1), (3, 3, 1), (7, 2, 1), (0, 10 ** 6, 5)):
        _same(o.sublist, g.sublist, s, start, end, step)  # (start == size with a negative step: gather(size) throws)
    for stype in (0, 1, 2, 3):
        for asc in (True, False):
            for nf in (True, False):
                assert g.order(s, stype, asc, nf) == o.order(s, stype, asc, nf), (stype, asc, nf)
                assert g.sort(s, stype, asc, nf) == o.sort(s, stype, asc, nf)
    strs = _rows(seed + 50, 300, max_len=12)
    p2 = [rnd.randrange(-3, len(s) + 3) for _ in range(300)]
    assert g.scatter(s, strs, p2) == o.scatter(s, strs, p2)
    assert g.scalar_scatter(s, "é+", p2) == o.scalar_scatter(s, "é+", p2)
    assert g.scalar_scatter(s, None, p2[:40]) == o.scalar_scatter(s, None, p2[:40])


def test_gpu_gather_mask_and_from_index(gpu_engine):
    from custrings_amd import _lib, nvstrings
    import ctypes as C

    s = _rows(9, 500, max_len=20)
    col = gpu_engine.col(s)
    mask = [i % 3 == 0 for i in range(len(s))]
    assert col.gather(mask).to_host() == [x for x, m in zip(s, mask) if m]
    assert col.remove_strings([0, 2, 499, 2]).to_host() == [x for i, x in enumerate(s) if i not in (0, 2, 499)]
    # create_from_index over the column's own device buffers: (pointer, length) pairs, null pointer = null row
    chars, offs, valid = col._export64()
    v = _lib.ColumnView()
    _lib.check(_lib.lib.cs_column_get_view(col.m_cptr, C.byref(v)))
    pairs = np.zeros((len(s), 2), dtype=np.uint64)
    order = list(range(len(s)))[::-1]
    for j, r in enumerate(order):
        if s[r] is not None:
            pairs[j, 0] = v.chars + int(offs[r]) if offs[r + 1] > offs[r] else v.chars  # (an empty string still needs a non-null pointer)
            pairs[j, 1] = int(offs[r + 1] - offs[r])
    out = C.c_void_p()
    _lib.check(_lib.lib.cs_column_from_index(pairs.ctypes.data, len(s), 0, 0, None, C.byref(out)))
    assert nvstrings.nvstrings(out.value).to_host() == [s[r] for r in order]
    _lib.check(_lib.lib.cs_column_from_index(pairs.ctypes.data, len(s), 0, 2, None, C.byref(out)))  # sorted by name, nulls first
    assert nvstrings.nvstrings(out.value).to_host() == col.sort(2).to_host()


@pytest.mark.parametrize("seed", [5, 6])
def test_gpu_combine(gpu_engine, oracle_engine, seed):
    g, o = gpu_engine, oracle_engine
    a, b_, c = (_rows(seed + k, 700, max_len=15) for k in (0, 10, 20))
    for sep in (None, "", ":", "é-"):
        for narep in (None, "", "_", "ná"):
            assert g.cat(a, [b_], sep, narep) == o.cat(a, [b_], sep, narep), (sep, narep)
            assert g.cat(a, [b_, c], sep, narep) == o.cat(a, [b_, c], sep, narep), (sep, narep)
            assert g.join(a, sep or "", narep) == o.join(a, sep or "", narep), (sep, narep)
    with pytest.raises(ValueError):
        g.cat(a, [b_[:10]], None, None)
    assert g.join([], ",", None) == o.join([], ",", None) == [""]
    assert g.join([None, None], ",", None) == o.join([None, None], ",", None)


RECORD_ROWS = ["", None, "a b", " a b ", "  aa  bb  ", " a  bbb   c", " aa b  ccc  ", "héllo", "a_bc_déf", "a__bc", "_ab_cd", "ab_cd_",
               "_", "__", "\ta\nb  c", "x" * 70 + " y", "é é é"]


@pytest.mark.parametrize("delim", [None, " ", "_", "b", "é"])
@pytest.mark.parametrize("n", [-1, 1, 2, 3])
def test_gpu_split_records(gpu_engine, oracle_engine, delim, n):
    g, o = gpu_engine, oracle_engine
    s = RECORD_ROWS + fuzzdata.rows(21, 400, max_len=24, alphabet=list("ab _é\t") + ["cd"])
    assert g.split_record(s, delim, n) == o.split_record(s, delim, n)
    assert g.rsplit_record(s, delim, n) == o.rsplit_record(s, delim, n)


@pytest.mark.parametrize("delim", [" ", "_", "ab", "é", "é_"])
def test_gpu_partition(gpu_engine, oracle_engine, delim):
    g, o = gpu_engine, oracle_engine
    s = RECORD_ROWS + fuzzdata.rows(22, 400, max_len=24, alphabet=list("ab _é"))
    assert g.partition(s, delim, False) == o.partition(s, delim, False)
    assert g.partition(s, delim, True) == o.partition(s, delim, True)
    assert gpu_engine.col(s).partition("") == [] and gpu_engine.col(s).rpartition(None) == []


def test_gpu_records_flat_form(gpu_engine):
    """the native record form: one column + list offsets, no per-row objects"""
    c = gpu_engine.col(["a b c", None, "", "d"])
    flat, lst = c.split_record(" ", -1, flat=True)
    assert flat.to_host() == ["a", "b", "c", "", "d"] and lst.tolist() == [0, 3, 3, 4, 5]
    flat = c.partition(" ", flat=True)
    assert flat.to_host() == ["a", " ", "b c", None, None, None, "", "", "", "d", "", ""]


@pytest.mark.parametrize("pats,repls", [([r"\d+", "ab", r"\bc"], ["<N>", "", "C"]), (["a+", "b"], ["_"]), ([r"[aeiou]", r"\s+"], [None, " "]),
                                        (["é", "a.c", "xyz"], ["e", "<>", ""]), (["(ab|a)(bc|c)?", "b"], ["1", "2"])])
def test_gpu_replace_multi(gpu_engine, oracle_engine, pats, repls):
    g, o = gpu_engine, oracle_engine
    s = fuzzdata.rows(31, 600, alphabet=list("aabbc xyz_.019") + ["é", "ü"]) + fuzzdata.log_rows(7, 300) + ["", None, "abcabc"]
    assert g.replace_multi(s, pats, repls) == o.replace_multi(s, pats, repls)


def test_gpu_replace_multi_arguments(gpu_engine):
    c = gpu_engine.col(["abc"])
    with pytest.raises(ValueError):
        c.replace_multi([], ["x"])
    with pytest.raises(ValueError):
        c.replace_multi(["a", "b"], ["1", "2", "3"])
    with pytest.raises(ValueError):
        c.replace_multi(["a*"], ["x"])  # matches the empty string: the reference does not terminate
    with pytest.raises(ValueError):
        c.replace_multi("a", ["x"])  # patterns must be a list (nvstrings.py:1516-1518)
    assert c.replace_multi(["a.c"], "x", regex=False).to_host() == ["abc"]  # literal: '.' is not a wildcard
    assert c.replace_multi(["b", "c"], ["1", "2"], regex=False).to_host() == ["a12"]
    # a program the tagged DFA does not take (many simultaneous threads) runs the list simulator
    big = "(a|b|c|d|e|f|g|h){8}"
    d = gpu_engine.col(["abcdefgh-abcdefgh", "abc", None])
    assert d.replace_multi([big, "-"], ["#", "+"]).to_host() == ["#+#", "abc", None]


@pytest.mark.parametrize("seed", [41, 42])
def test_gpu_text_counters(gpu_engine, oracle_engine, seed):
    g, o = gpu_engine, oracle_engine
    s = fuzzdata.rows(seed, 800, max_len=40, alphabet=list("ab cd_-é\t") + ["the", "cat"])
    for d in (None, " ", "_-", "é "):
        assert g.token_count(s, d) == o.token_count(s, d), d
        assert g.unique_tokens(s, d) == o.unique_tokens(s, d), d
        tk = o.unique_tokens(s, d)[:25] + [None, "zzz"]
        assert g.tokens_counts(s, tk, d) == o.tokens_counts(s, tk, d), d
        tg = ["the", "cat", "a", None, "ab"]
        assert g.replace_tokens(s, tg, ["T", "é", "", None, "longer replacement"], d) == o.replace_tokens(s, tg, ["T", "é", "", None, "longer replacement"], d)
        assert g.replace_tokens(s, tg, ["*"], d) == o.replace_tokens(s, tg, ["*"], d)
    assert g.normalize_spaces(s) == o.normalize_spaces(s)
    assert g.normalize_spaces(["", "  ", None]) is None  # nothing to hold: no instance (tokens.cu:703-704)


@pytest.mark.parametrize("seed", [51, 52])
def test_gpu_category_family(gpu_engine, oracle_engine, seed):
    g, o = gpu_engine, oracle_engine
    rnd = random.Random(seed)
    words = ["w%d" % i for i in range(30)] + ["é", "", "zz top"]
    s = [rnd.choice(words + [None]) for _ in range(600)]
    t = [rnd.choice(words[10:] + ["new1", "new2", None]) for _ in range(200)]
    nk = len(o.category(s)[0])
    assert g.cat_to_strings(s) == o.cat_to_strings(s) == s
    pos = [rnd.randrange(nk) for _ in range(300)]
    assert g.cat_gather_strings(s, pos) == o.cat_gather_strings(s, pos)
    assert g.cat_gather(s, pos + [-1]) == o.cat_gather(s, pos + [-1])
    assert g.cat_gather_and_remap(s, pos) == o.cat_gather_and_remap(s, pos)
    for bad in ([0, nk], [-1] if True else []):
        with pytest.raises(IndexError):
            g.cat_gather_strings(s, bad)
        with pytest.raises(IndexError):
            g.cat_gather_and_remap(s, bad)
    with pytest.raises(IndexError):
        g.cat_gather(s, [-2])
    for fn in ("cat_add_strings", "cat_remove_strings", "cat_add_keys", "cat_remove_keys", "cat_set_keys", "cat_merge_category",
               "cat_merge_and_remap"):
        got, exp = getattr(g, fn)(s, t), getattr(o, fn)(s, t)
        assert (list(got[0]), list(got[1])) == (list(exp[0]), list(exp[1])), fn
    # remove_unused_keys after set_keys (python/tests/test_category.py:213-220)
    keys, values = o.cat_set_keys(s, t)
    exp = o.cat_remove_unused_keys(keys, values)
    got = g.cat_set_keys_then_remove_unused(s, t)
    assert (list(got[0]), list(got[1])) == (list(exp[0]), list(exp[1]))
    # the key-only expectations of the reference's tests (python/tests/test_category.py:191-220)
    k = ["a", "b", "b", "f", "c", "f"]
    assert g.cat_add_keys(k, ["a", "b", "c", "d"])[0] == ["a", "b", "c", "d", "f"]
    assert g.cat_remove_keys(k, ["b", "d"])[0] == ["a", "c", "f"]
    assert g.cat_set_keys(k, ["b", "c", "e", "d"])[0] == ["b", "c", "d", "e"]
    assert g.cat_set_keys_then_remove_unused(k, ["b", "c", "e", "d"])[0] == ["b", "c"]


def test_gpu_tokenize_multi(gpu_engine, oracle_engine):
    g, o = gpu_engine, oracle_engine
    s = fuzzdata.rows(61, 700, max_len=40, alphabet=list("ab c,.;") + ["--", "é", "the"])
    for delims in ([" "], [",", "."], ["--", " ", "é"], ["the", "a"], [None, "", ";"], ["ab", "a"]):
        assert g.tokenize_multi(s, delims) == o.tokenize_multi(s, delims), delims
    assert g.tokenize_multi(s, []) == o.tokenize(s, None)


UNIT_PATS = [r"\d+\.\d+\.\d+\.\d+", r"\d+", r"[a-c]+@[a-c]+", r"a+b", r"\w+ \w+", r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", r"\d+$", r"[^x]+"]


@pytest.mark.parametrize("pat", UNIT_PATS)
def test_gpu_replace_re_unit_scan_edges(gpu_engine, oracle_engine, pat):
    """k_tdfa_replace_stream<.., UNITS> (the scan per unit instead of per row, regex_tdfa.cpp header word 31) against the
    oracle on columns built for its edges: sub-tiles with no unit at all, with up to 64 units (one round), 65..128
    (two rounds) and more than the queue holds (the rows are then scanned whole), rows with more matches than the
    register records keep, units at both ends of a row, null and empty rows in between, a non-ASCII row that sends
    its sub-tile to the generic scan, and growing / shrinking / empty replacements."""
    rnd = random.Random(len(pat))
    ip = lambda: ".".join(str(rnd.randrange(256)) for _ in range(4))
    word = lambda: "".join(rnd.choice("abcx@ ") for _ in range(rnd.randrange(1, 7)))
    blocks = []
    blocks.append([word() + "x" for _ in range(64)])                                   # few or no units
    blocks.append([("%s %s" % (word(), ip())) if i % 2 else word() for i in range(64)])  # about 32 units
    blocks.append(["%s %s x" % (ip(), ip()) for _ in range(45)] + [None, ""] * 9 + ["q"])  # 65..128 units: two rounds
    blocks.append([" ".join(str(rnd.randrange(10)) for _ in range(30)) for _ in range(64)])  # 30 units per row: beyond the queue
    blocks.append(["1.2.3.4 5.6.7.8 9.9.9.9 1.1.1.1 2.2.2.2 3.3.3.3 a@b c@c ab aab" for _ in range(20)] + [ip() for _ in range(44)])  # > 4 matches per row
    blocks.append([ip()] * 30 + ["é " + ip()] + [ip()] * 33)                             # one non-ASCII row
    blocks.append([ip() + " ", " " + ip(), ip(), "." + ip() + ".", ip() + "." + ip()] * 12 + ["12", "1.2", "", None])
    s = [r for b in blocks for r in b] + fuzzdata.log_rows(31, 700)
    from custrings_amd import _lib

    before = _lib.lib.cs_fallback_count()
    for repl in ("<IP>", "", "<a-much-longer-one>"):
        assert gpu_engine.replace_re(s, pat, repl, -1) == oracle_engine.replace_re(s, pat, repl, -1), (pat, repl)
    assert _lib.lib.cs_fallback_count() == before  # the single-pass kernel itself produced these results


CHAIN_PATS = [r"\d+\.\d+\.\d+\.\d+", r"[0-9]+\.[0-9]+\.[0-9]+\.[0-9]+", r"\d+", r"(\d+)\.(\d+)", r"[a-c]+@[a-c]+", r"\d\.\d+", r"\d+-+\d+", r"[a-cx-z]+_", r"a+b",
              # a literal suffix behind the chain (regex_tdfa.cpp: no unit decomposition, the chain brings its own x)
              r"\d+\.\d+\.\d+\.\d+ ", r"(\d+)\.(\d+)\.\d+\.(\d+) ", r"\d+\.\d+ -", r"[a-c]+@x", r"\d+-\.",
              # counted items, a `\b` at either end (round 5: the 26-instruction dotted quad of the ops table)
              r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", r"\b(\d{1,3})\.(\d{1,3})\b", r"\b(\d+)\.(\d+)\b", r"\d\d+\.\d+", r"(\d+)\.(\d{2})\.(\d+)", r"\d+\.\d\b",
              r"\d+\.{1,2}\d+", r"\b[a-c]{2,3}@", r"\d+\.(\d{1,2})x", r"\b\d{2}-\d{2}\b"]


@pytest.mark.gpu
@pytest.mark.parametrize("pat", CHAIN_PATS)
def test_gpu_chain_patterns(gpu_engine, oracle_engine, pat):
    """Chain patterns (regex_tdfa.h: chain_match -- every match of a plain-ASCII row from its candidate / x bit masks by
    integer arithmetic, in the replace and scan stream kernels): replace_re, count_re, contains_re and findall against
    the oracle on overlapping candidates (1.2.3.4.5.6.7.8), matches at both row ends, adjacent matches, rows at the
    96-byte mask limit and beyond it (those sub-tiles keep the automaton), null / empty rows, a non-ASCII row."""
    rnd = random.Random(len(pat) + 5)
    ip = lambda: ".".join(str(rnd.randrange(256)) for _ in range(4))
    junk = lambda k: "".join(rnd.choice("abcxyz@-_. 0123456789") for _ in range(rnd.randrange(0, k)))
    s = [junk(90) for _ in range(640)]
    s += ["1.2.3.4.5.6.7.8", "1.2.3.4 5.6.7.8", ".1.2.3.4.", "1..2.3.4", "999.999.999.999x1.1.1.1", "1.2.3.", "12", "", None, "0.0.0.0" * 13,
          "9" * 93, "1." * 46, ".1" * 46, "a@b@c@@", "ab@cb@ca", "1--2-3", "12.3456", "a_b_cx_", "aab ab b a",
          "a1.2.3.4", "1.2.3.4a", "a1.2.3.4.5", "1.2.3.4.5a", "1234.5.6.7", "1.2345.6.7", "1.2.3.4567", "123.123.123.123", "1.2.3.4_5.6.7.8", "_1.2.3.4_",
          "12-34-56", "12-34 123-45 12-345 x12-34", "1..2 1...234 1.2x 1.23x 1.234x", "ab@ abc@ abcd@ xab@ a@", "1.2" + "9" * 90, "1.2.3.4b 1.2.3.4"] * 3
    s += [ip() + " " + junk(40) + " " + ip() for _ in range(300)]
    s += [junk(60) for _ in range(64)] + ["é " + ip()] + [ip() for _ in range(63)]     # a sub-tile with a non-ASCII row
    s += [ip() + "x" * 120 + ip()] + [ip() for _ in range(63)]                        # a row beyond the masks
    s += fuzzdata.log_rows(37, 1500)
    from custrings_amd import _lib

    before = _lib.lib.cs_fallback_count()
    for repl in ("<IP>", "", "#"):
        assert gpu_engine.replace_re(s, pat, repl, -1) == oracle_engine.replace_re(s, pat, repl, -1), (pat, repl)
    assert _lib.lib.cs_fallback_count() == before
    assert gpu_engine.count_re(s, pat) == oracle_engine.count_re(s, pat)
    assert gpu_engine.contains_re(s, pat) == oracle_engine.contains_re(s, pat)
    assert gpu_engine.findall(s, pat) == oracle_engine.findall(s, pat)
    assert gpu_engine.replace_re(s, pat, "=", 1) == oracle_engine.replace_re(s, pat, "=", 1)
    if "(" in pat:  # extract: the chain form of the scan kernel (the first match's group ranges off the item boundaries)
        assert gpu_engine.extract(s, pat) == oracle_engine.extract(s, pat)
        refs = "".join("\\%d." % (g + 1) for g in reversed(range(pat.count("("))))
        assert gpu_engine.replace_with_backrefs(s, pat, refs) == oracle_engine.replace_with_backrefs(s, pat, refs)


@pytest.mark.gpu
def test_gpu_chain_extract_meets_non_ascii_tiles(gpu_engine, oracle_engine):
    """extract on the chain form of the scan kernel, chosen from a sample of the chars: sub-tiles with non-ASCII rows, a NUL
    byte or a row beyond the 96-byte masks take the form's automaton route -- same columns as the oracle."""
    rnd = random.Random(78)
    ip = lambda: ".".join(str(rnd.randrange(256)) for _ in range(4))
    s = ["%s GET /x/%d %s " % (ip(), i, ip() if i % 3 == 0 else "-") for i in range(40000)]
    for at in (9000, 9001, 9100, 29000, 31000):
        s[at] = "é " + ip() + " ü" + ip() + " "
    s[12000] = "1.2.3\x004.5.6.7 8.9.10.11 "
    s[20000] = "x" * 120 + ip() + " "
    s[30500] = None
    s[30501] = ""
    for pat in (r"(\d+)\.(\d+)\.\d+\.(\d+) ", r"(\d+)\.(\d+)", r"((\d+)\.\d+)\.(\d+\.(\d+))", r"(\d+)\.\d+ -"):
        assert gpu_engine.extract(s, pat) == oracle_engine.extract(s, pat), pat


def test_gpu_chain_form_meets_non_ascii_tiles(gpu_engine, oracle_engine):
    """The CHAIN form of the replace stream kernel (no unit / lean scans compiled in) is chosen from a SAMPLE of the chars
    (start, middle, end): a column whose sampled windows are plain ASCII but which holds non-ASCII rows and a NUL byte
    elsewhere sends those sub-tiles through the form's generic scan -- same bytes as the oracle, no fallback."""
    rnd = random.Random(77)
    ip = lambda: ".".join(str(rnd.randrange(256)) for _ in range(4))
    s = ["%s GET /x/%d %s" % (ip(), i, ip() if i % 3 == 0 else "-") for i in range(40000)]   # about 1.3 MB of chars
    for at in (9000, 9001, 9100, 29000, 31000):
        s[at] = "é " + ip() + " ü" + ip()
    s[12000] = "1.2.3.4\x005.6.7.8"
    s[30500] = None
    from custrings_amd import _lib

    before = _lib.lib.cs_fallback_count()
    for pat in (r"\d+\.\d+\.\d+\.\d+", r"\d+"):
        for repl in ("<IP>", "<a-longer-replacement>"):
            assert gpu_engine.replace_re(s, pat, repl, -1) == oracle_engine.replace_re(s, pat, repl, -1), (pat, repl)
        # (the scan stream kernel has the same form for count_re / findall)
        assert gpu_engine.count_re(s, pat) == oracle_engine.count_re(s, pat), pat
        assert gpu_engine.findall(s, pat) == oracle_engine.findall(s, pat), pat
    assert _lib.lib.cs_fallback_count() == before


@pytest.mark.parametrize("n,sep", [(2, "_"), (3, ""), (5, "--")])
def test_gpu_sharded_ngrams_pieces(gpu_engine, n, sep):
    """custrings_amd/dist.py sharded_ngrams on the GPU ops (drop_empty / head / export / column / concat / ngrams): the
    per-rank step run for three shards in one process -- the exchange itself is covered by the gloo tests -- gives,
    shard after shard, nvtext.ngrams of the whole token column."""
    from custrings_amd import dist as csd, nvstrings, nvtext

    ops = csd.GpuOps()
    rows = fuzzdata.rows(77, 400, max_len=40, alphabet=list("ab cd  e")) + ["x"]
    whole = nvtext.tokenize(nvstrings.to_device(rows))
    want = nvtext.ngrams(whole, n, sep).to_host()
    for cuts in ([0, 150, 300, 401], [0, 1, 2, 401], [0, 0, 400, 401], [0, 399, 401, 401]):
        shards = [nvtext.tokenize(nvstrings.to_device(rows[cuts[r]:cuts[r + 1]])) for r in range(3)]
        # (a shard with a null / empty token row in it: create_ngrams drops those rows)
        shards[1] = ops.concat([shards[1], nvstrings.to_device(["", None])]) if shards[1].size() else shards[1]
        mine = [ops.drop_empty(t) for t in shards]
        heads = []
        for m in mine:
            chars, offs, _ = ops.export(ops.head(m, n))
            heads.append(ops.column(chars, offs, False))
        counts = [m.size() for m in mine]
        got = []
        for r in range(3):
            got += csd._ngrams_of_shard(ops, r, 3, shards[r], mine[r], heads, counts, n, sep).to_host()
        assert got == want, cuts


def _ipc_child(col_rec, cat_rec, part_rec, q):
    import os
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    try:
        from custrings_amd import nvcategory, nvstrings

        s = nvstrings.create_from_ipc(col_rec)
        c = nvcategory.create_from_ipc(cat_rec)
        part = nvstrings.create_from_ipc(part_rec)  # (a split output column: int32 offsets)
        q.put(("ok", s.to_host(), s.upper().to_host(), c.keys().to_host(), c.values(), part.to_host()))
    except Exception as e:  # the parent reports it
        q.put(("error", repr(e)))


def test_gpu_ipc_transfer_between_processes(gpu_engine):
    """NVStrings / NVCategory::create_ipc_transfer + create_from_ipc (NVStrings.h:132,214; ipc_transfer.h:31-200) on HIP
    IPC handles: a second process maps the exporter's buffers, reads them and runs an op on them, no copy in between."""
    import multiprocessing as mp

    from custrings_amd import nvcategory, nvstrings

    rows = fuzzdata.rows(5, 3000, max_len=30)
    s = nvstrings.to_device(rows)
    cat = nvcategory.from_strings(s)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    parts = s.split(" ")
    p = ctx.Process(target=_ipc_child, args=(s.get_ipc_data(), cat.get_ipc_data(), parts[0].get_ipc_data(), q))
    p.start()
    got = q.get(timeout=180)
    p.join(timeout=60)
    assert got[0] == "ok", got
    assert got[1] == rows
    assert got[2] == s.upper().to_host()
    assert got[3] == cat.keys().to_host() and list(got[4]) == list(cat.values())
    assert got[5] == parts[0].to_host()
    with pytest.raises(ValueError):  # a record that is not one
        nvstrings.create_from_ipc(b"\0" * 231 + b"\7")


def test_gpu_to_device_keeps_embedded_nul(gpu_engine):
    """nvstrings.to_device with a NUL inside a string: the C-string ingest would cut the string there, so such lists take
    the offsets ingest; the round trip and a byte-level op keep the NUL."""
    from custrings_amd import nvstrings

    rows = ["ab\0cd", None, "", "\0", "x y\0z w"]
    s = nvstrings.to_device(rows)
    assert s.to_host() == rows and s.len() == [5, None, 0, 1, 7]
    assert s.upper().to_host() == ["AB\0CD", None, "", "\0", "X Y\0Z W"]
    assert [c.to_host() for c in s.split(" ")] == [["ab\0cd", None, "", "\0", "x"], [None, None, None, None, "y\0z"], [None, None, None, None, "w"]]


@pytest.mark.parametrize("config,extra", [("c4", ["--rows", "200000", "--keys", "500"]), ("c5", ["--rows", "50000"])])
def test_gpu_bench_exchange_configs_run(config, extra):
    """bench.py --config c4|c5 (the two BASELINE configs with an exchange step; at one rank the collectives are skipped):
    one JSON line with the contract's fields, and the work it reports is plausible."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", config, "--steps", "2", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["fallbacks_in_timed_region"] == 0
    if config == "c4":
        assert 0 < d["rank0"]["keys"] <= 501
    else:
        assert d["rank0"]["ngrams"] == d["rank0"]["tokens"] - 1


@pytest.mark.parametrize("needle", ["a", "ab", " ", "the ", "abcdefgh", "b a", ".", "aa", "aba", "abcdefghi"])
def test_gpu_literal_replace_by_byte_comparison(gpu_engine, oracle_engine, needle):
    """replace(str) with a needle of up to eight bytes and no border takes the byte-comparison scan of the replace stream
    kernel (cs_regex.hip: StreamArgs::lit; 'aa', 'aba' have a border and nine bytes are too many: those stay on the
    automaton): against the oracle on rows with non-ASCII text, nulls, empties, matches at both row ends and across what
    would be a row boundary, with growing / shrinking / empty / 16-byte replacements, and at the row-count edges."""
    rnd = random.Random(len(needle) * 7)
    words = ["a", "ab", "abc", "the ", "the", " ", "  ", "é", "😀", "abcdefgh", "abcdefg", "b a", ".", "..", "aa", "aba", "x", "hab", "tha"]
    rows = []
    for _ in range(1500):
        u = rnd.random()
        if u < 0.05:
            rows.append(None)
        elif u < 0.1:
            rows.append("")
        else:
            rows.append("".join(rnd.choice(words) for _ in range(rnd.randrange(1, 14)))[:90])
    rows += [needle, needle * 3, needle[:-1] if len(needle) > 1 else "", needle[1:] + needle[:1], "x" + needle, needle + "x"]
    from custrings_amd import _lib

    for repl in ("", "x", "xyz", "0123456789abcdef"):
        before = _lib.lib.cs_fallback_count()
        for cut in (len(rows), 65, 64, 1):
            assert gpu_engine.replace(rows[:cut], needle, repl, -1) == oracle_engine.replace(rows[:cut], needle, repl, -1), (needle, repl, cut)
        # (a 16-byte replacement of a one-byte needle may outgrow the provisioned room: that launch is repeated two-pass)
        assert len(repl) > 3 or _lib.lib.cs_fallback_count() == before


def test_gpu_literal_replace_does_not_match_across_sub_tiles(gpu_engine, oracle_engine):
    """Found by tools/soak_gpu.py: with all 64 rows of a sub-tile present nobody marked the end of the staged span, so the
    last row's 'a' and the next sub-tile's leading 'b' matched 'ab'.  Rows of equal length whose ends and starts line up."""
    rows = ["xa", "bx"] * 3000 + ["a", "b"] * 500 + ["ab"]
    for needle, repl in (("ab", "x"), ("ab", "12345"), ("xab", ""), ("a", "")):
        assert gpu_engine.replace(rows, needle, repl, -1) == oracle_engine.replace(rows, needle, repl, -1), (needle, repl)


def test_gpu_backrefs_on_columns_the_unit_kernel_declines(gpu_engine, oracle_engine):
    """Found by tools/soak_gpu.py: a column whose 64-row spans exceed the stream kernel's staging fell through to the older
    tile kernel, which knows nothing of the template and returned a plain replace.  Long rows, then a column with one
    row beyond the 96-byte masks, then one with a non-ASCII row (the unit route hands those launches over)."""
    pat, repl = r"(\d+)\.(\d+)", r"\2.\1"
    long_rows = [("w%d 12.34 " % i) * 30 for i in range(300)]
    mixed = ["a 1.2 b"] * 200 + ["x" * 120 + " 3.4"] + ["c 5.6"] * 200
    accents = ["a 1.2 b"] * 100 + ["é 7.8"] + ["c 5.6"] * 100
    for rows in (long_rows, mixed, accents):
        assert gpu_engine.replace_with_backrefs(rows, pat, repl) == oracle_engine.replace_with_backrefs(rows, pat, repl)


@pytest.mark.parametrize("shape", ["short", "long_spans", "beyond_staging"])
def test_gpu_span_bytes_written_by_sub_tiles(gpu_engine, oracle_engine, shape, monkeypatch):
    """extract / findall write the spans' bytes through 64-row sub-tiles in LDS (k_spans_write_tile) when the column's
    sub-tiles fit the staging buffer: spans of 1 .. 90 bytes at every alignment, null and empty rows, several sub-tiles and
    a ragged last one; a column whose sub-tiles do not fit takes the thread-per-row kernel.  Both against the oracle,
    and against each other with the tile kernel switched off."""
    rnd = random.Random(4242)
    if shape == "short":
        rows = [" ".join("%d.%d" % (rnd.randrange(300), rnd.randrange(300)) for _ in range(rnd.randrange(4))) for _ in range(1000)]
    elif shape == "long_spans":
        rows = ["".join(rnd.choice("abcdefgh") * rnd.randrange(1, 45) + rnd.choice([" ", "  ", ""]) for _ in range(2))[:90] for _ in range(777)]
    else:
        rows = [("tok%d " % i) * 40 for i in range(200)]
    rows[5] = None
    rows[70] = ""
    rows[-1] = None
    for pat in (r"\w+", r"\d+\.\d+", r"[a-h]{17,}"):
        want = oracle_engine.findall(rows, pat)
        assert gpu_engine.findall(rows, pat) == want, pat
    for pat in (r"(\w+) +(\w+)", r"(\d+)\.(\d+)"):
        want = oracle_engine.extract(rows, pat)
        assert gpu_engine.extract(rows, pat) == want, pat
    monkeypatch.setenv("CS_SPANS_ROWWISE", "1")
    assert gpu_engine.findall(rows, r"\w+") == oracle_engine.findall(rows, r"\w+")


def test_gpu_offsets_from_lengths_beyond_32_bits(gpu_engine):
    """The chunked lengths -> offsets scan (cs_core.hip) runs a chunk's inner sums in 32 bits only when the chunk's total
    fits; 200 copies of a 16 MiB row put 3.1 GiB into one 2048-row chunk.  Rows gathered from both ends of the result
    must come back whole, and the scan must also be right at sizes around a chunk (2047 .. 2049 rows) and a workgroup."""
    from custrings_amd import nvstrings

    big = "a" * (1 << 24)
    s = nvstrings.to_device([big, "b", None, ""])
    g = s.gather([0] * 200 + [1, 2, 3, 0])
    assert g.size() == 204
    lens = np.zeros(204, dtype=np.int32)
    assert g.byte_count(lens.ctypes.data) == 201 * (1 << 24) + 1
    assert lens[:200].tolist() == [1 << 24] * 200 and lens[200:].tolist() == [1, -1, 0, 1 << 24]
    tail = g.sublist(199, 204).to_host()
    assert tail[0] == big and tail[1:4] == ["b", None, ""] and tail[4] == big
    for rows in (1, 2047, 2048, 2049, 8191, 8193, 70_001):
        src = ["x" * (i % 7) if i % 11 else None for i in range(rows)]
        c = nvstrings.to_device(src)
        idx = list(range(rows - 1, -1, -1))
        assert c.gather(idx).to_host() == src[::-1], rows


def test_gpu_extract_groups_resolved_backwards_edges(gpu_engine, oracle_engine):
    """The extract kernel reads a match's groups off a backward walk from the matching thread when the match is ASCII and
    at most 32 steps long (regex_tdfa.h: group_find_back) and runs the forward form otherwise, lane by lane: matches of
    31 / 32 / 33 / 60 bytes, matches that end with the row, a non-ASCII character inside or next to the match, groups
    that do not take part, nested and repeated groups, more than four groups (two batches)."""
    rows = fuzzdata.group_edge_rows()
    for pat in fuzzdata.GROUP_EDGE_PATTERNS:
        assert gpu_engine.extract(rows, pat) == oracle_engine.extract(rows, pat), pat



def test_gpu_replace_re_with_a_co_tenant_on_the_gpu():
    """The persistent replace kernel sizes its grid for an empty device.  With half the CUs held by another stream's
    kernel only half of its workgroups are resident at first -- and that is all it needs: tiles are
    drawn by ticket, so the resident workgroups take every tile; the others start when CUs come free, find the tickets
    drawn and leave.  No wait runs into its time bound (a quarter of a second without progress), nothing falls back to the
    two-pass kernels, and the call returns in a small multiple of its time alone -- long before the co-tenant (1.5 s) ends."""
    import ctypes as C
    import time

    import torch

    import cpulibs
    import gpuutil
    from custrings_amd import _lib

    orc = cpulibs.Oracle()
    rows = 2_000_000
    g, o = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    import engines

    pat = r"\d+\.\d+\.\d+\.\d+"
    blob = engines.reference_blob(pat)
    blob = np.ascontiguousarray(blob if blob is not None else engines.product_blob(pat), dtype=np.int32)
    want = orc.replace_re(o, blob, "<IP>")
    L = _lib.lib
    gpuutil.assert_same(g.replace(pat, "<IP>"), want, "replace_re alone (warm-up: kernel load, buffer pool)")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    alone = g.replace(pat, "<IP>")
    dt_alone = time.perf_counter() - t0
    del alone
    side = torch.cuda.Stream()
    before = int(L.cs_fallback_count())
    # 128 workgroups with 160 KB of LDS each: 128 of the 256 CUs taken for 1.5 s
    _lib.check(L.cs_debug_spin(128, 160 * 1024, 1500, C.c_void_p(side.cuda_stream)))
    time.sleep(0.05)
    t0 = time.perf_counter()
    got = g.replace(pat, "<IP>")
    dt = time.perf_counter() - t0
    assert int(L.cs_fallback_count()) == before, "the single pass gave up next to a co-tenant"
    assert gpuutil.lib().lib.cs_debug_last_route() == b"chain"
    gpuutil.assert_same(got, want, "replace_re next to a co-tenant")
    # (the proof is the route and the fallback count above; the wall clock only has to stay below the quarter-second no-progress
    # bound -- anything tighter is the box's scheduling, not the kernel's: ADVICE r05)
    assert dt < 0.2, (dt, dt_alone)
    torch.cuda.synchronize()
    # and alone again: the single pass, no fallback
    gpuutil.assert_same(g.replace(pat, "<IP>"), want, "replace_re alone")
    assert int(L.cs_fallback_count()) == before
