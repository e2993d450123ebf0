"""Round 5: column metadata as a by-product of the producing op (no `k_max_span*` pass for an op in a chain), the buffer
pool after cs_stream_forget, the bit-parallel regex route, full-size cross-checks of the headline ops between the
tile route and the independent row-wise route."""
import ctypes as C
import random

import numpy as np
import pytest

import cpulibs
import engines
import gpuutil

pytestmark = pytest.mark.gpu

IPV4 = r"\d+\.\d+\.\d+\.\d+"


def cached_meta(g):
    L = gpuutil.lib()
    out = (C.c_int64 * 4)()
    L.check(L.lib.cs_column_cached_meta(g.m_cptr, out))
    return list(out)


def measured_meta(g):
    """(largest 64-row span, longest row) from the exported offsets."""
    _, offs, _ = g._export64()
    offs = np.asarray(offs, dtype=np.int64)
    rows = len(offs) - 1
    if rows == 0:
        return 0, 0
    lens = np.diff(offs)
    starts = np.arange(0, rows, 64)
    ends = np.minimum(starts + 64, rows)
    return int((offs[ends] - offs[starts]).max()), int(lens.max())


@pytest.mark.parametrize("kind,rows", [(3, 70_001), (2, 50_000), (5, 30_000)])
def test_gpu_fresh_columns_are_sized_at_ingest(kind, rows):
    """A generated column (as an ingested one) leaves ingest with its longest row and largest 64-row span known -- the
    reference sizes its strings once at ingest (NVStringsImpl.cu:399-444) -- and the numbers are the measured ones."""
    g = gpuutil.synth(kind, 0, rows)
    span, longest, _, _ = cached_meta(g)
    assert (span, longest) == measured_meta(g)


def test_gpu_chained_ops_inherit_their_metadata():
    """Every producer hands the sizing numbers to its output, exact or as an upper bound: an op in a chain never pays a
    metadata pass for its input (VERDICT r4 missing #4).  Checked on the C2 chain, the headline ops and the regex
    column writers: the cached numbers exist and bound the measured ones; exact where a pass over lengths made them."""
    c2 = gpuutil.synth(2, 0, 60_000)
    low = c2.lower()
    st = low.strip()
    for name, col, exact in (("lower", low, True), ("strip", st, True)):
        span, longest, _, _ = cached_meta(col)
        mspan, mlong = measured_meta(col)
        assert span >= mspan and longest >= mlong and span >= 0 and longest >= 0, name
        if exact:
            assert (span, longest) == (mspan, mlong), name
    for col in st.split(" "):
        span, longest, _, _ = cached_meta(col)
        mspan, mlong = measured_meta(col)
        assert span >= mspan >= 0 and longest >= mlong >= 0
        assert span <= 64 * 96 and longest <= 96
    c3 = gpuutil.synth(3, 0, 80_000)
    r = c3.replace(IPV4, "<IP>")
    span, longest, _, _ = cached_meta(r)
    mspan, mlong = measured_meta(r)
    assert span == mspan and longest >= mlong >= 0  # (the span is the largest sub-tile total the kernel saw)
    grown = c3.replace(r"\d+", "<NUMBER>")
    span, longest, _, _ = cached_meta(grown)
    assert span == measured_meta(grown)[0] and longest == -1  # (a growing replacement: the longest row is not known)
    for col in c3.extract(r"(\d+)\.(\d+)\.\d+\.(\d+) "):
        span, longest, _, _ = cached_meta(col)
        assert span >= measured_meta(col)[0] and longest >= measured_meta(col)[1] and span >= 0


def test_gpu_metadata_bounds_do_not_change_results():
    """A derived column (bounds) and the same column re-ingested (measured numbers) give identical results downstream."""
    c3 = gpuutil.synth(3, 0, 50_000)
    cols = c3.split(" ")
    orc = cpulibs.Oracle()
    for k in (0, 1, 3):
        derived = cols[k]
        fresh = gpuutil.from_col(gpuutil.to_col(derived))
        a, b = derived.strip("/"), fresh.strip("/")
        gpuutil.assert_same(a, gpuutil.to_col(b), "strip of split column %d" % k)
        gpuutil.assert_same(derived.replace(r"\d+", "#"), gpuutil.to_col(fresh.replace(r"\d+", "#")), "replace_re of split column %d" % k)
        gpuutil.assert_same(a, orc.strip(gpuutil.to_col(derived), "/"), "strip vs oracle")


def test_gpu_pool_after_stream_forget():
    """run on stream A, cs_stream_forget(A), destroy A, continue on stream B.  Cached blocks that name
    A as their last user must be handed out without touching the destroyed handle."""
    from custrings_amd import _lib, nvstrings

    L = gpuutil.lib()
    hip = _lib.loaded_hip()
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    sa, sb = C.c_void_p(), C.c_void_p()
    assert hip.hipStreamCreate(C.byref(sa)) == 0 and hip.hipStreamCreate(C.byref(sb)) == 0
    rows = 40_000
    orc = cpulibs.Oracle()
    expect = orc.synth(3, 0, rows)

    def make(stream):
        out = C.c_void_p()
        L.check(L.lib.cs_synth_column(3, 0, rows, gpuutil.SEED, 0, stream, C.byref(out)))
        return out

    a = make(sa)
    up = C.c_void_p()
    L.check(L.lib.cs_upper(a, sa, C.byref(up)))
    assert hip.hipStreamSynchronize(sa) == 0
    L.lib.cs_column_destroy(up)  # its blocks go to the pool naming stream A
    L.lib.cs_column_destroy(a)
    L.check(L.lib.cs_stream_forget(sa))
    assert hip.hipStreamDestroy(sa) == 0
    # the same sizes again on B: the pool's blocks (last stream: the destroyed A) are re-used
    for _ in range(3):
        b = make(sb)
        up = C.c_void_p()
        L.check(L.lib.cs_upper(b, sb, C.byref(up)))
        assert hip.hipStreamSynchronize(sb) == 0
        gpuutil.assert_same(nvstrings.nvstrings(b.value), expect, "column made on B out of A's blocks")
        L.lib.cs_column_destroy(up)
    L.check(L.lib.cs_stream_forget(sb))
    assert hip.hipStreamDestroy(sb) == 0
    # and the default stream afterwards
    gpuutil.assert_same(gpuutil.synth(3, 0, rows), expect, "default stream after both were forgotten")


# ---- the bit-parallel regex route (regex_bits.h; VERDICT r4 missing #3) ------------------------------------------------
GTEST = r"(\bin\b)|(\ba\b)|(\bthe\b)"  # cpp/tests/test_replace.cpp:40
BITS_PATTERNS = [GTEST, r"[aeiou]+", r"\bthe\b", r"cat|cot|cut", r"colou?r", r"#\w+", r"^GET|^PUT", r"ing$", r"[^ ]+", r"a.c", r"(a|b)(c|d)", r"x[0-9][0-9]"]


def last_route():
    return gpuutil.lib().lib.cs_debug_last_route().decode()


def blob_of(pat):
    b = engines.reference_blob(pat)
    return np.ascontiguousarray(b if b is not None else engines.product_blob(pat), dtype=np.int32)


def check_regex_ops(g, o, pat, orc, want_route=None, repls=("=", "", "<LONGER>")):
    blob = blob_of(pat)
    re = gpuutil.compile_re(pat)
    L = gpuutil.lib()
    try:
        got, n = gpuutil.bools(g, "cs_contains_re", re)
        if want_route:
            assert last_route() == want_route, (pat, "contains_re", last_route())
        exp, en = orc.contains_re(o, blob)
        assert np.array_equal(got, exp) and n == en, (pat, "contains_re")
        got, n = gpuutil.bools(g, "cs_match_re", re)
        exp, en = orc.contains_re(o, blob, 1)
        assert np.array_equal(got, exp) and n == en, (pat, "match")
        cnt = np.zeros(max(g.size(), 1), dtype=np.int32)
        found = C.c_int64()
        L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
        if want_route:
            assert last_route() == want_route, (pat, "count_re", last_route())
        exp, en = orc.count_re(o, blob)
        assert np.array_equal(cnt[: g.size()], exp) and found.value == en, (pat, "count_re")
    finally:
        L.lib.cs_regex_destroy(re)
    for repl in repls:
        out = g.replace(pat, repl)
        if want_route:
            assert last_route() == want_route, (pat, "replace_re", repl, last_route())
        gpuutil.assert_same(out, orc.replace_re(o, blob, repl), "replace_re(%r, %r)" % (pat, repl))


@pytest.mark.parametrize("pat", BITS_PATTERNS)
def test_gpu_bits_route_on_c3(pat, monkeypatch):
    """contains_re / match / count_re / replace_re through the bit-parallel form on the C3 log lines (plain ASCII, rows of
    48-80 bytes) against the oracle; the route is asserted, and no single-pass launch may give up."""
    monkeypatch.setenv("CS_BITS_ALWAYS", "1")
    L = gpuutil.lib()
    orc = cpulibs.Oracle()
    rows = 70_001
    g, o = gpuutil.synth(3, 0, rows), orc.synth(3, 0, rows)
    f0 = L.lib.cs_fallback_count()
    check_regex_ops(g, o, pat, orc, want_route="bits")
    assert L.lib.cs_fallback_count() == f0


def test_gpu_bits_route_chosen_by_candidate_density():
    """Without the switch the route follows the column and the op (cs_regex.hip: bits_route): candidates in a good share of
    the sampled bytes -> the bit form; a rare first byte -> the automaton's skipping scans; chains keep the chain form."""
    L = gpuutil.lib()
    g = gpuutil.synth(3, 0, 50_000)
    res = np.zeros(g.size(), dtype=np.uint8)
    cnt = np.zeros(g.size(), dtype=np.int32)
    found = C.c_int64()

    def routes(pat):
        re = gpuutil.compile_re(pat)
        try:
            L.check(L.lib.cs_contains_re(g.m_cptr, re, res.ctypes.data, 0, None, C.byref(found)))
            c = last_route()
            L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
            n = last_route()
        finally:
            L.lib.cs_regex_destroy(re)
        g.replace(pat, "=")
        return c, n, last_route()

    assert routes(GTEST) == ("bits", "bits", "bits")
    assert routes(r"[aeiou]+") == ("bits", "bits", "bits")
    assert routes(r"[^ ]+") == ("plain", "bits", "bits")  # (contains_re finds a match at once: the first-match scan)
    assert routes(r"cat|cot|cut") == ("bits", "bits", "bits")  # (few candidates, three live threads)
    assert "bits" not in routes(r"#\w+")  # ('#' does not occur: the skipping scans never stop)
    c, n, r = routes(r"^GET|^PUT")
    assert "bits" not in (c, n, r)
    assert routes(IPV4)[2] == "chain"


def test_gpu_bits_route_meets_other_tiles(monkeypatch):
    """Sub-tiles the bit form does not take -- a byte >= 0x80, a NUL byte, a row beyond 95 bytes -- go to the generic scan
    inside the same launch (rows placed outside the column's sampled windows); empty and null rows; row-count edges."""
    monkeypatch.setenv("CS_BITS_ALWAYS", "1")
    orc = cpulibs.Oracle()
    base = orc.synth(3, 0, 40_000)
    rows = base.to_list()
    # (the host gives a column to the 96-bit-mask forms while its longest row has at most 93 bytes: cs_regex.hip, choose_tile)
    specials = ["naïve in a the", "in\x00a the", "a" * 93, "é", "", None, "in a the", "x" * 92 + "a", "a " * 46 + "a", "in" + " " * 91, "the " * 23,
                "the end in a\n", "\nthe", "ünder the a", "a\x00", "in a the " * 10]
    for k, sp in enumerate(specials):
        rows[9_000 + 613 * k] = sp
        rows[31_000 + 311 * k] = sp
    o = cpulibs.Col.from_list(rows)
    g = gpuutil.from_col(o)
    for pat in (GTEST, r"[aeiou]+", r"ing$|^in", r"#\w+"):
        check_regex_ops(g, o, pat, orc, want_route="bits", repls=("=", "<LONGER>"))
    # rows beyond 93 bytes move the whole column to the long-row forms of the automaton kernels (the bit form's masks hold
    # 96 bits): same answers, another route
    for k, sp in enumerate(["the " * 30 + "a in", "in a the " * 12, "a" * 200, "a" * 96, "x" * 94 + "a", "a " * 47 + "a", "the " * 24]):
        rows[15_000 + 97 * k] = sp
    o = cpulibs.Col.from_list(rows)
    g = gpuutil.from_col(o)
    for pat in (GTEST, r"[aeiou]+"):
        check_regex_ops(g, o, pat, orc, repls=("=",))
    for n in (1, 63, 64, 65, 127, 129, 4097):
        sub = cpulibs.Col.from_list(rows[9_000 : 9_000 + n])
        check_regex_ops(gpuutil.from_col(sub), sub, GTEST, orc, repls=("=",))


# ---- counted items and `\\b` on the chain form (regex_tdfa.h: chain_item) ------------------------------------
IPV4B = r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b"


def test_gpu_counted_chain_with_its_tables_in_memory():
    """The 26-instruction dotted quad takes the chain arithmetic (`{1,3}` = a run of at most three, `\\b` = a byte test per
    match), and its 19 KB of DFA tables stay in memory so that a third workgroup fits a CU (cs_regex.hip: chain_global): the
    sub-tiles the arithmetic does not take -- a byte >= 0x80, a NUL, placed outside the column's sampled windows -- walk the tables THERE, inside the same launch.  Every regex op against the oracle."""
    orc = cpulibs.Oracle()
    L = gpuutil.lib()
    base = orc.synth(3, 0, 40_000)
    rows = base.to_list()
    specials = ["é 10.2.3.4", "10.2.3.4\x001.2.3.4 5.6.7.8", "1.2.3.4" + "x" * 79 + "5.6.7.8", "a1.2.3.4", "1.2.3.4a", "a1.2.3.4.5", "1234.5.6.7", "1.2.3.4567",
                "1.2.3.4.5.6.7.8", "255.255.255.255 0.0.0.0", "1.2.3.4_5.6.7.8", "", None, "ü", "1.2.3.4 ü 1.2.3.4x 1.2.3.4", "9" * 93, "1." * 46, "0.0.0.0" * 13]
    for k, sp in enumerate(specials):
        rows[9_000 + 613 * k] = sp
        rows[31_000 + 311 * k] = sp
    o = cpulibs.Col.from_list(rows)
    g = gpuutil.from_col(o)
    f0 = L.lib.cs_fallback_count()
    # (the third: counted items, a literal suffix and the tables in memory -- the suffix and the counts come from the image's
    # last words, staged in LDS beside the header: cs_regex.hip, tsetup)
    for pat in (IPV4B, r"\b\d+\.\d+\b", r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3} -", r"\b(\d{1,3})\.(\d{1,3})\b", r"\d+\.\d{2}\.\d+"):
        check_regex_ops(g, o, pat, orc, repls=("<IP>", "", "<a-much-longer-one>"))
        g.replace(pat, "#")
        assert last_route() == "chain", (pat, last_route())
        got, exp = g.findall(pat), orc.findall(o, blob_of(pat))
        assert len(got) == len(exp), pat
        for k, (gc, ec) in enumerate(zip(got, exp)):
            gpuutil.assert_same(gc, ec, "findall(%r) column %d" % (pat, k))
    assert L.lib.cs_fallback_count() == f0
    # the switch that keeps the tables in LDS: the same answers
    L.check(L.lib.cs_config_set(b"CS_CHAIN_TABLES_IN_LDS", b"1"))
    try:
        check_regex_ops(g, o, IPV4B, orc, repls=("<IP>",))
    finally:
        L.check(L.lib.cs_config_set(b"CS_CHAIN_TABLES_IN_LDS", None))
    for n in (1, 63, 64, 65, 129, 4097):
        sub = cpulibs.Col.from_list(rows[9_000 : 9_000 + n])
        check_regex_ops(gpuutil.from_col(sub), sub, IPV4B, orc, repls=("#",))


BREFS_CHAIN_CASES = [(r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"), (r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"<\1>"), (r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"[\0|\2\2]--a-longer-literal-part--"),
                     (r"(\d+)\.(\d+)", r"\2"), (r"(\d+)\.(\d+)", r""), (r"(\d+)\.(\d+)", r"\9x\3"), (r"((\d+)\.(\d+))\.\d+", r"\1=\3=\2"),
                     (IPV4B.replace(r"\d{1,3}", r"(\d{1,3})"), r"\4.\3.\2.\1"), (r"\b(\d{1,3})\.(\d{1,3})\b", r"\2\2\2\1"), (r"(\d+)\.(\d+)\.\d+\.(\d+) ", r"\3-\1 "),
                     (r"([a-z]+)=", r"=\1\1")]


@pytest.mark.parametrize("pat,repl", BREFS_CHAIN_CASES, ids=["%s->%s" % (p[:18], r[:10]) for p, r in BREFS_CHAIN_CASES])
def test_gpu_backrefs_chain_form(pat, repl):
    """replace_with_backrefs on a chain whose groups are runs of items (k_tdfa_replace_stream<.., BREFS, .., CHAIN>: no table,
    no group tags in LDS; the out tile sized from the template -- cs_regex.hip: brefs_grow): templates that shrink, keep and
    grow a match, a group named twice (bounded and not), group 0, groups the pattern does not have, nested groups, counted
    items with `\\b`, a literal suffix -- against the oracle, on the form itself (no fallback)."""
    orc = cpulibs.Oracle()
    L = gpuutil.lib()
    rows = 60_000
    g, o = gpuutil.synth(3, 1_000_000, rows), orc.synth(3, 1_000_000, rows)
    f0 = L.lib.cs_fallback_count()
    got = g.replace_with_backrefs(pat, repl)
    assert last_route() == "brefs-chain", (pat, repl, last_route())
    assert L.lib.cs_fallback_count() == f0, (pat, repl)
    gpuutil.assert_same(got, orc.replace_with_backrefs(o, blob_of(pat), repl), "replace_with_backrefs(%r, %r)" % (pat, repl))


def test_gpu_backrefs_chain_form_on_dense_matches():
    """Sub-tiles with more matches than the record table holds (128) stay on the chain form -- the group ranges are derived once
    more at the assembly -- and a template that outgrows the first attempt's room (a quarter of the input) is repeated with the
    worst-case sizing inside the same call: rows of short numbers, `(\\d+)` -> `<\\1>` triples them.  No fallback, the oracle's rows."""
    orc = cpulibs.Oracle()
    L = gpuutil.lib()
    rnd = random.Random(77)
    rows = [" ".join(str(rnd.randrange(1000)) for _ in range(rnd.randrange(1, 16))) for _ in range(20_000)]      # ~8 matches a row
    rows += ["1 2 3 4 5 6 7 8 9 0 1 2 3 4 5 6 7 8 9 0 1 2 3 4 5 6 7 8 9 0 1 2 3 4 5 6 7 8 9 0"] * 300                  # 40 matches a row: 2 560 a sub-tile
    rows += orc.synth(3, 0, 20_000).to_list()                                                                  # log lines among them
    rows += ["", None, "x", "7", "no digits here"] * 10
    o = cpulibs.Col.from_list(rows)
    g = gpuutil.from_col(o)
    for pat, repl in ((r"(\d+)", r"<\1>"), (r"(\d+)", r"\1\1"), (r"(\d+)", r""), (r"(\d+) (\d+)", r"\2 \1")):
        f0 = L.lib.cs_fallback_count()
        got = g.replace_with_backrefs(pat, repl)
        assert last_route() == "brefs-chain", (pat, repl, last_route())
        assert L.lib.cs_fallback_count() == f0, (pat, repl)
        gpuutil.assert_same(got, orc.replace_with_backrefs(o, blob_of(pat), repl), "replace_with_backrefs(%r, %r)" % (pat, repl))
    # (no chain -- a single digit is not a run: the backrefs form proper gives such sub-tiles up and the two-pass form answers)
    gpuutil.assert_same(g.replace_with_backrefs(r"(\d)", r"[\1]"), orc.replace_with_backrefs(o, blob_of(r"(\d)"), r"[\1]"), "replace_with_backrefs((\\d))")


def test_gpu_long_replacement_of_short_matches():
    """A replacement longer than four times the shortest match rides the stream kernel when the sample says matches are few
    (cs_regex.hip: few_matches) -- `-` -> 35 bytes on log lines -- and a column that holds many after all (rows of dashes outside
    the sampled windows) is answered by the kernels behind it: the oracle's rows either way."""
    orc = cpulibs.Oracle()
    L = gpuutil.lib()
    repl = "<a-replacement-of-thirty-five-bytes>"
    base = orc.synth(3, 0, 40_000)
    g = gpuutil.from_col(base)
    f0 = L.lib.cs_fallback_count()
    for pat in (r"-", r"\[", r"GET"):
        gpuutil.assert_same(g.replace(pat, repl), orc.replace_re(base, blob_of(pat), repl), "replace_re(%r, 35 bytes)" % pat)
        assert last_route() != "", pat  # (a stream launch)
    assert L.lib.cs_fallback_count() == f0
    rows = base.to_list()
    for k in range(64):
        rows[17_000 + k] = "-" * 60 + " - - -"
    o = cpulibs.Col.from_list(rows)
    gpuutil.assert_same(gpuutil.from_col(o).replace(r"-", repl), orc.replace_re(o, blob_of(r"-"), repl), "replace_re('-', 35 bytes), a sub-tile of dashes")


def test_gpu_backrefs_chain_form_gives_up_like_the_backrefs_form():
    """A sub-tile the marker arithmetic does not take (a byte >= 0x80 or a NUL outside the sampled windows), a sub-tile with
    more than 128 matches: the launch is given up and the two-pass form answers -- the same rows as the oracle's."""
    orc = cpulibs.Oracle()
    base = orc.synth(3, 0, 30_000)
    rows = base.to_list()
    pat, repl = r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"
    for special in ("é 10.2.3.4 1.2.3.4", "1.2.3.4\x005.6.7.8", "1.2.3.4 " * 11):
        r2 = list(rows)
        for k in range(64 if special.startswith("1.2.3.4 1") else 1):
            r2[11_003 + k] = special
        o = cpulibs.Col.from_list(r2)
        g = gpuutil.from_col(o)
        gpuutil.assert_same(g.replace_with_backrefs(pat, repl), orc.replace_with_backrefs(o, blob_of(pat), repl), repr(special[:12]))


# ---- full-size parity of the headline ops ----------------------------------------------------------
def test_gpu_full_size_headline_routes_agree():
    """split(' ') and replace_re(IPv4, '<IP>') on the FULL 100M-row C3 column by two independent implementations each -- the
    tile kernels (measure + emit4; the single-pass stream kernel with the chain arithmetic) and the first-generation
    kernels (a thread per row: CS_SPLIT_GENERIC; the two-pass size / write kernels on the tagged DFA: CS_REGEX_TWO_PASS) --
    must give the same digest (offsets, bytes, validity of every row: include/cs_synth_spec.h) for every output column."""
    L = gpuutil.lib()
    g = gpuutil.synth(3, 0, 100_000_000)
    f0 = L.lib.cs_fallback_count()
    fast_split = [c.digest() for c in g.split(" ")]
    fast_rep = g.replace(IPV4, "<IP>")
    assert last_route() == "chain"
    fast_rep = fast_rep.digest()
    assert L.lib.cs_fallback_count() == f0
    for name in ("CS_SPLIT_GENERIC", "CS_REGEX_TWO_PASS"):
        L.check(L.lib.cs_config_set(name.encode(), b"1"))
    try:
        slow_split = [c.digest() for c in g.split(" ")]
        slow_rep = g.replace(IPV4, "<IP>")
        assert last_route() == ""  # (no stream launch)
        slow_rep = slow_rep.digest()
    finally:
        for name in ("CS_SPLIT_GENERIC", "CS_REGEX_TWO_PASS"):
            L.check(L.lib.cs_config_set(name.encode(), None))
    assert len(fast_split) == 20 and fast_split == slow_split
    assert fast_rep == slow_rep


def test_gpu_full_size_round5_forms_agree_with_the_two_pass_kernels():
    """The forms added late in round 5, on the FULL 100M-row C3 column, each against an independent implementation by digest:
    replace_re of the counted dotted quad with `\\b` (the counted chain arithmetic on three words, tables in memory) and
    replace_with_backrefs (the backrefs chain form, out tile sized from the template) against the two-pass size / write kernels on
    the tagged DFA; contains_re / count_re of the gtest pattern (bit form, tables in memory) against the row-wise scan."""
    L = gpuutil.lib()
    g = gpuutil.synth(3, 0, 100_000_000)
    quad, tmpl = r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"
    f0 = L.lib.cs_fallback_count()
    fast = g.replace(IPV4B, "<IP>")
    assert last_route() == "chain"
    fast = fast.digest()
    fastb = g.replace_with_backrefs(quad, tmpl)
    assert last_route() == "brefs-chain"
    fastb = fastb.digest()
    res = np.zeros(g.size(), dtype=np.uint8)
    cnt = np.zeros(g.size(), dtype=np.int32)
    found = C.c_int64()
    re = gpuutil.compile_re(GTEST)
    try:
        L.check(L.lib.cs_contains_re(g.m_cptr, re, res.ctypes.data, 0, None, C.byref(found)))
        assert last_route() == "bits"
        fast_found = found.value
        L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
        assert last_route() == "bits"
        fast_hits, fast_sum, fast_res = found.value, int(cnt.sum(dtype=np.int64)), res.copy()
        assert L.lib.cs_fallback_count() == f0
        for name in ("CS_REGEX_TWO_PASS", "CS_BACKREFS_TWO_PASS", "CS_REGEX_ROWWISE"):
            L.check(L.lib.cs_config_set(name.encode(), b"1"))
        try:
            slow = g.replace(IPV4B, "<IP>")
            assert last_route() == ""
            slow = slow.digest()
            slowb = g.replace_with_backrefs(quad, tmpl).digest()
            L.check(L.lib.cs_contains_re(g.m_cptr, re, res.ctypes.data, 0, None, C.byref(found)))
            assert last_route() == "" and found.value == fast_found and np.array_equal(res, fast_res)
            L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
            assert found.value == fast_hits and int(cnt.sum(dtype=np.int64)) == fast_sum
        finally:
            for name in ("CS_REGEX_TWO_PASS", "CS_BACKREFS_TWO_PASS", "CS_REGEX_ROWWISE"):
                L.check(L.lib.cs_config_set(name.encode(), None))
    finally:
        L.lib.cs_regex_destroy(re)
    assert fast == slow and fastb == slowb


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _all_null_digest(rows):
    """digest of a column of `rows` null rows (cs_synth_spec.h: cs_digest_row with valid == 0)"""
    with np.errstate(over="ignore"):
        r = np.arange(1, rows + 1, dtype=np.uint64)
        return int(_mix64(np.uint64(0x6C6C756E5F5F5F5F) + _mix64(r)).sum(dtype=np.uint64))


def test_gpu_full_size_headline_vs_oracle_10m_rows():
    """The oracle on TEN MILLION rows of the 100M-row column (a tenth of it, in windows spread over its whole length; round 4
    compared 0.15 %): one oracle worker process per usable core (tests/cpu_digest_worker.py) runs split(' ') and
    replace_re(IPv4, '<IP>') on its windows and reports the digest of every output column; the same windows of the GPU's
    100M-row results must give the same digests.  Scaled down when the box has few usable cores (the test states what it
    covered)."""
    import json
    import os
    import subprocess
    import sys
    import tempfile

    sys.path.insert(0, cpulibs.ROOT)
    import bench

    cores = max(1, min(16, bench.effective_cpus()))
    total, win = 100_000_000, 125_000
    nwin = 80 if cores >= 8 else 10 * cores  # 10M rows on a GPU box's host; about 1.2M rows a core otherwise
    firsts = [int(i * (total - win) / (nwin - 1)) // 64 * 64 for i in range(nwin)]
    firsts[-1] = total - win  # (the column's last rows too)
    g = gpuutil.synth(3, 0, total)
    cols = g.split(" ")
    rep = g.replace(IPV4, "<IP>")
    tmpd = tempfile.mkdtemp(prefix="cs_digest_")
    prog = os.path.join(tmpd, "ipv4.npy")
    np.save(prog, blob_of(IPV4))
    # (the forms added late in round 5 ride along: the counted dotted quad with `\\b`, replace_with_backrefs on the chain form)
    quad, tmpl = r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"
    more = {"replace_b": g.replace(IPV4B, "<IP>"), "backrefs": g.replace_with_backrefs(quad, tmpl)}
    assert last_route() == "brefs-chain"
    np.save(os.path.join(tmpd, "ipv4b.npy"), blob_of(IPV4B))
    np.save(os.path.join(tmpd, "quad.npy"), blob_of(quad))
    with open(os.path.join(tmpd, "extra.json"), "w") as f:
        json.dump([["replace_b", "replace", os.path.join(tmpd, "ipv4b.npy"), "<IP>"], ["backrefs", "backrefs", os.path.join(tmpd, "quad.npy"), tmpl]], f)
    worker = os.path.join(cpulibs.ROOT, "tests", "cpu_digest_worker.py")
    env = dict(os.environ, CS_CPULIBS_PREBUILT="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", CS_DIGEST_EXTRA=os.path.join(tmpd, "extra.json"))
    procs = [subprocess.Popen([sys.executable, worker, prog, str(win)] + [str(f) for f in firsts[i::cores]], stdout=subprocess.PIPE, text=True, env=env)
             for i in range(cores) if firsts[i::cores]]
    null_digest = _all_null_digest(win)
    checked = 0
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0
        for line in out.splitlines():
            w = json.loads(line)
            first = w["first"]
            assert len(w["split"]) <= len(cols)
            for k, c in enumerate(cols):
                want = w["split"][k] if k < len(w["split"]) else null_digest
                assert c.sublist(first, first + win).digest() == want, ("split column", k, "rows from", first)
            assert rep.sublist(first, first + win).digest() == w["replace"], ("replace_re, rows from", first)
            for name, col in more.items():
                assert col.sublist(first, first + win).digest() == w[name], (name, "rows from", first)
            checked += win
    assert checked == nwin * win
    print("oracle cross-check: %d rows of the 100M-row column in %d windows on %d cores" % (checked, nwin, cores))


# ---- category: the key inside the slot (cs_category.hip: 32-byte slots) ------------------------------
@pytest.mark.parametrize("plain_slots", [False, True])
def test_gpu_category_keys_inside_the_slots(plain_slots, monkeypatch):
    """Keys of 0..40 bytes around the 21 bytes a slot holds -- equal up to byte 21 and different behind it, prefixes of each
    other, empty, null, multi-byte characters, a NUL byte inside -- against the oracle's sorted unique keys and codes; the
    same with the slots without key words (CS_CAT_PLAIN_SLOTS: the comparison through the chars, as large tables do)."""
    import random

    from custrings_amd import nvcategory

    if plain_slots:
        monkeypatch.setenv("CS_CAT_PLAIN_SLOTS", "1")
    rnd = random.Random(17)
    stems = ["", "a", "ab", "x" * 20, "x" * 21, "x" * 22, "y" * 21 + "1", "y" * 21 + "2", "y" * 21, "é" * 10, "é" * 11, "k\x00ey", "k\x00ez",
             "0123456789abcdefghij", "0123456789abcdefghijk", "0123456789abcdefghijkl", "0123456789abcdefghijkm", "z" * 40, "z" * 39 + "y"]
    pool = stems + [rnd.choice(stems)[: rnd.randint(0, 25)] + rnd.choice(["", "q", "qq", "7"]) for _ in range(400)]
    rows = [rnd.choice(pool) if rnd.random() > 0.01 else None for _ in range(200_000)]
    o = cpulibs.Col.from_list(rows)
    orc = cpulibs.Oracle()
    ok, ov = orc.category(o)
    cat = nvcategory.from_strings(gpuutil.from_col(o))
    gpuutil.assert_same(cat.keys(), ok, "keys")
    assert cat.values() == ov.tolist()
    # C4 rows (16 bytes exactly: the whole key in the slot) with a few thousand keys
    g, o4 = gpuutil.synth(4, 0, 300_000, param=5000), orc.synth(4, 0, 300_000, param=5000)
    ok, ov = orc.category(o4)
    cat = nvcategory.from_strings(g)
    gpuutil.assert_same(cat.keys(), ok, "C4 keys")
    assert cat.values() == ov.tolist()


def test_gpu_compare_with_the_empty_string():
    """compare('') used to come back as "equal" for every row (the untouched, zero-initialised buffer)."""
    from custrings_amd import nvstrings

    g = nvstrings.to_device(["abc", "", None, "é"])
    assert g.compare("") == [1, 0, None, 1]
    assert g.compare("abc") == [0, -1, None, g.compare("abc")[3]] and g.compare("abc")[3] > 0
    try:
        import pyniNVStrings  # noqa: F401  (the rebuilt CPython glue answers the same)
    except ImportError:
        return


# ---- one character class, once or in a `+` loop: byte-parallel compaction on rows of any length (cs_runs.hip) -----------
@pytest.mark.parametrize("pat", [r"[aeiou]+", r"[^ ]+", r".", r".+", r"e", r"[a-e]", r"[^a-z ]+", r" +", r"[0-9.]+",
                                 # classes with builtins: a non-ASCII character is a member through the unicode flags (tiles with such bytes row by row)
                                 r"\w+", r"\W+", r"\S+", r"[\w.]+", r"[^\w]", r"\d", r"[^\d ]+", r"\s", r"[\W\d]+", r"\w"])
def test_gpu_class_runs_vs_oracle(pat, monkeypatch):
    """replace_re of a single-class pattern through the byte-parallel kernels -- C5's rows of 40-150 bytes with non-ASCII
    characters and nulls, then rows chosen to sit on tile and piece boundaries -- against the oracle, route asserted."""
    monkeypatch.setenv("CS_CLASS_RUNS_ALWAYS", "1")
    orc = cpulibs.Oracle()
    blob = blob_of(pat)
    rows = 50_000
    g, o = gpuutil.synth(5, 0, rows), orc.synth(5, 0, rows)
    for repl in ("*", "", "<sixteen bytes!>"):
        out = g.replace(pat, repl)
        # (a pattern that replaces every byte by sixteen outgrows the out tile: the route declines, the answer is the same)
        assert last_route() == "runs" or len(repl) == 16, (pat, repl, last_route())
        gpuutil.assert_same(out, orc.replace_re(o, blob, repl), "replace_re(%r, %r) on C5" % (pat, repl))
    special = ["", None, "a", "aeiou", "xaeioux", " ", "  ", "a e i", "é", "aéa", "naïve café", "e" * 300, "b" * 300, "ae" * 700 + "x", "x" * 15 + "a", "x" * 16 + "a",
               "a" + "x" * 15, "\n", "a\nb", "1.2.3.4", ". . .", "\x00a\x00", "日本語 テキスト aeiou", "a" * 1023, "a" * 1024, "a" * 1025 + " b",
               # what the unicode flags decide: accented letters, digits of other scripts, spaces of other kinds, an emoji (beyond the
               # flags table), '_' and the newline
               "señor_1 ½ ٣٤ ３ 　x\u00a0y\u2003z", "x😀y 😀😀 _a_", "é" * 40 + " " + "ü" * 40, "a_b\nc_d", "Ωμέγα-3", "\u0660\u0661 ۲۳", "x" * 15 + "é" + "y" * 15 + "日"]
    rnd = np.random.default_rng(7)
    more = ["".join(rnd.choice(list("aeiou xyz.é1_") + ["日", "😀", "\u00a0", "٣"], size=int(rnd.integers(0, 180)))) for _ in range(3000)]
    col = cpulibs.Col.from_list(special * 3 + more + special)
    gc = gpuutil.from_col(col)
    for repl in ("#", "", "=+="):
        out = gc.replace(pat, repl)
        assert last_route() == "runs", (pat, repl)
        gpuutil.assert_same(out, orc.replace_re(col, blob, repl), "replace_re(%r, %r) on edge rows" % (pat, repl))
    # count_re: the same size pass counting matches
    L = gpuutil.lib()
    re = gpuutil.compile_re(pat)
    try:
        for gg, oo, what in ((g, o, "C5"), (gc, col, "edge rows")):
            cnt = np.zeros(max(gg.size(), 1), dtype=np.int32)
            found = C.c_int64()
            L.check(L.lib.cs_count_re(gg.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
            assert last_route() == "runs", (pat, what)
            exp, en = orc.count_re(oo, blob)
            assert np.array_equal(cnt[: gg.size()], exp) and found.value == en, (pat, "count_re", what)
    finally:
        L.lib.cs_regex_destroy(re)


@pytest.mark.parametrize("pat", [r"[a-c]+", r".", r"[^ ]+", r"\w+", r"[^\w]", r"b"])
def test_gpu_class_runs_on_arbitrary_bytes(pat, monkeypatch):
    """Malformed UTF-8: the executor takes a character's width from its lead byte and swallows what follows -- an ASCII member
    behind a lead without its continuation bytes is no match, a stray continuation byte is a character of its own.  The
    byte-parallel route answers byte by byte only on tiles whose bytes >= 0x80 form whole characters (cs_runs.hip: the
    announced continuation bytes) and decodes the others row by row: replace_re and count_re on rows of random bytes against
    the automaton kernels (the route off).  (Found by the soak: count_re([a-c]+) on `x\\xa9\\xe2b\\xf0BAb6`.)"""
    monkeypatch.setenv("CS_CLASS_RUNS_ALWAYS", "1")
    orc = cpulibs.Oracle()
    L = gpuutil.lib()
    rng = np.random.default_rng(20108)
    pool = np.array(list(b"ab1.2 3.4.5.6 x9_\t\nABc") + [0, 0xC3, 0xA9, 0xE2, 0x82, 0xAC, 0xFF, 0x80, 0x1F, 0xF0, 0x9F], dtype=np.uint8)
    lens = np.concatenate([rng.integers(0, 250, 6000), (rng.pareto(1.5, 300) * 20).astype(np.int64) % 3000, [0, 1, 2, 15, 16, 17, 1023, 1024, 1025]])
    offs = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    col = cpulibs.Col(pool[rng.integers(0, len(pool), int(offs[-1]))], offs, None)
    rows = [b"x\xa9\xe2b\xf0BAb6..\x00\xac.6x", b"\xe2.b\x0031.4\x82", b"\xe2ab", b"ab\xe2", b"ab\xf0\x9f", b"\x80a\x80b", b"\xc3\xa9a\xc3", b"a\xffb\xffc", b"\xf0abcd", b"\xf0\x9f\x98\x80ab"]
    extra = cpulibs.Col(np.frombuffer(b"".join(rows), dtype=np.uint8).copy(), np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64), None)
    blob = blob_of(pat)
    re = gpuutil.compile_re(pat)
    try:
        for c in (extra, col):
            g = gpuutil.from_col(c)
            got = g.replace(pat, "#")
            assert last_route() == "runs", pat
            cnt = np.zeros(c.rows, dtype=np.int32)
            found = C.c_int64()
            L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
            assert last_route() == "runs", pat
            L.check(L.lib.cs_config_set(b"CS_CLASS_RUNS_ALWAYS", None))
            L.check(L.lib.cs_config_set(b"CS_NO_CLASS_RUNS", b"1"))
            try:
                want = g.replace(pat, "#")
                assert last_route() != "runs"
                cnt2 = np.zeros(c.rows, dtype=np.int32)
                L.check(L.lib.cs_count_re(g.m_cptr, re, cnt2.ctypes.data, 0, None, C.byref(found)))
            finally:
                L.check(L.lib.cs_config_set(b"CS_NO_CLASS_RUNS", None))
                L.check(L.lib.cs_config_set(b"CS_CLASS_RUNS_ALWAYS", b"1"))
            # (against the product's automaton kernels, as every test on malformed input: the reference's contract is valid UTF-8,
            # where the oracle is the witness -- test_gpu_class_runs_vs_oracle)
            gpuutil.assert_same(got, gpuutil.to_col(want), "replace_re(%r) on arbitrary bytes: runs against the automaton" % pat)
            assert np.array_equal(cnt, cnt2), (pat, "count_re: runs against the automaton", np.flatnonzero(cnt != cnt2)[:5])
    finally:
        L.lib.cs_regex_destroy(re)


def test_gpu_class_runs_route_is_for_long_or_non_ascii_columns():
    """The route is taken where the 96-bit-mask forms are not: C5 (long rows) for the patterns that would take the chain
    arithmetic on the column's pieces or that no piece can hold (a class with white space in it); since round 6 the other
    single classes go to the pieces' bit form, the few rows with bytes >= 0x80 left holes (cs_regex.hip: pieces_for); C3 keeps
    the single pass."""
    g5 = gpuutil.synth(5, 0, 40_000)
    g5.replace(r"[a-z]+", "*")  # (a chain)
    assert last_route() == "runs"
    g5.replace(r"[^ ]+", "*")  # (tabs and line feeds are members: white space is no safe cut)
    assert last_route() == "runs"
    g5.replace(r"[aeiou]+", "*")
    assert last_route() == "pieces:bits+later"
    g5.replace(r"#+", "*")  # (no candidates in the sample: the skipping scans)
    assert last_route() != "runs"
    g5.replace(r"\w+", "*")  # (a builtin class: cs_runs.hip would take every tile with a byte >= 0x80 row by row)
    assert last_route() == "pieces:bits+later"
    g3 = gpuutil.synth(3, 0, 40_000)
    g3.replace(r"[aeiou]+", "*")
    assert last_route() == "bits"
