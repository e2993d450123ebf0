"""The product's per-row device logic (custrings_amd/csrc/row_ops.h, regex_vm.h,
regex_compile.cpp), compiled for the host by tests/rowemu, against the golden
vectors and differentially against the oracle on random columns.  CPU only; the
same functions run inside the HIP kernels (checked again by the -m gpu tests)."""
import pytest

import engines
import fuzzdata

REF = [c for c in engines.load_cases("reference_tests.json") if c["op"] not in ("category", "ngrams", "tokenize_ngrams")]
APX = [c for c in engines.load_cases("survey_appendix_a.json") if c["op"] not in ("category", "ngrams", "tokenize_ngrams")]


@pytest.mark.parametrize("case", REF + APX, ids=[c["id"] for c in REF + APX])
def test_rowemu_golden(emu_engine, case):
    assert engines.run_case(emu_engine, case) == case["expect"], case["src"]


PATTERNS = [r"\d+\.\d+\.\d+\.\d+", r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", r"a*", r"x*", r"a|aa", r"aa|a", r"a+?",
            r"\w+", r"\W", r"[\W]", r"\s+", r"^a", r"a$", r"\bc", r"c\b", r"\B", r"[a-c]+[x-z]?", r"[^a-c ]+", r"é+",
            r"[é-ü]", r"(a|b)*c", r".", r".*", r"^$", r"\Aa", r"z\Z", r"(ab|a)(bc|c)?", r"a{2,3}", r"(a|b|c){3}"]


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_rowemu_vs_oracle_string_ops(emu_engine, oracle_engine, seed):
    s = fuzzdata.rows(seed, 600)
    o, e = oracle_engine, emu_engine
    assert e.lower(s) == o.lower(s)
    assert e.upper(s) == o.upper(s)
    for ts in (None, " ", "ab ", "é ", "\n\t x"):
        for side in (0, 1, 2):
            assert e.strip(s, ts, side) == o.strip(s, ts, side)
    for sub in ("a", "é", "ab", " ", "", "bc", "😀", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"):
        for st, en in ((0, -1), (1, 5), (3, 2), (2, 100), (-3, -1)):
            assert e.find(s, sub, st, en) == o.find(s, sub, st, en), (sub, st, en)
        assert e.contains(s, sub) == o.contains(s, sub)
    for pat, repl in (("a", "xx"), ("é", ""), ("ab", "é"), (" ", "__"), ("aa", "a")):
        for n in (-1, 0, 1, 2):
            assert e.replace(s, pat, repl, n) == o.replace(s, pat, repl, n), (pat, repl, n)
    for d in (None, " ", "a", "é", "ab", "éa", ",", "aa"):
        for n in (-1, 0, 1, 2, 5):
            assert e.split(s, d, n) == o.split(s, d, n), (d, n)
    for d in (None, " ", "_-", "é ", "a\t"):
        assert e.tokenize(s, d) == o.tokenize(s, d)


@pytest.mark.parametrize("pat", PATTERNS)
def test_rowemu_vs_oracle_regex(emu_engine, oracle_engine, pat):
    s = fuzzdata.rows(11, 300, alphabet=list("aabbc xyz_.\n019") + ["é", "ü", "😀"]) + fuzzdata.log_rows(5, 300)
    o, e = oracle_engine, emu_engine
    assert e.contains_re(s, pat) == o.contains_re(s, pat)
    assert e.match(s, pat) == o.match(s, pat)
    assert e.count_re(s, pat) == o.count_re(s, pat)
    for n in (-1, 1, 2):
        assert e.replace_re(s, pat, "<é>", n) == o.replace_re(s, pat, "<é>", n), n
