"""Rows with embedded NUL bytes.  The reference's executor ends a call at a NUL (`while (c && ...)`, regexec.inl:443) -- but its
search for the next start, for a program whose FIRST instruction is a literal character, is custring_view::find (regexec.inl:220-232):
by length, over NUL bytes.  count_re('a') of "a\\0a" is 2, count_re('[a]') of it is 1.  The executors restate both (regex_tdfa.cpp:
step; regex_vm.h: the prefilter's skip; cs_runs.hip: nul_blind); found by the soak's columns of ASCII text with a few NUL bytes
against the oracle (round 6)."""
import pytest

ROWS = ["xa\x00a", "a\x00", "\x00a", "b\x00b a", "aéa", "é", "ab\x00", "\x00", "aa\x00aa\x00", "\x00\x00a\x00", "ab\x00ab", "a\x00b ab", "", None,
        "ab\x00" * 20, "x" * 70 + "\x00" + "ab" * 10, "\x00" * 5 + "a1 a2 a3",
        # (a first instruction `^`, multi-line: the jump to the byte behind the next line feed passes NUL bytes too -- regexec.inl:233-246)
        "x\x00\na", "\x00\na", "a\x00\na", "x\n\x00a", "xa\x00\nab\nab", "\n\x00\na", "ab\nab", "x\nab\n\nab x\x00y\nab", "\n"]
PATTERNS = ["a", "a+", "ab", r"a\d", "ab?", "a|b", "(a)b", "[ab]", r"\da", "a$", "^a", r"a\b", "[a]+", r"\w+", "b a", r"a\d a",
            "^ab", "^a|^b", r"^\w", r"^\w+$", r"^a\b", "^x|^a", r"\Aa"]


@pytest.mark.parametrize("engine", [0, 1])
def test_emulated_executors_on_rows_with_nul_bytes(emu_engine, oracle_engine, engine):
    e = emu_engine.e
    e.set_engine(engine)
    try:
        for pat in PATTERNS:
            assert emu_engine.contains_re(ROWS, pat) == oracle_engine.contains_re(ROWS, pat), pat
            assert emu_engine.match(ROWS, pat) == oracle_engine.match(ROWS, pat), pat
            assert emu_engine.count_re(ROWS, pat) == oracle_engine.count_re(ROWS, pat), pat
            for repl in ("<>", ""):
                assert emu_engine.replace_re(ROWS, pat, repl, -1) == oracle_engine.replace_re(ROWS, pat, repl, -1), (pat, repl)
            assert emu_engine.replace_re(ROWS, pat, "#", 1) == oracle_engine.replace_re(ROWS, pat, "#", 1), pat
    finally:
        e.set_engine(1)


def test_the_literal_first_jump_is_what_differs(oracle_engine):
    """(the oracle itself: the two behaviours side by side, so that a change of either shows)"""
    assert oracle_engine.count_re(["a\x00a"], "a")[0] == [2]
    assert oracle_engine.count_re(["a\x00a"], "[a]")[0] == [1]
    assert oracle_engine.contains_re(["\x00a"], "a")[0] == [True]
    assert oracle_engine.contains_re(["\x00a"], "[a]")[0] == [False]
