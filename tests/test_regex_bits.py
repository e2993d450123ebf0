"""The bit-parallel regex form (custrings_amd/csrc/regex_bits.h / regex_bits.cpp): which programs convert, and -- on the
host emulation of exactly what the stream kernels run (table classification of 16-byte pieces into one bitmap per class,
the row's masks cut out of the bitmaps, mask arithmetic) -- contains_re / match / count_re / replace_re against the
oracle's restatement of regexec.inl:204-442.  The reference's own dense-candidate pattern is
cpp/tests/test_replace.cpp:40: (\\bin\\b)|(\\ba\\b)|(\\bthe\\b)."""
import random

import pytest

import fuzzdata

GTEST = r"(\bin\b)|(\ba\b)|(\bthe\b)"

# pattern -> (classes, alternatives) of its bit form, or None when the program does not convert
FORMS = {
    GTEST: (7, 3),
    r"[aeiou]+": (1, 1),
    r"\bthe\b": (4, 1),
    r"cat|dog|bird": (10, 3) if False else None,  # (ten distinct letters: more than eight classes)
    r"cat|cot|cut": (5, 3),
    r"colou?r": (5, 2),
    r"ab?c?d": (4, 4),
    r"#\w+": (2, 1),
    r"x[0-9][0-9]": (2, 1),
    r"^GET|^PUT": (6, 2),
    r"ing$": (4, 1),
    r"\Aab": (2, 1),
    r"a.c": (3, 1),
    r"[^ ]+": (1, 1),
    r"\Bin\B": (3, 1),
    r"(a|b)(c|d)": (4, 4),
    r"a*": None, r"(ab)+": None, r"a+b": None, r"\d+\.\d+": None, r"a+?": None, r"": None, r"\b": None, r"a{2,3}x": (2, 2),
    r"(a|b)*c": None, r"[a-c]+[x-z]": None, r"é": None if False else (1, 1),
    # assertions behind the trailing `+` loop (the TAIL: the loop's exits are tried longest first)
    # (`$` reads the newline's class, `\b` the word characters': `_` is a `\w` and not one of them)
    r"[^ ]+$": (2, 1), r"\w+\b": (2, 1), r"[a-z]+\b": (2, 1), r"#\w+$": (3, 1), r"[a-e]+\B": (2, 1), r"\bth[a-z]+\b$": (5, 1), r"x[0-9]+\Z": (2, 1),
    r"[^ ]+$x": None,
}


def test_which_programs_convert(emu_engine):
    e = emu_engine.e
    for pat, want in FORMS.items():
        got = e.bits(pat)
        if want is None:
            assert got is None, (pat, got)
        else:
            assert got is not None and (got[0], got[1]) == want, (pat, got)


def rows_for_fuzz(seed):
    rnd = random.Random(seed)
    words = ["in", "a", "the", "inn", "an", "then", "at", "tin", "cat", "cot", "cut", "color", "colour", "ing", "sing", "GET", "PUT", "ab", "abd", "abcd", "acd", "ad",
             "#tag", "#", "x12", "x1", "aeiou", "bcd", "oo", "e", "i_n", "in_", "_a", "a1", "1a", "the9", "\n", "a\nthe", "x07y", "ac", "bd", "bc", "axc", "a c"]
    out = []
    for _ in range(700):
        n = rnd.randint(0, 14)
        seps = [" ", " ", " ", ",", ".", "\n", "_", "-", "", "  "]
        s = "".join(rnd.choice(words) + rnd.choice(seps) for _ in range(n))
        if rnd.random() < 0.3:
            s = s.strip()
        out.append(s[: rnd.choice([95, 95, 96, 120, 40])])
    out += fuzzdata.rows(seed + 1, 300, max_len=97, alphabet=list("aeinth  .,\n_xyGETPU#01c\x00"))
    out += fuzzdata.log_rows(seed + 2, 200)
    out += ["", None, "a", "in", "the", " a ", "a a a a", "in the a", "thea", "a" * 95, "a " * 47 + "a", "a " * 48, "x" * 94 + "a", "x" * 93 + " a", "the" * 31 + "xx",
            "naïve a in", "in\x00a", "é a", "aeiou" * 19, "b" * 95, "ua" * 47, "colourcolorcolouur", "ing\ning", "GET x\nPUT y", "PUTGET", "#a#b ##c", "x123 x1 x12"]
    return out


@pytest.mark.parametrize("pat", [p for p, f in FORMS.items() if f])
def test_bit_form_vs_oracle(emu_engine, oracle_engine, pat):
    e = emu_engine.e
    s = rows_for_fuzz(len(pat) * 7 + 1)
    e.set_engine(1)
    e.set_bits(1)
    try:
        assert emu_engine.contains_re(s, pat) == oracle_engine.contains_re(s, pat), pat
        assert emu_engine.match(s, pat) == oracle_engine.match(s, pat), pat
        assert emu_engine.count_re(s, pat) == oracle_engine.count_re(s, pat), pat
        for repl in ("=", "", "<LONGER>"):
            assert emu_engine.replace_re(s, pat, repl, -1) == oracle_engine.replace_re(s, pat, repl, -1), (pat, repl)
        assert emu_engine.replace_re(s, pat, "#", 2) == oracle_engine.replace_re(s, pat, "#", 2), pat
    finally:
        e.set_bits(0)


def test_generated_alternations_vs_oracle(emu_engine, oracle_engine):
    """Random alternations of short literals / classes with optional parts and assertions: every one that converts is
    compared with the oracle on rows made of its own alphabet (dense matches, overlapping candidates)."""
    rnd = random.Random(5)
    e = emu_engine.e
    e.set_engine(1)
    atoms = ["a", "b", "c", "[ab]", "[^a]", ".", r"\w", r"\d", " "]
    pre = ["", "", "", r"\b", r"\B", "^", r"\A"]
    post = ["", "", "", r"\b", r"\B", "$", r"\Z"]
    done = 0
    for _ in range(400):
        alts = []
        for _ in range(rnd.randint(1, 4)):
            body = "".join(rnd.choice(atoms) + rnd.choice(["", "", "", "?"]) for _ in range(rnd.randint(1, 4)))
            alts.append(rnd.choice(pre) + body + rnd.choice(post))
        pat = "|".join(alts) if rnd.random() < 0.7 else "(" + ")|(".join(alts) + ")"
        if rnd.random() < 0.25:
            pat = rnd.choice(pre) + rnd.choice(atoms) + rnd.choice(atoms) + "+" + rnd.choice(post) + rnd.choice(["", "", r"\b", "$"])
        if e.bits(pat) is None:
            continue
        done += 1
        s = fuzzdata.rows(done, 120, max_len=96, alphabet=list("aabbc  1_\n.\x00")) + ["", "a", "ab", "abc", "a b c", "a" * 95, "ab " * 31]
        e.set_bits(1)
        try:
            assert emu_engine.contains_re(s, pat) == oracle_engine.contains_re(s, pat), pat
            assert emu_engine.match(s, pat) == oracle_engine.match(s, pat), pat
            assert emu_engine.count_re(s, pat) == oracle_engine.count_re(s, pat), pat
            assert emu_engine.replace_re(s, pat, "<>", -1) == oracle_engine.replace_re(s, pat, "<>", -1), pat
        finally:
            e.set_bits(0)
    assert done >= 150, done
