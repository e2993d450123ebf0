"""Long rows as pieces (custrings_amd/csrc/cs_virtual.hip), the property behind it, on the CPU: whenever the product's DFA
builder says white space is a SAFE CUT for a program (regex_tdfa.cpp: header word 31 bit 25), the reference's semantics --
the oracle, which knows nothing of pieces -- must give, for any well-formed row cut behind white space into pieces,
    contains_re(row) = OR contains_re(piece),  count_re(row) = sum count_re(piece),
    replace_re(row)  = concatenation of replace_re(piece).
Checked for every committed fixture pattern the builder marks and for hand-picked ones, on random rows of words, digits,
dots, tabs, line feeds and multi-byte characters; patterns the builder must NOT mark are listed too."""
import json
import os

import numpy as np
import pytest

import cpulibs
import engines

HERE = os.path.dirname(os.path.abspath(__file__))
PIECE = 92

SAFE = [r"\d+\.\d+\.\d+\.\d+", r"(\bin\b)|(\ba\b)|(\bthe\b)", r"\w+@\w+", r"#\w+", r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", r"[aeiou]+", r"\d+$", r"GET|POST",
        r"\bthe\b", r"[a-z]+ing\b", r"\S+", r"x*y", r"\w+", r"[0-9a-f]+-[0-9a-f]+", r"(\d+)\.(\d+)", r"é+", r"\d+\b", r"\Bing"]
UNSAFE = [r"^\d+", r"\s+", r"(\d+) (\d+)", r"[^a]+", r".+", r"\w+\s\w+", r"a b", r"[^ ]+$", r"a\nb", r"\t", r"^", r"\Athe", r"x*"]


def safe_cut(emu, pattern):
    re = emu.compile(pattern)
    try:
        return bool((emu._regex_units(re) >> 25) & 1)
    finally:
        emu._regex_free(re)


def cut_rows(rows):
    """every row as its pieces: greedy, each piece ends behind the last space / tab / LF / CR of its first 92 bytes"""
    pieces, first = [], [0]
    for r in rows:
        s = 0
        while len(r) - s > PIECE:
            w = r[s:s + PIECE]
            p = max(w.rfind(b" "), w.rfind(b"\t"), w.rfind(b"\n"), w.rfind(b"\r"))
            assert p >= 0
            pieces.append(r[s:s + p + 1])
            s += p + 1
        pieces.append(r[s:])
        first.append(len(pieces))
    return pieces, first


def col_of(rows):
    chars = np.frombuffer(b"".join(rows), dtype=np.uint8).copy()
    offs = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    return cpulibs.Col(chars, offs, None)


def random_rows(rng, n):
    words = [b"the", b"in", b"a", b"running", b"ing", b"10.2.33.4", b"7.7", b"#tag", b"me@x.org", b"GET", b"POST", b"deadbeef-01", b"caf\xc3\xa9", b"\xe2\x82\xac5", b"y", b"xxy", b"12", b"x",
             b"\xc3\xa9\xc3\xa9", b"aeiou", b"thein", b"a.b"]
    seps = [b" ", b" ", b" ", b"  ", b"\t", b"\n", b"\r\n", b" \t "]
    out = []
    for _ in range(n):
        target = int(rng.integers(0, 400))
        parts, size = [], 0
        while size < target:
            w = words[int(rng.integers(0, len(words)))] + seps[int(rng.integers(0, len(seps)))]
            parts.append(w)
            size += len(w)
        row = b"".join(parts)
        if rng.random() < 0.3:
            row = row.rstrip()
        out.append(row)
    return out


def fixture_patterns():
    with open(os.path.join(HERE, "golden", "regex_programs.json")) as f:
        progs = json.load(f)["programs"]
    return sorted(progs.keys())  # (pattern -> program words)


def test_safe_cut_bit_on_known_patterns():
    emu = cpulibs.RowEmu()
    assert [p for p in SAFE if not safe_cut(emu, p)] == [], "programs for which white space IS a safe cut"
    assert [p for p in UNSAFE if safe_cut(emu, p)] == [], "programs that can tell a piece from a row"


def test_rows_equal_their_pieces_under_the_reference_semantics():
    emu = cpulibs.RowEmu()
    orc = cpulibs.Oracle()
    rng = np.random.default_rng(25)
    rows = random_rows(rng, 600)
    pieces, first = cut_rows(rows)
    assert max(len(p) for p in pieces) <= PIECE and len(pieces) > len(rows)
    crow, cpiece = col_of(rows), col_of(pieces)
    first = np.asarray(first)
    pats = [p for p in SAFE + fixture_patterns() if safe_cut(emu, p)]
    assert len(pats) >= len(SAFE)
    checked = 0
    for pat in pats:
        blob = engines.reference_blob(pat)
        if blob is None:
            blob = engines.product_blob(pat)
        blob = np.ascontiguousarray(blob, dtype=np.int32)
        rc, _ = orc.count_re(crow, blob)
        pc, _ = orc.count_re(cpiece, blob)
        sums = np.add.reduceat(pc, first[:-1])
        assert np.array_equal(rc, sums), (pat, np.flatnonzero(rc != sums)[:3])
        rh, _ = orc.contains_re(crow, blob)
        ph, _ = orc.contains_re(cpiece, blob)
        assert np.array_equal(rh != 0, np.maximum.reduceat(ph, first[:-1]) != 0), pat
        for repl in ("", "<R>"):
            rr = orc.replace_re(crow, blob, repl).to_bytes_list()
            pr = orc.replace_re(cpiece, blob, repl).to_bytes_list()
            glued = [b"".join(pr[first[i]:first[i + 1]]) for i in range(len(rows))]
            assert rr == glued, (pat, repl, next(i for i in range(len(rows)) if rr[i] != glued[i]))
        checked += 1
    assert checked >= 20
