#!/usr/bin/env python3
"""Soak (GPU box): every tile kernel against the row-wise kernels on randomly shaped columns of
random bytes / text, many seeds.  usage: python tools/soak_gpu.py [seconds] [first_seed]
Prints one line per failing (seed, op); exit status 1 if any."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cpulibs  # noqa: E402
import gpuutil  # noqa: E402
from custrings_amd import nvtext, nvcategory, _lib  # noqa: E402

LIB = _lib.lib

TOGGLES = ("CS_REGEX_TWO_PASS", "CS_REGEX_ROWWISE", "CS_SPLIT_GENERIC", "CS_TOKENIZE_ROWWISE", "CS_STRIP_ROWWISE",
           "CS_FIND_ROWWISE", "CS_REPLACE_ROWWISE", "CS_CASE_ROWWISE", "CS_NGRAM_ROWWISE",
           "CS_NO_CLASS_RUNS")  # (the byte-parallel class route is a fast path too: off in the witness -- it sat on both sides until round 5's last soak)
PATS = [(r"\d+\.\d+\.\d+\.\d+", "<IP>"), (r"\d", "#"), (r"[a-c]+", "xyz__"), (r"\s+", " "), (r"\w+", "<w>"), (r"b|ab", ""),
        (r"\bx", "YY"), (r"[0-9]+", "<number-here>"), (r"a", "aa"), (r"(a|b)c", "-"),
        (r"\d+\.\d+ ", "<n>"), (r"[a-c]+=>", ""), (r"\d+ab", "#"),  # (chains with a literal suffix)
        # chains with counted items and `\b` (regex_tdfa.h: chain_match_counted96)
        (r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", "<IP>"), (r"\b\d+\.\d+\b", ""), (r"\d+\.\d{2}\.\d+", "#"), (r"\b[a-c]{2,3}=", "<a-longer-replacement>"),
        (r"\d\d+\.\d+", "."), (r"\d+\.{1,2}\d+", "x"),
        # the bit-parallel form (regex_bits.h) and the single-class byte-parallel route (cs_runs.hip)
        (r"\w+", "<w>"), (r"\S+", "s"), (r"[^\w]", "_"), (r"[\w.]+", ""), (r"\W+", " "), (r"\d", "9"),  # (classes with builtins: cs_runs.hip, F_FLAG_CLASS)
        (r"(\bab\b)|(\bc\b)|(\bxyz\b)", "="), (r"[abc1]+", "*"), (r"ab|a1|bc", "#"), (r"[^ ]+", "_"), (r".", "?"), (r"x?y?z", "Q")]


def make_column(seed, max_rows=40_000):
    rng = np.random.default_rng(seed)
    kind = seed % 7
    rows = int(rng.integers(1, max_rows))
    if kind == 0:
        lens = rng.integers(0, 95, rows)
    elif kind == 1:
        lens = rng.integers(0, 20, rows)
    elif kind == 2:
        lens = np.full(rows, int(rng.integers(1, 90)))
    elif kind == 3:
        lens = rng.integers(0, 400, rows)
    elif kind == 4:
        lens = (rng.pareto(1.5, rows) * 20).astype(np.int64) % 7000
    elif kind == 5:
        lens = rng.integers(0, 64, rows) * (rng.random(rows) < 0.5)
    else:  # rows beyond the 96-byte masks but within the long-row variants
        lens = rng.integers(0, 250, rows)
    offs = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    flavour = (seed // 7) % 4
    if flavour == 0:  # arbitrary bytes
        pool = np.array(list(b"ab1.2 3.4.5.6 x9_\t\nABc") + [0, 0xC3, 0xA9, 0xE2, 0x82, 0xAC, 0xFF, 0x80, 0x1F, 0xF0, 0x9F], dtype=np.uint8)
        chars = pool[rng.integers(0, len(pool), int(offs[-1]))]
    elif flavour == 1:  # ASCII text
        pool = np.array(list(b"abcxyzABC 0123456789..  __\t"), dtype=np.uint8)
        chars = pool[rng.integers(0, len(pool), int(offs[-1]))]
    elif flavour == 3:  # valid UTF-8 rows (the reference's contract): these are also checked against the oracle
        toks = [bytes([c]) for c in b"abcab 12.3 XY  .7"] + ["é".encode(), "É".encode(), "€".encode(), "😀".encode(), "İ".encode(), "ß".encode()]
        tl = np.array([len(t) for t in toks])
        pick = rng.integers(0, len(toks), int(offs[-1]) + 1)
        # row r takes lens[r] tokens
        starts = offs.copy()
        data = b"".join(toks[i] for i in pick[:int(offs[-1])])
        bl = np.concatenate([[0], np.cumsum(tl[pick[:int(offs[-1])]])])
        offs = bl[starts].astype(np.int64)
        chars = np.frombuffer(data, dtype=np.uint8).copy()
    else:  # valid UTF-8 text cut at arbitrary places
        toks = [bytes([c]) for c in b"abcab 12.3 XY"] + ["é".encode(), "É".encode(), "€".encode(), "😀".encode(), "İ".encode(), "ß".encode()]
        data = b"".join(toks[i] for i in rng.integers(0, len(toks), int(offs[-1]) + 4))
        chars = np.frombuffer(data[:int(offs[-1])], dtype=np.uint8).copy()
    if flavour == 1 and (seed // 28) % 2 == 1 and int(offs[-1]) >= 2:
        # ASCII text with a FEW rows that hold a two-byte character or a NUL -- and such a byte where the sample's first
        # window sees it: the rows the scans put off and the single-pass replace leaves holes (cs_regex.hip: deferred / hole_mask)
        for r in np.flatnonzero((rng.random(rows) < 0.01) & (lens >= 2)):
            k = int(offs[r]) + int(rng.integers(0, int(lens[r]) - 1))
            if rng.random() < 0.8:
                chars[k], chars[k + 1] = 0xC3, 0xA9
            else:
                chars[k] = 0
        r0 = int(np.argmax(lens >= 2))
        if lens[r0] >= 2:
            chars[int(offs[r0])], chars[int(offs[r0]) + 1] = 0xC3, 0xA9
    valid = np.packbits(rng.random(rows) > 0.03, bitorder="little") if seed % 4 else None
    return cpulibs.Col(chars, offs, valid), flavour


def snapshot(g, rows, rng_seed, pats=None, regex_only=False):
    """Every op of the path on one device column -> {name: result}.  `pats`: the (pattern, replacement) pairs of the
    replace_re / contains_re / count_re leg (default: four of PATS drawn from the seed); `regex_only`: that leg and findall only."""
    L = gpuutil.lib()
    rng = np.random.default_rng(rng_seed)
    out = {}
    for pat, repl in (pats if pats is not None else [PATS[i] for i in rng.choice(len(PATS), 4, replace=False)]):
        n = int(rng.choice([-1, -1, 1, 3]))
        out["replace_re %r %r %d" % (pat, repl, n)] = gpuutil.to_col(g.replace(pat, repl, n))
        re = gpuutil.compile_re(pat)
        out["contains_re %r" % pat] = gpuutil.bools(g, "cs_contains_re", re)
        cnt = np.zeros(max(rows, 1), dtype=np.int32)
        found = C.c_int64()
        L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
        out["count_re %r" % pat] = (cnt[:rows], found.value)
        L.lib.cs_regex_destroy(re)
        if regex_only:
            out["findall %r" % pat] = [gpuutil.to_col(c) for c in g.findall(pat)]
    if regex_only:
        return out
    for pat, repl in (("a", "b"), ("ab", "x"), (" ", "  "), ("1", "one")):
        out["replace %r %r" % (pat, repl)] = gpuutil.to_col(g.replace(pat, repl, regex=False))
    for d, n in ((" ", -1), (".", 2), (None, -1), (None, 2), ("ab", -1)):
        out["split %r %d" % (d, n)] = [gpuutil.to_col(c) for c in g.split(d, n)]
    for d, n in ((" ", 2), (None, 1), ("ab", -1), ("aa", -1)):
        out["rsplit %r %d" % (d, n)] = [gpuutil.to_col(c) for c in g.rsplit(d, n)]
    for pat in (r"(\d+)\.(\d+)", r"(a|b)(c)?", r"(\w+) (\w+)", r"\b(\d{1,3})\.(\d{1,3})\b"):
        out["extract %r" % pat] = [gpuutil.to_col(c) for c in g.extract(pat)]
    for pat in (r"\d+", r"[a-c]+", r"\b\d+\.\d+\b"):
        out["findall %r" % pat] = [gpuutil.to_col(c) for c in g.findall(pat)]
    for pat, repl in ((r"(\d+)\.(\d+)", r"\2.\1"), (r"(a|b)(c)?", r"<\2\1\0>"), (r"\b(\d{1,3})\.(\d{1,3})\b", r"\2\2-\1"), (r"(\d+)\.(\d+)", r"[\0|\1]"),
                      (r"([a-c]+)=", r"=\1\1")):
        out["backrefs %r %r" % (pat, repl)] = gpuutil.to_col(g.replace_with_backrefs(pat, repl))
    tok = nvtext.tokenize(g)
    out["tokenize"] = gpuutil.to_col(tok)
    out["tokenize set"] = gpuutil.to_col(nvtext.tokenize(g, " ."))
    for N, sep in ((2, "_"), (3, ""), (2, "<sep>")):
        out["ngrams %d %r" % (N, sep)] = gpuutil.to_col(nvtext.ngrams(tok, N, sep))
    out["strip"] = gpuutil.to_col(g.strip())
    out["strip set"] = gpuutil.to_col(g.strip("ab é"))
    out["lower"] = gpuutil.to_col(g.lower())
    out["upper"] = gpuutil.to_col(g.upper())
    for sub in ("3.4", "é", "ab"):
        f = np.zeros(max(rows, 1), dtype=np.int32)
        found = C.c_int64()
        L.check(L.lib.cs_find(g.m_cptr, sub.encode(), 0, -1, f.ctypes.data, 0, None, C.byref(found)))
        out["find %r" % sub] = (f, found.value)
    return out


try:
    import engines  # noqa: E402
    ORC = cpulibs.Oracle()
except Exception as e:  # the oracle library is test infrastructure: absent -> fast-vs-rowwise only
    print("oracle leg off:", e)
    ORC = None


def oracle_leg(col, fast, seed):
    """The fast paths' results against the CPU oracle (valid UTF-8 columns only)."""
    bad = 0
    want = {}
    for k in fast:
        op, _, rest = k.partition(" ")
        try:
            if op == "replace_re":
                pat, repl, n = _parse3(rest)
                want[k] = ORC.replace_re(col, np.ascontiguousarray(engines.reference_blob(pat)), repl, n)
            elif op == "contains_re":
                import ast
                want[k] = ORC.contains_re(col, np.ascontiguousarray(engines.reference_blob(ast.literal_eval(rest))))
            elif op == "count_re":
                import ast
                want[k] = ORC.count_re(col, np.ascontiguousarray(engines.reference_blob(ast.literal_eval(rest))))
            elif op == "replace":
                pat, repl = _split_reprs(rest)
                want[k] = ORC.replace(col, pat, repl)
            elif op == "split":
                d, n = _parse2(rest)
                want[k] = ORC.split(col, d, n)
            elif op == "rsplit":
                d, n = _parse2(rest)
                want[k] = ORC.rsplit(col, d, n)
            elif op == "extract":
                import ast
                want[k] = ORC.extract(col, np.ascontiguousarray(engines.reference_blob(ast.literal_eval(rest))))
            elif op == "findall":
                import ast
                want[k] = ORC.findall(col, np.ascontiguousarray(engines.reference_blob(ast.literal_eval(rest))))
            elif op == "backrefs":
                pat, repl = _split_reprs(rest)
                want[k] = ORC.replace_with_backrefs(col, np.ascontiguousarray(engines.reference_blob(pat)), repl)
            elif k == "tokenize":
                want[k] = ORC.tokenize(col)
            elif k == "tokenize set":
                want[k] = ORC.tokenize(col, " .")
            elif k == "strip":
                want[k] = ORC.strip(col)
            elif k == "strip set":
                want[k] = ORC.strip(col, "ab é")
            elif k == "lower":
                want[k] = ORC.lower(col)
            elif k == "upper":
                want[k] = ORC.upper(col)
        except Exception as e:
            print("oracle error seed=%d op=%s: %s" % (seed, k, e), flush=True)
            bad += 1
    for k, w in want.items():
        if not same(fast[k], w):
            bad += 1
            print("ORACLE MISMATCH seed=%d rows=%d op=%s" % (seed, col.rows, k), flush=True)
    return bad


def _parse2(rest):
    import ast
    i = rest.rindex(" ")
    return ast.literal_eval(rest[:i]), ast.literal_eval(rest[i + 1:])


def _parse3(rest):
    import ast
    i = rest.rindex(" ")
    n = ast.literal_eval(rest[i + 1:])
    a, b = _split_reprs(rest[:i])
    return a, b, n


def _split_reprs(s):
    """two python reprs of str separated by one space"""
    import ast
    for i in range(1, len(s)):
        if s[i] == " ":
            try:
                return ast.literal_eval(s[:i]), ast.literal_eval(s[i + 1:])
            except Exception:
                continue
    raise ValueError(s)


def same(a, b):
    if isinstance(a, list):
        return len(a) == len(b) and all(x.same_as(y) for x, y in zip(a, b))
    if isinstance(a, tuple):
        return np.array_equal(np.asarray(a[0]).astype(np.int64), np.asarray(b[0]).astype(np.int64)) and int(a[1]) == int(b[1])
    return a.same_as(b)


def check_column(seed, pats=None, regex_only=False, max_rows=40_000, forced=(), category=True, log=print):
    """One column of the soak: the fast paths (plus the switches in `forced`, e.g. ("CS_BITS_ALWAYS",)) against (a) the
    row-wise / two-pass kernels -- the witness on arbitrary bytes, where the reference is undefined (its iterator steps by the
    lead byte's width, custring_view.inl:361-366, past the row's end) -- and (b) the CPU ORACLE on valid UTF-8 / ASCII columns
    (flavours 1 and 3: the reference's contract).  Returns the number of mismatches."""
    bad = 0
    col, flavour = make_column(seed, max_rows)
    g = gpuutil.from_col(col)
    for v in TOGGLES:  # (the library reads its switches once: cs_config_set changes them at run time)
        LIB.cs_config_set(v.encode(), None)
    for v in forced:
        LIB.cs_config_set(v.encode(), b"1")
    try:
        fast = snapshot(g, col.rows, seed, pats, regex_only)
        cat_f = nvcategory.from_strings(g) if category else None
    finally:
        for v in forced:
            LIB.cs_config_set(v.encode(), None)
    for v in TOGGLES:
        LIB.cs_config_set(v.encode(), b"1")
    try:
        slow = snapshot(g, col.rows, seed, pats, regex_only)
    finally:
        for v in TOGGLES:
            LIB.cs_config_set(v.encode(), None)
    for k in fast:
        if not same(fast[k], slow[k]):
            bad += 1
            log("MISMATCH seed=%d rows=%d op=%s" % (seed, col.rows, k))
    if flavour in (1, 3) and ORC is not None:
        bad += oracle_leg(col, fast, seed)
    if not category:
        return bad
    # category against numpy's view of the same bytes
    keys = gpuutil.to_col(cat_f.keys())
    vals = np.zeros(max(col.rows, 1), dtype=np.int32)
    cat_f.values(vals)
    rows_b = col.to_bytes_list()
    uniq = sorted({x for x in rows_b if x is not None})
    want_keys = ([None] if any(x is None for x in rows_b) else []) + uniq
    got_keys = keys.to_bytes_list()
    if got_keys != want_keys:
        bad += 1
        log("MISMATCH seed=%d category keys" % seed)
    else:
        idx = {k: i for i, k in enumerate(want_keys)}
        want_vals = np.array([idx[x] for x in rows_b], dtype=np.int32)
        if not np.array_equal(vals[:col.rows], want_vals):
            bad += 1
            log("MISMATCH seed=%d category values" % seed)
    return bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    bad = done = 0
    while time.time() - t0 < budget:
        bad += check_column(seed, log=lambda m: print(m, flush=True))
        done += 1
        seed += 1
    print("soak: %d columns, %d mismatches, %.0f s" % (done, bad, time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
