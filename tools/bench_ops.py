#!/usr/bin/env python3
"""Per-operation throughput of every SURVEY section-8(a) row on the synthetic configs of
BASELINE.json (C2-C5), one GPU, inputs resident in HBM.  Secondary to bench.py (which is the
headline metric): this prints one JSON line per op with wall time per call (device
synchronised), input GB/s and the fraction of the 8 TB/s HBM roofline over the op's
ALGORITHMIC bytes (SURVEY.md 8d formulas).  Usage: python tools/bench_ops.py [--scale 1.0]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime per process)

from custrings_amd import _lib, nvcategory, nvstrings, nvtext  # noqa: E402

L = _lib.lib
_lib.ensure_init(0)
SEED = 20240607
IPV4 = r"\d+\.\d+\.\d+\.\d+"
IPV4B = r"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b"
GTEST = r"(\bin\b)|(\ba\b)|(\bthe\b)"


def synth(kind, rows, param=0):
    out = C.c_void_p()
    _lib.check(L.cs_synth_column(kind, 0, rows, SEED, param, None, C.byref(out)))
    return nvstrings.nvstrings(out.value)


def nbytes(col):
    return int(L.cs_column_nbytes(col.m_cptr))


def col_ov(col):
    """offset + validity bytes per row of an output column, at the offset width it was actually written with (bench.py's rule)"""
    return int(L.cs_column_offset_width(col.m_cptr)) + 0.125


def timed(fn, reps=3):
    fn()  # warm-up (allocator cache, kernel load)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
        del r
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def report(config, op, rows, in_bytes, alg_bytes, dt):
    print(json.dumps({"config": config, "op": op, "rows": rows, "ms": round(dt * 1e3, 3),
                      "input_GBps": round(in_bytes / dt / 1e9, 1), "alg_bytes_per_row": round(alg_bytes / rows, 1),
                      "alg_GBps": round(alg_bytes / dt / 1e9, 1), "frac_of_8TBps": round(alg_bytes / dt / 8e12, 4)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0, help="row-count multiplier (1.0 = BASELINE.json single-GPU sizes)")
    ap.add_argument("--only", default="C2,C3,C4,C5", help="comma-separated configs to run")
    a = ap.parse_args()
    only = set(a.only.split(","))
    ov = 8.125  # native offset + validity bytes per row

    if "C2" in only:
        run_c2(a, ov)
    if "C3" in only:
        run_c3(a, ov)
    if "C4" in only:
        run_c4(a, ov)
    if "C5" in only:
        run_c5(a, ov)


def run_c2(a, ov):
    # ---- C2: 10M x 64 chars, lower + strip + split(' ')
    rows = int(10_000_000 * a.scale)
    c2 = synth(2, rows)
    b = nbytes(c2)
    low = c2.lower()
    report("C2", "lower", rows, b, 2 * b + 2 * ov * rows, timed(lambda: c2.lower()))
    st = low.strip()
    report("C2", "strip", rows, nbytes(low), nbytes(low) + nbytes(st) + 2 * ov * rows, timed(lambda: low.strip()))
    cols = st.split(" ")
    out_b = sum(nbytes(c) for c in cols)
    report("C2", "split(' ')", rows, nbytes(st), nbytes(st) + ov * rows + out_b + sum(col_ov(c) for c in cols) * rows, timed(lambda: st.split(" ")))
    report("C2", "upper", rows, b, 2 * b + 2 * ov * rows, timed(lambda: c2.upper()))
    res = torch.empty(rows, dtype=torch.int32, device="cuda")
    report("C2", "find('é')", rows, b, b + ov * rows + 4 * rows, timed(lambda: c2.find("é", devptr=res.data_ptr())))
    resb = torch.empty(rows, dtype=torch.uint8, device="cuda")
    report("C2", "contains('ab', regex=False)", rows, b, b + ov * rows + rows, timed(lambda: c2.contains("ab", regex=False, devptr=resb.data_ptr())))
    rl = c2.replace("a", "xx", regex=False)
    report("C2", "replace('a','xx') literal", rows, b, b + nbytes(rl) + 2 * ov * rows, timed(lambda: c2.replace("a", "xx", regex=False)))
    rs = c2.replace("ab", "x", regex=False)
    report("C2", "replace('ab','x') literal", rows, b, b + nbytes(rs) + 2 * ov * rows, timed(lambda: c2.replace("ab", "x", regex=False)))
    # ingest / egress through the reference's Arrow boundary (int32 offsets + bitmask, device buffers):
    # create_offsets (NVStrings.cu:402-482), create_from_offsets (NVStringsImpl.cu:399-444), byte_count
    chars = torch.empty(b + 64, dtype=torch.uint8, device="cuda")
    offs = torch.empty(rows + 1, dtype=torch.int32, device="cuda")
    mask = torch.empty((rows + 7) // 8 + 8, dtype=torch.uint8, device="cuda")
    report("C2", "egress to_offsets (device, int32)", rows, b, 2 * b + (ov + 4.125) * rows,
           timed(lambda: c2.to_offsets(chars.data_ptr(), offs.data_ptr(), mask.data_ptr(), bdevmem=True)))
    report("C2", "ingest from_offsets (device, int32)", rows, b, 2 * b + (ov + 4.125) * rows,
           timed(lambda: nvstrings.from_offsets(chars.data_ptr(), offs.data_ptr(), rows, mask.data_ptr(), 0, bdevmem=True)))
    lens = torch.empty(rows, dtype=torch.int32, device="cuda")
    report("C2", "byte_count (device)", rows, b, (ov + 4) * rows, timed(lambda: c2.byte_count(lens.data_ptr(), bdevmem=True)))
    del c2, low, st, cols, rl, rs, chars, offs, mask, lens



def run_c3(a, ov):
    # ---- C3: 100M log lines, contains_re + replace_re + split
    rows = int(100_000_000 * a.scale)
    c3 = synth(3, rows)
    b = nbytes(c3)
    resb = torch.empty(rows, dtype=torch.uint8, device="cuda")
    report("C3", "contains_re(IPv4)", rows, b, b + ov * rows + rows, timed(lambda: c3.contains(IPV4, devptr=resb.data_ptr())))
    resi = torch.empty(rows, dtype=torch.int32, device="cuda")
    report("C3", "count_re(IPv4)", rows, b, b + ov * rows + 4 * rows, timed(lambda: c3.count(IPV4, devptr=resi.data_ptr())))
    fa = c3.findall(IPV4)
    report("C3", "findall(IPv4) -> %d columns" % len(fa), rows, b, b + ov * rows + sum(nbytes(c) for c in fa) + sum(col_ov(c) for c in fa) * rows,
           timed(lambda: c3.findall(IPV4)))
    del fa
    report("C3", "extract((\\d+)\\.(\\d+)\\.\\d+\\.(\\d+) ), 3 groups", rows, b, b + ov * rows + 3 * (ov * rows) + 6 * rows,
           timed(lambda: c3.extract(r"(\d+)\.(\d+)\.\d+\.(\d+) "), reps=2))
    bk = c3.replace_with_backrefs(r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1")
    report("C3", "replace_with_backrefs(IPv4 octets reversed)", rows, b, b + nbytes(bk) + 2 * ov * rows,
           timed(lambda: c3.replace_with_backrefs(r"(\d+)\.(\d+)\.(\d+)\.(\d+)", r"\4.\3.\2.\1"), reps=2))
    del bk
    rep = c3.replace(IPV4, "<IP>")
    report("C3", "replace_re(IPv4,'<IP>')", rows, b, b + nbytes(rep) + 2 * ov * rows, timed(lambda: c3.replace(IPV4, "<IP>")))
    del rep
    # BASELINE.md section 3's secondary pattern (26 instructions) and the reference gtest's alternation of word-bounded
    # literals (cpp/tests/test_replace.cpp:41), each with the executor it takes (cs_regex_engine)
    for name, pat, repl in (("IPv4 with \\b and {1,3}", IPV4B, "<IP>"), ("(\\bin\\b)|(\\ba\\b)|(\\bthe\\b)", GTEST, "=")):
        re = nvstrings._compile(pat)
        e = int(L.cs_regex_engine(re))
        how = ("tagged DFA, %d states, %d threads%s" % (e >> 16, (e >> 8) & 15, ", unit scan offered" if e & 2 else "")) if e & 1 else "list simulator"
        ninst = int(L.cs_regex_inst_count(re))
        L.cs_regex_destroy(re)
        report("C3", "contains_re(%s) [%d instructions; %s]" % (name, ninst, how), rows, b, b + ov * rows + rows,
               timed(lambda: c3.contains(pat, devptr=resb.data_ptr())))
        report("C3", "match(%s)" % name, rows, b, b + ov * rows + rows, timed(lambda: c3.match(pat, devptr=resb.data_ptr())))
        cnt32 = torch.zeros(rows, dtype=torch.int32, device="cuda")
        report("C3", "count_re(%s)" % name, rows, b, b + ov * rows + 4 * rows, timed(lambda: c3.count(pat, devptr=cnt32.data_ptr())))
        del cnt32
        rep = c3.replace(pat, repl)
        route = L.cs_debug_last_route().decode()
        report("C3", "replace_re(%s,'%s') [route: %s]" % (name, repl, route), rows, b, b + nbytes(rep) + 2 * ov * rows, timed(lambda: c3.replace(pat, repl), reps=2))
        del rep
    # a small set in a `+` loop: candidates in a good share of the bytes (the bit-parallel form, regex_bits.h)
    rep = c3.replace(r"[aeiou]+", "*")
    report("C3", "replace_re([aeiou]+,'*') [route: %s]" % L.cs_debug_last_route().decode(), rows, b, b + nbytes(rep) + 2 * ov * rows, timed(lambda: c3.replace(r"[aeiou]+", "*"), reps=2))
    del rep
    report("C3", "match(IPv4)", rows, b, b + ov * rows + rows, timed(lambda: c3.match(IPV4, devptr=resb.data_ptr())))
    rc = c3.rsplit(" ")
    report("C3", "rsplit(' ') (no limit: the split kernels)", rows, b, b + ov * rows + sum(nbytes(c) for c in rc) + sum(col_ov(c) for c in rc) * rows,
           timed(lambda: c3.rsplit(" ")))
    del rc
    rc = c3.rsplit(" ", 3)
    report("C3", "rsplit(' ', 3) (the split kernels, the row's first delimiters struck from the mask)", rows, b, b + ov * rows + sum(nbytes(c) for c in rc) + sum(col_ov(c) for c in rc) * rows,
           timed(lambda: c3.rsplit(" ", 3), reps=2))
    del rc
    cols = c3.split(" ")
    out_b = sum(nbytes(c) for c in cols)
    out_ov = sum(col_ov(c) for c in cols)
    del cols
    report("C3", "split(' ')", rows, b, b + ov * rows + out_b + out_ov * rows, timed(lambda: c3.split(" ")))
    cols = c3.split()
    out_b = sum(nbytes(c) for c in cols)
    out_ov = sum(col_ov(c) for c in cols)
    del cols
    report("C3", "split() whitespace", rows, b, b + ov * rows + out_b + out_ov * rows, timed(lambda: c3.split(), reps=2))
    # first touch: a fresh column's first op also pays the column's metadata passes (largest 64-row span, longest row, byte
    # classes: kept on the immutable column afterwards) -- the steady-state figures above do not show them
    def fresh_first(op):
        c = synth(3, rows)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = op(c)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        del r, c
        return dt
    fresh_first(lambda c: c.split(" "))  # (kernel load, allocator)
    report("C3", "split(' '), FIRST op on a fresh column", rows, b, b + ov * rows + out_b + out_ov * rows, fresh_first(lambda c: c.split(" ")))
    rep = c3.replace(IPV4, "<IP>")
    # (once unmeasured: with no block of the output's size in the pool the call pays a hipMalloc of gigabytes -- 120 ms -- which is
    # the allocator's first touch, not the column's; tools/probe_fresh_first.py)
    fresh_first(lambda c: c.replace(IPV4, "<IP>"))
    report("C3", "replace_re(IPv4,'<IP>'), FIRST op on a fresh column", rows, b, b + nbytes(rep) + 2 * ov * rows, fresh_first(lambda c: c.replace(IPV4, "<IP>")))
    del rep
    del c3, resb, resi



def run_c4(a, ov):
    # ---- C4: 16-char tokens, category build (per-GPU shard of the 1B-row config: 125M rows)
    rows = int(125_000_000 * a.scale)
    # (K = 2^40 names of the log-uniform generator: nearly every row is its own key -- BASELINE.md section 3's "K = 100M" case)
    for K in (1000, 1 << 20, 1 << 27, 1 << 40):
        c4 = synth(4, rows, K)
        b = nbytes(c4)
        cat = nvcategory.from_strings(c4)
        nk = cat.keys_size()
        del cat
        report("C4", "category build K=%d (%d distinct keys)" % (K, nk), rows, b, b + ov * rows + 4 * rows, timed(lambda: nvcategory.from_strings(c4), reps=2))
        del c4



def run_c5(a, ov):
    # ---- C5: tweet-like rows, tokenize + bigrams (per-GPU shard: 62.5M rows)
    rows = int(62_500_000 * a.scale)
    c5 = synth(5, rows)
    b = nbytes(c5)
    tok = nvtext.tokenize(c5)
    t = tok.size()
    report("C5", "tokenize", rows, b, b + ov * rows + nbytes(tok) + ov * t, timed(lambda: nvtext.tokenize(c5), reps=2))
    # a single class in a `+` loop on rows of 40-150 bytes (beyond the 96-bit masks): byte-parallel compaction (cs_runs.hip)
    rp = c5.replace(r"[aeiou]+", "*")
    report("C5", "replace_re([aeiou]+,'*') [route: %s]" % L.cs_debug_last_route().decode(), rows, b, b + nbytes(rp) + 2 * ov * rows, timed(lambda: c5.replace(r"[aeiou]+", "*"), reps=2))
    del rp
    del c5
    rep = c5.replace(r"[aeiou]+", "*") if False else None
    ng = nvtext.ngrams(tok, 2, "_")
    report("C5", "ngrams(2)", t, nbytes(tok), nbytes(tok) + ov * t + nbytes(ng) + ov * ng.size(), timed(lambda: nvtext.ngrams(tok, 2, "_"), reps=2))


if __name__ == "__main__":
    main()
