#!/usr/bin/env python3
"""Generated patterns on the GPU against the CPU oracle (GPU box): random alternations / classes / counted items / assertions /
`+` tails over columns of ASCII text with a few rows of two-byte characters and NUL bytes -- contains_re, count_re, replace_re.
usage: python tools/fuzz_patterns_gpu.py [seconds] [seed]"""
import ctypes as C
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import cpulibs  # noqa: E402
import engines  # noqa: E402
import gpuutil  # noqa: E402


def gen_pattern(rnd):
    atoms = ["a", "b", "c", "[ab]", "[^a]", ".", r"\w", r"\d", " ", "1", r"\s", "[a-c1]", r"\.", "x"]
    pre = ["", "", "", r"\b", r"\B", "^", r"\A"]
    post = ["", "", "", r"\b", r"\B", "$", r"\Z"]
    quant = ["", "", "", "?", "+", "*", "{1,3}", "{2}"]
    kind = rnd.random()
    if kind < 0.4:
        alts = []
        for _ in range(rnd.randint(1, 3)):
            body = "".join(rnd.choice(atoms) + rnd.choice(quant[:4]) for _ in range(rnd.randint(1, 4)))
            alts.append(rnd.choice(pre) + body + rnd.choice(post))
        return "|".join(alts) if rnd.random() < 0.7 else "(" + ")|(".join(alts) + ")"
    if kind < 0.7:
        return rnd.choice(pre) + "".join(rnd.choice(atoms) + rnd.choice(quant) for _ in range(rnd.randint(1, 4))) + rnd.choice(post)
    if kind < 0.85:
        return rnd.choice(pre) + rnd.choice(atoms) + rnd.choice(atoms) + "+" + rnd.choice(post)
    return "(" + rnd.choice(atoms) + rnd.choice(quant) + ")" + rnd.choice(atoms) + rnd.choice(quant) + rnd.choice(post)


def make_col(rng, rows, lo, hi, odd):
    glyphs = list("aabbc  1_\n.x1ab c")
    out = []
    for _ in range(rows):
        n = int(rng.integers(lo, hi + 1))
        r = "".join(rng.choice(glyphs, n))
        if rng.random() < odd and n >= 3:
            k = int(rng.integers(0, n - 2))
            r = r[:k] + str(rng.choice(["é", "Ж", "\x00"])) + r[k + 2:]
        out.append(r.encode()[:hi].decode("utf-8", "ignore").encode())
    out[0] = "é a first window b".encode()
    offs = np.zeros(rows + 1, dtype=np.int64)
    np.cumsum([len(r) for r in out], out=offs[1:])
    chars = np.frombuffer(b"".join(out), dtype=np.uint8).copy()
    valid = np.packbits(rng.random(rows) > 0.03, bitorder="little")
    return cpulibs.Col(chars, offs, valid)


def run(budget, seed, max_patterns=1 << 30):
    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    L = gpuutil.lib()
    orc = cpulibs.Oracle()
    cols = [make_col(rng, 20_000, 0, 60, 0.01), make_col(rng, 8_000, 30, 200, 0.01), make_col(rng, 10_000, 0, 40, 0.0)]
    gcols = [gpuutil.from_col(c) for c in cols]
    t0 = time.time()
    done = bad = skipped = 0
    while time.time() - t0 < budget and done < max_patterns:
        pat = gen_pattern(rnd)
        try:
            blob = np.ascontiguousarray(engines.reference_blob(pat))
        except Exception:
            skipped += 1
            continue
        try:
            re = gpuutil.compile_re(pat)
        except Exception as e:
            skipped += 1
            continue
        empty_ok = True
        try:
            for ci, (g, c) in enumerate(zip(gcols, cols)):
                has, n = gpuutil.bools(g, "cs_contains_re", re)
                r1 = L.lib.cs_debug_last_route().decode()
                want_has, want_n = orc.contains_re(c, blob)
                if not (np.array_equal(has, want_has) and n == want_n):
                    bad += 1
                    print("MISMATCH contains_re %r column %d route %s" % (pat, ci, r1), flush=True)
                cnt = np.zeros(c.rows, dtype=np.int32)
                found = C.c_int64()
                L.check(L.lib.cs_count_re(g.m_cptr, re, cnt.ctypes.data, 0, None, C.byref(found)))
                r2 = L.lib.cs_debug_last_route().decode()
                if not np.array_equal(cnt, orc.count_re(c, blob)[0]):
                    bad += 1
                    print("MISMATCH count_re %r column %d route %s" % (pat, ci, r2), flush=True)
                try:  # the span ops: findall always, extract / replace_with_backrefs where the pattern has groups
                    got_cols = [gpuutil.to_col(x) for x in g.findall(pat)]
                    want_cols = orc.findall(c, blob)
                    if len(got_cols) != len(want_cols) or not all(a.same_as(b) for a, b in zip(got_cols, want_cols)):
                        bad += 1
                        print("MISMATCH findall %r column %d route %s (%d / %d columns)" % (pat, ci, L.lib.cs_debug_last_route().decode(), len(got_cols), len(want_cols)), flush=True)
                    if "(" in pat:
                        got_cols = [gpuutil.to_col(x) for x in g.extract(pat)]
                        want_cols = orc.extract(c, blob)
                        if len(got_cols) != len(want_cols) or not all(a.same_as(b) for a, b in zip(got_cols, want_cols)):
                            bad += 1
                            print("MISMATCH extract %r column %d route %s" % (pat, ci, L.lib.cs_debug_last_route().decode()), flush=True)
                        try:
                            got = gpuutil.to_col(g.replace_with_backrefs(pat, r"<\1|\0>"))
                        except Exception:
                            got = None  # (refused: a pattern that matches the empty string)
                        if got is not None and not got.same_as(orc.replace_with_backrefs(c, blob, r"<\1|\0>")):
                            bad += 1
                            print("MISMATCH backrefs %r column %d route %s" % (pat, ci, L.lib.cs_debug_last_route().decode()), flush=True)
                except Exception as e:
                    print("EXCEPTION span ops %r: %s" % (pat, e), flush=True)
                    bad += 1
                for repl in ("<>", ""):
                    try:
                        got = g.replace(pat, repl)
                    except Exception:
                        empty_ok = False  # (a pattern the library refuses: e.g. one that matches the empty string with this replacement)
                        break
                    r3 = L.lib.cs_debug_last_route().decode()
                    if not gpuutil.to_col(got).same_as(orc.replace_re(c, blob, repl)):
                        bad += 1
                        print("MISMATCH replace_re %r -> %r column %d route %s" % (pat, repl, ci, r3), flush=True)
        finally:
            L.lib.cs_regex_destroy(re)
        done += 1
    print("pattern fuzz: %d patterns (%d skipped), %d mismatches, %.0f s" % (done, skipped, bad, time.time() - t0))
    return done, bad


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    done, bad = run(budget, seed)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
