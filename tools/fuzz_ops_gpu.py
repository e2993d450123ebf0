#!/usr/bin/env python3
"""Random ARGUMENTS for the non-regex ops on the GPU against the CPU oracle (GPU box): split / rsplit (delimiters of one to three
characters or whitespace, limits), strip (random character sets, both / left / right where the mirror offers them), literal
replace (random needles and replacements), tokenize (random delimiter sets), lower / upper -- on columns of text with a few rows
of two-byte characters and NUL bytes, short rows and rows beyond the 96-bit masks.
usage: python tools/fuzz_ops_gpu.py [seconds] [seed]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import cpulibs  # noqa: E402
import gpuutil  # noqa: E402
from custrings_amd import nvtext  # noqa: E402
from fuzz_patterns_gpu import make_col  # noqa: E402


def same_cols(a, b):
    return len(a) == len(b) and all(x.same_as(y) for x, y in zip(a, b))


def run(budget, seed, max_rounds=1 << 30):
    rnd = random.Random(seed)
    rng = np.random.default_rng(seed)
    orc = cpulibs.Oracle()
    cols = [make_col(rng, 20_000, 0, 60, 0.01), make_col(rng, 8_000, 30, 200, 0.01), make_col(rng, 10_000, 0, 40, 0.0)]
    gcols = [gpuutil.from_col(c) for c in cols]
    glyphs = list("ab c1_.\nx") + ["é", "  ", "ab", "\t"]
    t0 = time.time()
    done = bad = 0

    def check(ok, what):
        nonlocal bad
        if not ok:
            bad += 1
            print("MISMATCH " + what, flush=True)

    while time.time() - t0 < budget and done < max_rounds:
        ci = rnd.randrange(len(cols))
        g, c = gcols[ci], cols[ci]
        d = rnd.choice([None, " ", ".", "a", "ab", "  ", "_", "1", "é", "a b", "\n"])
        n = rnd.choice([-1, -1, 0, 1, 2, 3, 7])
        try:
            check(same_cols([gpuutil.to_col(x) for x in g.split(d, n)], orc.split(c, d, n)), "split %r %d column %d" % (d, n, ci))
            check(same_cols([gpuutil.to_col(x) for x in g.rsplit(d, n)], orc.rsplit(c, d, n)), "rsplit %r %d column %d" % (d, n, ci))
            chars = "".join(rnd.choice(glyphs) for _ in range(rnd.randint(0, 3))) or None
            check(gpuutil.to_col(g.strip(chars)).same_as(orc.strip(c, chars) if chars is not None else orc.strip(c)), "strip %r column %d" % (chars, ci))
            needle = "".join(rnd.choice(glyphs) for _ in range(rnd.randint(1, 2)))
            repl = "".join(rnd.choice(glyphs) for _ in range(rnd.randint(0, 3)))
            check(gpuutil.to_col(g.replace(needle, repl, regex=False)).same_as(orc.replace(c, needle, repl)), "replace %r %r column %d" % (needle, repl, ci))
            delims = "".join(rnd.choice(list(" ._1a\n")) for _ in range(rnd.randint(1, 3)))
            check(gpuutil.to_col(nvtext.tokenize(g, delims)).same_as(orc.tokenize(c, delims)), "tokenize %r column %d" % (delims, ci))
            if done % 8 == 0:
                check(gpuutil.to_col(g.lower()).same_as(orc.lower(c)), "lower column %d" % ci)
                check(gpuutil.to_col(g.upper()).same_as(orc.upper(c)), "upper column %d" % ci)
        except Exception as e:
            bad += 1
            print("EXCEPTION round %d (%r %d): %s" % (done, d, n, e), flush=True)
        done += 1
    print("ops fuzz: %d rounds, %d mismatches, %.0f s" % (done, bad, time.time() - t0))
    return done, bad


if __name__ == "__main__":
    d, b = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    sys.exit(1 if b else 0)
