"""Seeded random string columns for differential tests (python lists)."""
import random

ALPHA = list("abcABC xyz_-,.019 \t\n") + ["é", "É", "ß", "İ", "٣", " ", "Σ", "😀", "ａ", "ǅ"]


def rows(seed, n, max_len=24, null_p=0.08, empty_p=0.08, alphabet=None, nul_p=0.1):
    """(`nul_p`: the share of rows in which one character is a NUL byte -- the reference's strings are counted, not terminated,
    and its regex executor treats the byte in two ways: tests/test_nul_bytes.py.  Drawn from a generator of its own, so that the
    rows without one are the rows of earlier rounds.)"""
    rnd = random.Random(seed)
    rnd0 = random.Random(seed * 7919 + 13)
    alphabet = alphabet or ALPHA
    out = []
    for _ in range(n):
        u = rnd.random()
        if u < null_p:
            out.append(None)
        elif u < null_p + empty_p:
            out.append("")
        else:
            r = [rnd.choice(alphabet) for _ in range(rnd.randint(1, max_len))]
            if rnd0.random() < nul_p:
                r[rnd0.randrange(len(r))] = "\x00"
            out.append("".join(r))
    return out


def log_rows(seed, n):
    """IPv4-ish log lines with adversarial near-misses."""
    rnd = random.Random(seed)
    out = []
    for _ in range(n):
        parts = []
        for _ in range(rnd.randint(0, 5)):
            k = rnd.random()
            if k < 0.35:
                parts.append(".".join(str(rnd.randint(0, 999)) for _ in range(rnd.choice([2, 3, 4, 4, 4, 5]))))
            elif k < 0.45:
                parts.append("." * rnd.randint(1, 3))
            elif k < 0.55:
                parts.append(str(rnd.randint(0, 99999)))
            else:
                parts.append("".join(rnd.choice("abcxyz_é") for _ in range(rnd.randint(1, 6))))
        out.append(rnd.choice(["", " ", "x"]).join(parts) if rnd.random() < 0.2 else " ".join(parts))
    return out


# matches around the limits of the backward group resolution (regex_tdfa.h: group_find_back): 31 / 32 / 33 / 60 steps,
# matches that end with the row, non-ASCII inside or next to the match, groups that do not take part, nested and
# repeated groups, more than four groups
GROUP_EDGE_PATTERNS = [r"(\w+)=(\d+)(?:\.(\d+))?", r"((a+)|(x+))=(\d*)", r"(\w)(\w)(\w)(\w)(\w)(\w)?(\w)?", r"(a|b|c)+(X)?",
                       r"(\d+)\.(\d+)\.(\d+)\.(\d+)$", r"((\w+) )?(/\S*)", r"(a(b(c)?)?)+", r"(.*)=(.*)", r"(é+)( ?)(k)?"]


def group_edge_rows():
    out = []
    for k in (1, 2, 30, 31, 32, 33, 60):
        out += ["x" * k + "=1.2", "k " + "a" * k + "=" + "9" * k + " z", "a" * k]
    out += ["é=1.2", "ab=é1", "key=12.5é", "é" * 20 + " k=3.4", "k=3.4" + "é" * 20, None, "", "=", "a=", "=1", "a=1", "ab=12.34=56.78"]
    out += ["GET /a/b 10.2.3.4 200", "POST / 1.2.3.4", "abcdefgh", "aXbXcXdXeXfXgXhX", "abcabcabc", "aaaa", "ab" * 20]
    return out * 3  # several sub-tiles' worth, the same matches at different lanes
