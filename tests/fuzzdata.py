"""Seeded random string columns for differential tests (python lists)."""
import random

ALPHA = list("abcABC xyz_-,.019 \t\n") + ["é", "É", "ß", "İ", "٣", " ", "Σ", "😀", "ａ", "ǅ"]


def rows(seed, n, max_len=24, null_p=0.08, empty_p=0.08, alphabet=None):
    rnd = random.Random(seed)
    alphabet = alphabet or ALPHA
    out = []
    for _ in range(n):
        u = rnd.random()
        if u < null_p:
            out.append(None)
        elif u < null_p + empty_p:
            out.append("")
        else:
            out.append("".join(rnd.choice(alphabet) for _ in range(rnd.randint(1, max_len))))
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
