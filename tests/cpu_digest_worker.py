"""One worker of the full-size oracle cross-check (tests/test_gpu_round5.py): for every row window it is given, synthesises
the window of the C3 column with the oracle, runs split(' ') and replace_re(IPv4, '<IP>') (and the ops CS_DIGEST_EXTRA lists) on it and prints the digests
(include/cs_synth_spec.h: cs_digest_row) of every output column.  Test infrastructure: never part of the product path.
usage: python tests/cpu_digest_worker.py <program.npy> <rows per window> <first row> [<first row> ...]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import cpulibs  # noqa: E402


def main():
    blob = np.ascontiguousarray(np.load(sys.argv[1]), dtype=np.int32)
    rows = int(sys.argv[2])
    orc = cpulibs.Oracle()
    extra = json.load(open(os.environ["CS_DIGEST_EXTRA"])) if os.environ.get("CS_DIGEST_EXTRA") else []
    for first in (int(a) for a in sys.argv[3:]):
        c = orc.synth(3, first, rows)
        cols = orc.split(c, " ")
        rep = orc.replace_re(c, blob, "<IP>")
        out = {"first": first, "rows": rows, "split": [orc.digest(k) for k in cols], "replace": orc.digest(rep)}
        # more regex ops on the same window (CS_DIGEST_EXTRA: a JSON list of [name, "replace" | "backrefs", program.npy, text])
        for name, kind, prog, text in extra:
            b = np.ascontiguousarray(np.load(prog), dtype=np.int32)
            out[name] = orc.digest(orc.replace_re(c, b, text) if kind == "replace" else orc.replace_with_backrefs(c, b, text))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
