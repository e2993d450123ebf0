"""One CPU-baseline worker of bench.py's all-cores leg: synthesises its row range of the C3
column with the oracle, reports "ready", waits for "go" on stdin, runs split(' ') +
replace_re(IPv4) on it and reports the bytes it processed.  Test infrastructure: never part
of the product path.  usage: python tests/cpu_worker.py <first_row> <rows> <program.npy>  (the compiled regex program, int32 words)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import cpulibs  # noqa: E402


def main():
    first, rows = int(sys.argv[1]), int(sys.argv[2])
    orc = cpulibs.Oracle()
    c = orc.synth(3, first, rows)
    blob = np.ascontiguousarray(np.load(sys.argv[3]), dtype=np.int32)
    print("ready", flush=True)
    sys.stdin.readline()
    orc.split(c, " ")
    orc.replace_re(c, blob, "<IP>")
    print("done %d" % c.chars.size, flush=True)


if __name__ == "__main__":
    main()
