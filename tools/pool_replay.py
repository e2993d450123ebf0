#!/usr/bin/env python3
"""Replays a pool trace (tools/probe_pool_trace.py: `pool+ bytes ...` / `pool- bytes ...` lines on stderr) through an
allocation policy and counts the hipMalloc calls per step: the buffer pool's policy (cs_core.hip: size_class, reuse_limit,
the first-touch provision of page-sized blocks) is tried out here before it is built.
usage: python tools/pool_replay.py trace.txt [limit_factor]"""
import bisect
import sys
from collections import defaultdict


def size_class(want):
    if want <= 4096:
        return 4096
    small = want < (1 << 20)
    w = want + want // (8 if small else 32)
    e = w.bit_length() - 1
    step = 1 << (e - (2 if small else 3))
    return (w + step - 1) // step * step


def main():
    path = sys.argv[1]
    factor = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
    provision = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    idle = []  # sorted capacities
    live = defaultdict(list)  # bytes -> capacities of live blocks with that request size
    misses = defaultdict(int)
    mallocs_bytes = defaultdict(int)
    step = "start"
    tiny_provisioned = False
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "step":
            step = line.strip()
        elif t[0] == "pool+":
            b = int(t[1])
            want = (b + 64 + 511) // 512 * 512
            limit = want + max(int(want * (factor - 1)) + 4096, 256 << 10)
            i = bisect.bisect_left(idle, want)
            if i < len(idle) and idle[i] <= limit:
                cap = idle.pop(i)
            else:
                cap = size_class(want)
                misses[step] += 1
                mallocs_bytes[step] += cap
                if cap == 4096 and not tiny_provisioned and provision:
                    tiny_provisioned = True
                    for _ in range(provision):
                        bisect.insort(idle, 4096)
            live[b].append(cap)
        elif t[0] == "pool-":
            b = int(t[1])
            if live[b]:
                bisect.insort(idle, live[b].pop())
    for k in misses:
        print(k, "mallocs", misses[k], "bytes", mallocs_bytes[k])
    print("idle blocks at the end:", len(idle), "bytes", sum(idle))


if __name__ == "__main__":
    main()
