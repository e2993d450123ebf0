#!/usr/bin/env python3
"""Condenses a tools/profile_round.sh output directory into the small files kept under
profiles/<tag>/: kernel stats CSV, per-kernel PMC sums and traffic.json (HBM bytes per
launch; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out, tag = sys.argv[1], sys.argv[2]
KEYS = {"k_tdfa_replace_stream": "k_replace_re", "k_tdfa_replace_tile": "k_replace_re", "k_split_emit2": "k_split_emit", "k_split_emit3": "k_split_emit", "k_split_emit4": "k_split_emit", "k_split_emit5": "k_split_emit",
        "k_split_emit(": "k_split_emit", "k_split_measure": "k_split_measure"}


def find(pattern):
    hits = glob.glob(os.path.join(out, pattern), recursive=True)
    return hits[0] if hits else None


def pmc(path, counter):
    agg = {}
    if not path:
        return agg
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            for frag, key in KEYS.items():
                if frag in name:
                    a = agg.setdefault(key, [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
                    break
    return {k: v[0] / v[1] for k, v in agg.items() if v[1]}


stats = find("stats/**/*kernel_stats.csv")
if stats:
    os.makedirs(os.path.join(out, "keep"), exist_ok=True)
    with open(stats) as f, open(os.path.join(out, "keep", "%s_kernel_stats.csv" % tag), "w") as g:
        g.write(f.read())
fetch = pmc(find("pmc_fetch/**/*counter_collection.csv"), "FETCH_SIZE")
write = pmc(find("pmc_write/**/*counter_collection.csv"), "WRITE_SIZE")


def calibrate(path, counter, frag, known_bytes):
    """counter (KB) of the stream_rate kernel whose name holds `frag`, against the bytes that kernel is known to move:
    known / (KB x 1024) -- the guide's "calibrate on a known byte count in your own access pattern" (MI355X_MICROARCH.md, HBM)."""
    if not path:
        return None
    vals = []
    with open(path) as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter and frag in row["Kernel_Name"] and float(row["Counter_Value"]) > 0:
                vals.append(known_bytes / (float(row["Counter_Value"]) * 1024.0))
    vals.sort()
    return vals[len(vals) // 2] if vals else None


# the round's kernels stream with NON-TEMPORAL 16-byte loads and stores (tile_utils.h): the factors are measured on
# tools/ubench/stream_rate's kernels of the same access kinds, in the same two --pmc passes (cal_fetch / cal_write)
GiB4 = 4.0 * (1 << 30)
cal = {"fetch_nt_16B": calibrate(find("cal_fetch/**/*counter_collection.csv"), "FETCH_SIZE", "k_read16<4, true>", GiB4),
       "fetch_plain_16B": calibrate(find("cal_fetch/**/*counter_collection.csv"), "FETCH_SIZE", "k_read16<4, false>", GiB4),
       "write_nt_16B": calibrate(find("cal_write/**/*counter_collection.csv"), "WRITE_SIZE", "k_fill16<4, true>", GiB4),
       "write_plain_16B": calibrate(find("cal_write/**/*counter_collection.csv"), "WRITE_SIZE", "k_fill16<4, false>", GiB4)}
f_fetch = cal["fetch_nt_16B"] or 2.0
f_write = cal["write_nt_16B"] or 1.0
kern = {}
for k in sorted(set(fetch) | set(write)):
    fk, wk = fetch.get(k, 0.0), write.get(k, 0.0)
    kern[k] = {"fetch_size_kb": round(fk, 1), "write_size_kb": round(wk, 1), "hbm_bytes": int(fk * 1024 * f_fetch + wk * 1024 * f_write)}
tj = {"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, bench.py, 100M rows). "
              "FETCH_SIZE (KB) x 1024 x %.3f and WRITE_SIZE (KB) x 1024 x %.3f: the factors are known bytes over reported bytes of "
              "tools/ubench/stream_rate's 4 GiB read-only / write-only kernels with non-temporal 16-byte accesses -- what the headline "
              "kernels use -- measured in the same passes (the guide's gfx950 note gives 2 for plain 16-B/lane streaming reads and leaves the "
              "rest to calibration: MI355X_MICROARCH.md, HBM section)." % (f_fetch, f_write),
      "calibration": {k: (round(v, 4) if v else None) for k, v in cal.items()},
      "rows": 100000000, "source_hash": __import__("bench").source_hash(), "kernels": kern}
os.makedirs(os.path.join(out, "keep"), exist_ok=True)
with open(os.path.join(out, "keep", "traffic.json"), "w") as f:
    json.dump(tj, f, indent=2)
print(json.dumps(tj, indent=2))
if stats:
    with open(stats) as f:
        for i, line in enumerate(f):
            if i < 8:
                print(line.rstrip())
