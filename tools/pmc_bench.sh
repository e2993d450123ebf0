#!/bin/bash
# SQ instruction-mix / stall counters of the headline kernels under bench.py (GPU box, via gpurun):
#   bash tools/pmc_bench.sh <tag> [VAR=value ...]     (the variables select kernel variants, e.g. CS_SPLIT_EMIT2=1)
# Two separate --pmc passes (no trace domain besides --kernel-trace); per-launch averages per kernel.
export TMPDIR=/tmp
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
REPO=$PWD
OUT=$PWD/gpurun_out/pmcb_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $OUT/a -o a -- python $REPO/bench.py --no-cpu --steps 2 --warmup 1 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/b -o b -- python $REPO/bench.py --no-cpu --steps 2 --warmup 1 > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT SQ_INST_LEVEL_LDS -d $OUT/c -o c -- python $REPO/bench.py --no-cpu --steps 2 --warmup 1 > $OUT/c.log 2>&1
cd $REPO
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if "replace_stream" in n or "split_emit" in n or "split_measure" in n or "k_replace_bp" in n:
            import re
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", n)
            short = m.group(0)[:40] if m else n[:40]
            key = (short, row["Counter_Name"])
            agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-42s %-24s %16.0f per launch (%d)" % (k, c, v / n, n))
PY
find $OUT -name "*.csv" -size +5M -delete
