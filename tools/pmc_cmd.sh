#!/bin/bash
# SQ instruction-mix / stall counters of the kernels a command launches (GPU box, via gpurun):
#   bash tools/pmc_cmd.sh <tag> <kernel-name regex> <command ...>
# Two separate --pmc passes (no trace domain besides --kernel-trace); per-launch averages per kernel.
export TMPDIR=/tmp
TAG=$1; KRE=$2; shift 2
REPO=$PWD
OUT=$PWD/gpurun_out/pmcc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $OUT/a -o a -- "$@" > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT/b -o b -- "$@" > $OUT/b.log 2>&1
cd $REPO
python - "$OUT" "$KRE" <<'PY' | tee $OUT/summary.txt
import csv, glob, collections, sys, re
out, kre = sys.argv[1], re.compile(sys.argv[2])
agg = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if kre.search(n):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", n)
            short = m.group(0)[:56] if m else n[:56]
            key = (short, row["Counter_Name"])
            agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
for (k, c), (v, n) in sorted(agg.items()):
    print("%-58s %-24s %16.0f per launch (%d)" % (k, c, v / n, n))
PY
find $OUT -name "*.csv" -size +5M -delete
