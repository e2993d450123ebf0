#!/bin/bash
# Memory-path counters of one kernel of a secondary config (GPU box, via gpurun):
#   bash tools/pmc_kernel.sh k_cat_insert C4
# Separate --pmc passes (no trace domains besides --kernel-trace), summed per kernel launch.
export TMPDIR=/tmp
KERN=${1:-k_cat_insert}
CFG=${2:-C4}
REPO=$PWD
OUT=$PWD/gpurun_out/pmc_kernel
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
for set in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
  "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
  "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_sum" \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCC_UC_REQ_sum TCC_CC_REQ_sum TCC_NC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $OUT/p$i -o p -- python $REPO/tools/bench_ops.py --only $CFG > $OUT/p$i.log 2>&1
done
cd $REPO
python - "$KERN" <<'PY'
import csv, glob, collections, sys
kern = sys.argv[1]
for f in sorted(glob.glob("gpurun_out/pmc_kernel/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        n = row["Kernel_Name"]
        if kern in n:
            key = (n.split("(")[0][-30:], row["Counter_Name"])
            agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
    for (k, c), (v, n) in sorted(agg.items()):
        print("%-32s %-38s %16.0f per launch (%d launches)" % (k, c, v / n, n))
PY
find $OUT -name "*.csv" -size +5M -delete
