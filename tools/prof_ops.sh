#!/bin/bash
# kernel-level stats of the secondary configs (GPU box): bash tools/prof_ops.sh C4,C5
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/prof_ops
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ops -- python $REPO/tools/bench_ops.py --only ${1:-C4,C5} > $OUT/ops.log 2>&1
cd $REPO
grep '^{' $OUT/ops.log
head -25 $(find $OUT -name "*kernel_stats.csv" | head -1) | cut -c1-150
find $OUT -name "*kernel_trace.csv" -delete
