#!/bin/bash
# Dev probe (GPU box): k_cat_insert time under the measurement switches of CS_CAT_DEBUG
# (1: no byte compare, 2: no table at all).  Results under those switches are wrong by design.
export TMPDIR=/tmp
REPO=$PWD
for dbg in 0 1 2; do
  OUT=$REPO/gpurun_out/probe_cat/d$dbg
  rm -rf $OUT; mkdir -p $OUT
  (cd /tmp && CS_CAT_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c -- python $REPO/tools/bench_ops.py --only C4 > $OUT/log 2>&1)
  echo "CS_CAT_DEBUG=$dbg"; grep "k_cat_insert" $OUT/c_kernel_stats.csv | cut -d, -f1-4,6,7 | cut -c1-60,150-
  find $OUT -name "*trace.csv" -delete
done
