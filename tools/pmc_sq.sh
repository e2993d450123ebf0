#!/bin/bash
# SQ instruction-mix counters for the two headline kernels (GPU box, via gpurun)
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/pmc_sq
mkdir -p $OUT
cd /tmp
for what in replace split; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $OUT/$what -o a -- python $REPO/tools/probe_replace.py 100000000 $what > $OUT/$what.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -d $OUT/$what -o b -- python $REPO/tools/probe_replace.py 100000000 $what >> $OUT/$what.log 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections, re
for what in ("replace", "split"):
    for f in sorted(glob.glob("gpurun_out/pmc_sq/%s/**/*counter_collection.csv" % what, recursive=True)):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "replace_stream" in n or "split_emit" in n or "split_measure" in n:
                key = (re.search(r"k_[a-z0-9_]+", n).group(0), row["Counter_Name"])
                agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
        for (k, c), (v, n) in sorted(agg.items()):
            print("%-30s %-24s %14.0f per launch" % (k, c, v / n))
PY
