#!/bin/bash
# instruction-cache counters of the headline kernels (GPU box, via gpurun): bash tools/pmc_icache.sh
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/pmc_ic
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' ' > $OUT/avail.txt
for what in replace split; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/$what -o a -- python $REPO/tools/probe_replace.py 100000000 $what > $OUT/$what.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/$what -o b -- python $REPO/tools/probe_replace.py 100000000 $what >> $OUT/$what.log 2>&1
done
cd $REPO
cat $OUT/avail.txt; echo
python - <<'PY'
import csv, glob, collections, re
for what in ("replace", "split"):
    for f in sorted(glob.glob("gpurun_out/pmc_ic/%s/**/*counter_collection.csv" % what, recursive=True)):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"]
            if "replace_stream" in n or "split_emit" in n or "split_measure" in n:
                key = (re.search(r"k_[a-z0-9_]+", n).group(0), row["Counter_Name"])
                agg[key][0] += float(row["Counter_Value"]); agg[key][1] += 1
        for (k, c), (v, n) in sorted(agg.items()):
            print("%-30s %-28s %16.0f per launch" % (k, c, v / n))
PY
