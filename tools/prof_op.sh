#!/bin/bash
# kernel stats of single ops (GPU box): bash tools/prof_op.sh extract backrefs ...
export TMPDIR=/tmp
REPO=$PWD
OUT=$PWD/gpurun_out/r04/prof_op
mkdir -p $OUT
for op in "$@"; do
  cd /tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$op -o $op -- python $REPO/tools/probe_op.py $op > $OUT/$op.log 2>&1
  cd $REPO
  echo "== $op"
  python - $OUT/$op <<'P'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no stats"); raise SystemExit
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-90s calls %4s avg %9.3f ms  %5.1f %%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
print("sum of kernel time / 4 calls: %.3f ms" % (tot / 4e6))
P
  find $OUT/$op -name "*kernel_trace.csv" -delete
done
