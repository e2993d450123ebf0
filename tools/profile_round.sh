#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of bench.py plus two separate --pmc
# passes (FETCH_SIZE, WRITE_SIZE); everything lands under gpurun_out/prof_<tag>/.
# usage: bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
REPO=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $REPO/bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_write.log 2>&1
# the counters calibrated on known byte counts: the streaming microbenchmark under the same two passes
if [ -x $REPO/tools/ubench/stream_rate ]; then
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/cal_fetch -o cal -- $REPO/tools/ubench/stream_rate 1 > $OUT/cal_fetch.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/cal_write -o cal -- $REPO/tools/ubench/stream_rate 1 > $OUT/cal_write.log 2>&1
fi
cd $REPO
python tools/summarize_profile.py $OUT $TAG > $OUT/summary.log 2>&1
cat $OUT/summary.log
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT $OUT/* | head -40
