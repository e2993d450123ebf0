#!/bin/bash
# Round 6 measurement set on one GPU box (via gpurun): bash tools/profile_r06.sh
# everything lands under gpurun_out/r06/; the summaries kept for the judge are copied to profiles/r06/ afterwards.
set -u
REPO=$PWD
OUT=$PWD/gpurun_out/r06
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/profile_round.sh r06 > $OUT/profile_round.log 2>&1
cp gpurun_out/prof_r06/bench_r06.json gpurun_out/prof_r06/keep/* $OUT/ 2>/dev/null
python bench.py --config c2 --steps 20 --warmup 3 > $OUT/bench_c2_r06.json 2> $OUT/bench_c2.err
python tools/bench_ops.py > $OUT/ops_r06.jsonl 2> $OUT/ops.err
python tools/probe_c5regex.py > $OUT/c5regex.jsonl 2> $OUT/c5regex.err
python tools/cliff_scan.py > $OUT/cliff_scan.txt 2> $OUT/cliff_scan.err
python tools/soak_gpu.py 150 90000 > $OUT/soak.txt 2>&1
tools/ubench/stream_rate 5 > $OUT/stream_rate.txt 2>&1
(python tools/prof_cat.py cat1m; python tools/prof_cat.py cat1k) 2>&1 | grep "^cat" > $OUT/category_timers.txt
bash tools/pmc_bench.sh r06 > $OUT/pmc_sq.log 2>&1
cp gpurun_out/pmcb_r06/summary.txt $OUT/sq_counters.txt 2>/dev/null
ls -la $OUT
