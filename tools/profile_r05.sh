#!/bin/bash
# Round 5's measurement set on one GPU box (via gpurun): bash tools/profile_r05.sh
# everything lands under gpurun_out/r05/; the summaries kept for the judge are copied to profiles/r05/ afterwards.
set -u
REPO=$PWD
OUT=$PWD/gpurun_out/r05
mkdir -p $OUT
export TMPDIR=/tmp
bash tools/profile_round.sh r05 > $OUT/profile_round.log 2>&1
cp gpurun_out/prof_r05/bench_r05.json gpurun_out/prof_r05/keep/* $OUT/ 2>/dev/null
python bench.py --config c2 --steps 20 --warmup 3 > $OUT/bench_c2_r05.json 2> $OUT/bench_c2.err
python tools/bench_ops.py > $OUT/ops_r05.jsonl 2> $OUT/ops.err
python tools/probe_bits.py > $OUT/probe_bits.jsonl 2> $OUT/probe_bits.err
python tools/probe_backrefs.py > $OUT/backrefs.txt 2>&1
python tools/probe_tok_passes.py > $OUT/tokenize_passes.txt 2>&1
(python tools/prof_cat.py cat1m; python tools/prof_cat.py cat1k; CS_CAT_PLAIN_SLOTS=1 python tools/prof_cat.py cat1m; CS_CAT_DEBUG=1 python tools/prof_cat.py cat1m; CS_CAT_DEBUG=2 python tools/prof_cat.py cat1m) 2>&1 | grep "^cat" > $OUT/category_timers.txt
bash tools/pmc_bench.sh r05 > $OUT/pmc_sq.log 2>&1
cp gpurun_out/pmcb_r05/summary.txt $OUT/sq_counters.txt 2>/dev/null
ls -la $OUT
