#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu > gpurun_out/r06/t5_round6.log 2>&1
echo "round6 rc=$?" >> gpurun_out/r06/t5_round6.log
for v in tile4 tile2 rowwise; do
  case $v in tile4) e="CS_CAT_ROWS_PER_LANE=4";; tile2) e="CS_CAT_ROWS_PER_LANE=2";; rowwise) e="CS_CAT_ROWWISE=1";; esac
  for op in cat1m cat1k; do
    echo "$v $(env $e python tools/prof_cat.py $op 2>/dev/null | tail -1)" >> gpurun_out/r06/cat_timers.txt
  done
done
python tools/bench_ops.py --only c4 > gpurun_out/r06/ops_c4.jsonl 2> gpurun_out/r06/ops_c4.err
R=$PWD
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/suite_trace -- python -m pytest tests -m gpu -q > gpurun_out/r06/t5_suite_traced.log 2>&1
find gpurun_out/suite_trace -name "*kernel_trace.csv" -delete
python tools/kernel_coverage.py gpurun_out/suite_trace > gpurun_out/r06/kernel_coverage.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/r06/bench5.json 2> gpurun_out/r06/bench5.err
du -sh gpurun_out/suite_trace; grep -E "passed|failed" gpurun_out/r06/t5_suite_traced.log | tail -2; tail -3 gpurun_out/r06/t5_round6.log; cat gpurun_out/r06/cat_timers.txt; tail -2 gpurun_out/r06/kernel_coverage.txt
