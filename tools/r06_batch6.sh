#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu > gpurun_out/r06/t6_round6.log 2>&1
echo "round6 rc=$?" >> gpurun_out/r06/t6_round6.log
python tools/probe_c5regex.py 62500000 1000 > gpurun_out/r06/c5regex_b6.jsonl 2> gpurun_out/r06/c5regex_b6.err
R=$PWD
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/suite_trace -- python -m pytest tests -m gpu -q > gpurun_out/r06/t6_suite_traced.log 2>&1
find gpurun_out/suite_trace -name "*kernel_trace.csv" -delete
python tools/kernel_coverage.py gpurun_out/suite_trace > gpurun_out/r06/kernel_coverage.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/r06/bench6.json 2> gpurun_out/r06/bench6.err
grep -E "passed|failed" gpurun_out/r06/t6_suite_traced.log | tail -2; tail -3 gpurun_out/r06/t6_round6.log; tail -1 gpurun_out/r06/kernel_coverage.txt
