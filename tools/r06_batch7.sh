#!/bin/bash
mkdir -p gpurun_out/r06
CS_STREAM_INFO=1 CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_prof.so python tools/probe_c5phase.py > gpurun_out/r06/c5_phases_b7.txt 2>&1
for v in base e192 base2 e192b; do
  lib=/root/repo/custrings_amd/libcustrings_amd.so
  case $v in e192*) lib=/root/repo/custrings_amd/libcustrings_amd_e192.so;; esac
  CS_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu --cold-steps 0 --concurrent-steps 0 --no-box > gpurun_out/r06/ab7_$v.json 2> gpurun_out/r06/ab7_$v.err
done
R=$PWD
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c2_trace -- python bench.py --config c2 --steps 50 --warmup 5 > gpurun_out/r06/bench_c2_traced.json 2> gpurun_out/r06/bench_c2_traced.err
python bench.py --config c2 --steps 50 --warmup 5 > gpurun_out/r06/bench_c2.json 2> gpurun_out/r06/bench_c2.err
find gpurun_out/c2_trace -name "*kernel_trace.csv" -delete
cat gpurun_out/r06/bench_c2.json | head -c 600
