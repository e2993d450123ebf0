#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu -k pool > gpurun_out/r06/t4_pool.log 2>&1
CS_STREAM_INFO=1 CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_prof.so python tools/probe_c5phase.py > gpurun_out/r06/c5_phases.txt 2>&1
CS_STREAM_INFO=1 CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_prof.so python tools/probe_c5phase.py 100000000 3 > gpurun_out/r06/c3_phases.txt 2>&1
R=$PWD
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/suite_trace -- python -m pytest tests -m gpu -q > gpurun_out/r06/t4_suite_traced.log 2>&1
find gpurun_out/suite_trace -name "*kernel_trace.csv" -delete
find gpurun_out/suite_trace -name "*.db" -delete
python tools/kernel_coverage.py gpurun_out/suite_trace > gpurun_out/r06/kernel_coverage.txt 2>&1
du -sh gpurun_out/suite_trace; tail -3 gpurun_out/r06/t4_suite_traced.log; tail -2 gpurun_out/r06/t4_pool.log; tail -2 gpurun_out/r06/kernel_coverage.txt
