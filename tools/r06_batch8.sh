#!/bin/bash
mkdir -p gpurun_out/r06
CS_REPLACE_TRACE=1 CS_STREAM_INFO=1 CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_prof.so python tools/probe_c5phase.py 62500000 5 > gpurun_out/r06/c5_trace.txt 2>&1
CS_REPLACE_TRACE=1 CS_STREAM_INFO=1 CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_prof.so python tools/probe_c5phase.py 100000000 3 > gpurun_out/r06/c3_trace.txt 2>&1
grep -c trace gpurun_out/r06/c5_trace.txt gpurun_out/r06/c3_trace.txt
