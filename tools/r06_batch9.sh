#!/bin/bash
mkdir -p gpurun_out/r06
for v in base tl base2 tl2; do
  lib=/root/repo/custrings_amd/libcustrings_amd.so
  case $v in tl*) lib=/root/repo/custrings_amd/libcustrings_amd_tl.so;; esac
  CS_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu --cold-steps 0 --concurrent-steps 0 --no-box > gpurun_out/r06/ab9_$v.json 2> gpurun_out/r06/ab9_$v.err
done
CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_tl.so python tools/probe_c5regex.py 62500000 1000 > gpurun_out/r06/c5regex_tl.jsonl 2> gpurun_out/r06/c5regex_tl.err
CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_tl.so python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_round2.py -q -m gpu -x > gpurun_out/r06/t9_tl.log 2>&1
tail -2 gpurun_out/r06/t9_tl.log
