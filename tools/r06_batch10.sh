#!/bin/bash
mkdir -p gpurun_out/r06
for v in base al base2 al2; do
  lib=/root/repo/custrings_amd/libcustrings_amd.so
  case $v in al*) lib=/root/repo/custrings_amd/libcustrings_amd_al.so;; esac
  CS_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu --cold-steps 0 --concurrent-steps 0 --no-box > gpurun_out/r06/ab10_$v.json 2> gpurun_out/r06/ab10_$v.err
done
CS_LIB_PATH=/root/repo/custrings_amd/libcustrings_amd_al.so python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py -q -m gpu -k "split" > gpurun_out/r06/t10_al.log 2>&1
tail -2 gpurun_out/r06/t10_al.log
