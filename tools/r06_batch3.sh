#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu -k pool > gpurun_out/r06/t3_pool.log 2>&1
for v in new oldntb new2 oldntb2; do
  lib=/root/repo/custrings_amd/libcustrings_amd.so
  case $v in oldntb*) lib=/root/repo/custrings_amd/libcustrings_amd_ntb.so;; esac
  CS_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu --cold-steps 0 --concurrent-steps 0 --no-box > gpurun_out/r06/ab3_$v.json 2> gpurun_out/r06/ab3_$v.err
done
python tools/probe_c5regex.py > gpurun_out/r06/c5regex.jsonl 2> gpurun_out/r06/c5regex.err
python -m pytest tests -q -m gpu -x > gpurun_out/r06/t3_all.log 2>&1
echo "all rc=$?" >> gpurun_out/r06/t3_all.log
tail -3 gpurun_out/r06/t3_all.log; grep -h "pool:" gpurun_out/r06/t3_pool.log | head
