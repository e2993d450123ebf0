#!/bin/bash
# round 6, GPU batch 1: the new tests, the nt A/B on ONE box, the new bench line
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py tests/test_gpu_dist2.py -q -m gpu -x > gpurun_out/r06/t1.log 2>&1
echo "tests rc=$?" >> gpurun_out/r06/t1.log
for v in base nts ntb base2 nts2; do
  lib=/root/repo/custrings_amd/libcustrings_amd.so
  case $v in nts*) lib=/root/repo/custrings_amd/libcustrings_amd_nts.so;; ntb*) lib=/root/repo/custrings_amd/libcustrings_amd_ntb.so;; esac
  CS_LIB_PATH=$lib python bench.py --steps 20 --warmup 3 --no-cpu --cold-steps 0 --concurrent-steps 0 --no-box > gpurun_out/r06/ab_$v.json 2> gpurun_out/r06/ab_$v.err
done
python bench.py --steps 20 --warmup 3 > gpurun_out/r06/bench1.json 2> gpurun_out/r06/bench1.err
tail -5 gpurun_out/r06/t1.log
