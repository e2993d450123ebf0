#!/bin/bash
# round 6, GPU batch 2: round-6 tests (pool, malformed UTF-8, soak slices, long-row split), dist2 (failure agreement), C5 split, bench
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu > gpurun_out/r06/t2_round6.log 2>&1
echo "round6 rc=$?" >> gpurun_out/r06/t2_round6.log
python -m pytest tests/test_gpu_dist2.py -q -m gpu > gpurun_out/r06/t2_dist2.log 2>&1
echo "dist2 rc=$?" >> gpurun_out/r06/t2_dist2.log
python tools/probe_c5split.py > gpurun_out/r06/c5split.txt 2>&1
CS_SPLIT_NO_LONG_WALK=1 python tools/probe_c5split.py > gpurun_out/r06/c5split_old.txt 2>&1
python bench.py --steps 20 --warmup 3 --no-cpu > gpurun_out/r06/bench2.json 2> gpurun_out/r06/bench2.err
tail -4 gpurun_out/r06/t2_round6.log; tail -3 gpurun_out/r06/t2_dist2.log; cat gpurun_out/r06/c5split.txt gpurun_out/r06/c5split_old.txt | grep -v amdgpu.ids
