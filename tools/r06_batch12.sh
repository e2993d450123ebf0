#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu -k "growing" > gpurun_out/r06/t14.log 2>&1
tail -4 gpurun_out/r06/t14.log
python tools/cliff_scan.py 2>/dev/null | head -8
