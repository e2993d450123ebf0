#!/bin/bash
mkdir -p gpurun_out/r06
python -m pytest tests/test_gpu_round6.py -q -m gpu -x -k "pieces or long" > gpurun_out/r06/t11_pieces.log 2>&1
tail -15 gpurun_out/r06/t11_pieces.log
python tools/probe_c5regex.py 62500000 1000 > gpurun_out/r06/c5regex_pieces.jsonl 2> gpurun_out/r06/c5regex_pieces.err
