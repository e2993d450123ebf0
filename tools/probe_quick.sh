#!/bin/bash
# GPU box: headline kernels under a list of CS_TILE_DEBUG values (default 0), then the replace/split parity tests
for d in ${DBG:-0}; do
  CS_TILE_DEBUG=$d python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids | tail -1
done
python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids | tail -1
python -m pytest tests -m gpu -x -q -k "${TESTS:-replace or split}" 2>&1 | tail -3
