#!/bin/bash
# GPU box: headline kernels under a list of CS_TILE_DEBUG values (default 0), then the replace/split parity tests
# (CS_TILE_DEBUG is live in the profiling build only -- make -C custrings_amd/csrc prof -- so the probes run on it: the product kernels
# have the switches compiled out)
export CS_LIB_PATH=${CS_LIB_PATH:-$PWD/custrings_amd/libcustrings_amd_prof.so}
for d in ${DBG:-0}; do
  CS_TILE_DEBUG=$d python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids | tail -1
done
python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids | tail -1
python -m pytest tests -m gpu -x -q -k "${TESTS:-replace or split}" 2>&1 | tail -3
