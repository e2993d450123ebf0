#!/bin/bash
# GPU box: the replace stream kernel with phases switched off (CS_TILE_DEBUG bits: 1 no scan, 2 no assembly,
# 4 no flush, 8 no look-back, 32 generic scan, 64 late first poll, 128 single poll, 256 static round-robin) -- ceilings per phase
# (CS_TILE_DEBUG is live in the profiling build only -- make -C custrings_amd/csrc prof -- so the probes run on it: the product kernels
# have the switches compiled out)
export CS_LIB_PATH=${CS_LIB_PATH:-$PWD/custrings_amd/libcustrings_amd_prof.so}
for d in 0 8 1 9 2 4 12 15 64 256; do
  CS_TILE_DEBUG=$d python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids | tail -1
done
