#!/bin/bash
# GPU box: headline kernels, plain and with the per-phase cycle counters of the `make prof` build
python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids
python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids
if [ -f custrings_amd/libcustrings_amd_prof.so ]; then
  CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_prof.so python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids | tail -2
  CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_prof.so python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids | tail -2
fi
