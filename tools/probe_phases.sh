python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids
CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_prof.so python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids | tail -2
