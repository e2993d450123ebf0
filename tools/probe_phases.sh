python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids
CS_TDFA_GLOBAL_TABLE=1 python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids
CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_w4.so python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids
CS_TDFA_GLOBAL_TABLE=1 CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_w4.so python tools/probe_replace.py 100000000 replace 2>&1 | grep -v amdgpu.ids
