python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids
for w in 5 6; do CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_s$w.so python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids; done
