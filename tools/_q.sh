ulimit -c 0; mkdir -p gpurun_out/r04
( for op in ipv4b dense; do CS_STREAM_INFO=1 CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_prof.so python tools/probe_op.py $op 2>&1 | grep -v "look-back" | tail -3; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r04/q1.txt
