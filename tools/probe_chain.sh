# replace_re(IPv4) on the 100M-row column with phases of the stream kernel switched off (CS_TILE_DEBUG; results are wrong then)
# (CS_TILE_DEBUG is live in the profiling build only -- make -C custrings_amd/csrc prof -- so the probes run on it: the product kernels
# have the switches compiled out)
export CS_LIB_PATH=${CS_LIB_PATH:-$PWD/custrings_amd/libcustrings_amd_prof.so}
for d in 0 8192 8 4 12 1 13 256 264 268 269; do CS_TILE_DEBUG=$d python tools/probe_replace.py 100000000 replace 2>&1 | tail -1; done
