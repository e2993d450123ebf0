for d in 0 64 8; do CS_TILE_DEBUG=$d python tools/probe_replace.py 100000000 replace; done 2>&1 | grep -v amdgpu.ids
for b in 1 2; do CS_STREAM_BLOCKS_PER_CU=$b python tools/probe_replace.py 100000000 replace; done 2>&1 | grep -v amdgpu.ids
