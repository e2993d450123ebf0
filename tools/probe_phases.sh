for d in 0 32 1 8 9; do CS_TILE_DEBUG=$d python tools/probe_replace.py 100000000 replace; done 2>&1 | grep -v amdgpu.ids
