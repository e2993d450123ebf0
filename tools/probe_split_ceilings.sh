#!/bin/bash
# GPU box: split emit with phases switched off (CS_SPLIT_DEBUG bits: 1 no offset stores, 2 no chars stores, 4 no assembly, 8 no column loop)
for d in 0 1 2 3 4 7 8; do
  echo -n "CS_SPLIT_DEBUG=$d: "; CS_SPLIT_DEBUG=$d python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids | tail -1
done
