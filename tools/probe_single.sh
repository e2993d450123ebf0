#!/bin/bash
# GPU box: the single-pass split kernel with phases switched off (CS_SPLIT_DEBUG bits, cs_split1.hip) and ticket chunk sizes
for d in ${@:-0 1 2 4 6 7 8}; do
  CS_SPLIT_DEBUG=$d timeout 120 python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids | grep "single pass\|^split" | tail -2
done
for c in ${CHUNKS:-0 6}; do
  echo "chunk_log2=$c"; CS_SPLIT1_CHUNK=$c timeout 120 python tools/probe_replace.py 100000000 split 2>&1 | grep -v amdgpu.ids | grep "single pass\|^split" | tail -2
done
