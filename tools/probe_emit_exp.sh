#!/bin/bash
# emit3 phase switches (exp build: make EXTRA=-DCS_EMIT3_EXP OUT=../libcustrings_amd_exp.so BUILD=_build_exp), GPU box
export CS_LIB_PATH=$PWD/custrings_amd/libcustrings_amd_exp.so
for d in ${@:-0 1 2 4 8 16 32 64 3 7 39 103}; do
  CS_SPLIT_DEBUG=$d python tools/probe_replace.py 100000000 split 2>&1 | tail -1 | sed "s/^/debug=$d  /"
done
