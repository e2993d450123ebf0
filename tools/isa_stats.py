#!/usr/bin/env python3
"""Per-kernel instruction mix from a hipcc --save-temps .s file (dev tool)."""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"\n\t\.globl\t|\n\t\.protected\t", s)
seen = set()
for m in re.finditer(r"^(_Z\w+):\s*; @\1\n(.*?)\n\t\.end_amdhsa_kernel", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name or name in seen:
        continue
    seen.add(name)
    c = lambda p: len(re.findall(p, body))
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
    sg = re.search(r"\.amdhsa_next_free_sgpr (\d+)", body)
    print("%s\n  vgpr %s sgpr %s | flat_ld %d flat_st %d | ds_rd %d ds_wr %d | glb_ld %d glb_st %d | scratch %d | buffer %d | insts ~%d" % (
        name, vg and vg.group(1), sg and sg.group(1), c(r"\bflat_load"), c(r"\bflat_store"), c(r"\bds_read"), c(r"\bds_write"),
        c(r"\bglobal_load"), c(r"\bglobal_store"), c(r"\bscratch_"), c(r"\bbuffer_"), c(r"\n\t[vs]_")))
