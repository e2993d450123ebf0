#!/bin/bash
# VGPRs, spills and LDS of the kernels in a built object (dev tool): bash tools/codeobj_notes.sh cs_regex [name filter]
O=custrings_amd/csrc/_build/$1.o
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $O
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.o | python3 -c '
import sys, re, subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ""
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if not name: continue
    try:
        dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name.group(1)], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name.group(1)
    if flt not in dem: continue
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
    print("%-110s vgpr %s spill %s sgpr %s lds %s scratch %s" % (dem.replace("(anonymous namespace)::", "")[:110], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
' "$2"
rm -rf $T
