#!/usr/bin/env python3
"""VALU pipe mix of a kernel (dev tool): python tools/isa_mix.py cs_split k_split_emit3 [--blocks]

Disassembles custrings_amd/csrc/_build/<obj>.o's gfx950 code object and classifies every vector ALU
instruction by the issue class tools/ubench/valu_rate.hip measured on MI355X (profiles/r04/valu_rate.txt):

  simple  (two waves of a SIMD issue them concurrently: 0.43-0.47 wave-instructions per clock and SIMD):
          v_add/sub/subrev_u32, v_and/or/xor/not_b32, v_mov_b32, v_lshrrev_b32, v_ashrrev_i32, v_fma/add/mul_f32
  complex (one at a time: 0.24 per clock and SIMD): everything else -- v_lshlrev_b32, 64-bit shifts, every
          three-operand VOP3 (bfe, bfi, perm, alignbyte/bit, add3, lshl_or, and_or, lshl_add, min3 ...),
          v_cmp*, v_cndmask, min/max, DPP, SDWA, v_pk_*, bcnt, ffbl, mbcnt, mul, dot4, sad, readlane

Static counts (per basic block with --blocks: the hot loops are what matters).  Under the two-pipe reading of the
ubench a SIMD needs about 4.1 * max(complex, (complex + simple) / 2) cycles for a block.
"""
import re
import subprocess
import sys
import tempfile
import os

LLVM = "/opt/rocm/lib/llvm/bin/"
SIMPLE = re.compile(r"^v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|not_b32|mov_b32|lshrrev_b32|ashrrev_i32|fma_f32|add_f32|mul_f32|mov_b64)(_e32|_e64)?$")


def classify(mn, ops):
    if not mn.startswith("v_"):
        return None
    if "dpp" in mn or "sdwa" in mn or "row_" in ops or "quad_perm" in ops or "sel:" in ops:
        return "complex"
    return "simple" if SIMPLE.match(mn) else "complex"


def main():
    obj, flt = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    o = os.path.join(root, "custrings_amd/csrc/_build/%s.o" % obj)
    with tempfile.TemporaryDirectory() as t:
        subprocess.run([LLVM + "llvm-objcopy", "--dump-section", ".hip_fatbin=%s/fat.bin" % t, o], check=True)
        subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=%s/fat.bin" % t,
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=%s/dev.o" % t], check=True)
        dis = subprocess.run([LLVM + "llvm-objdump", "-d", "--no-show-raw-insn", "-C", "%s/dev.o" % t], capture_output=True, text=True).stdout
    cur = None
    kernels = {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1)
            kernels.setdefault(cur, [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        body = line.strip().split("//")[0].strip()
        if not body:
            continue
        kernels[cur].append(body)
    # llvm-objdump labels basic blocks as separate symbols "<L123>"?  No: branch targets appear as addresses only, so
    # blocks are cut at branch instructions and at branch targets taken from the operands.
    for name, ins in kernels.items():
        if flt not in name or not ins:
            continue
        tot = {"simple": 0, "complex": 0, "salu": 0, "lds": 0, "vmem": 0, "other": 0}
        top = {}
        for b in ins:
            parts = b.split(None, 1)
            mn, ops = parts[0], parts[1] if len(parts) > 1 else ""
            c = classify(mn, ops)
            if c:
                tot[c] += 1
                if c == "complex":
                    top[mn] = top.get(mn, 0) + 1
            elif mn.startswith("s_"):
                tot["salu"] += 1
            elif mn.startswith("ds_"):
                tot["lds"] += 1
            elif mn.startswith(("global_", "flat_", "buffer_", "scratch_")):
                tot["vmem"] += 1
            else:
                tot["other"] += 1
        valu = tot["simple"] + tot["complex"]
        print("%s\n  VALU %d: simple %d (%.0f %%), complex %d | SALU %d | LDS %d | VMEM %d" % (
            name[:150], valu, tot["simple"], 100.0 * tot["simple"] / max(valu, 1), tot["complex"], tot["salu"], tot["lds"], tot["vmem"]))
        print("  complex by mnemonic: " + ", ".join("%s %d" % kv for kv in sorted(top.items(), key=lambda kv: -kv[1])[:24]))


if __name__ == "__main__":
    main()
