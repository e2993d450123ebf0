#!/usr/bin/env python3
"""Every kernel in libcustrings_amd.so's code object -> registers, spills, and how often the -m gpu suite launched it
("which instantiations are reachable, and which are tested, is not recorded anywhere").

  1. on the GPU box:   rocprofv3 --kernel-trace --stats -d gpurun_out/suite_trace -- python -m pytest tests -m gpu -q
  2. anywhere:         python tools/kernel_coverage.py gpurun_out/suite_trace > profiles/r06/kernel_coverage.txt

The code object is read from the built objects (custrings_amd/csrc/_build/*.o: llvm-objcopy + clang-offload-bundler +
llvm-readelf --notes); the launches from every *kernel_stats.csv / *kernel_trace.csv under the trace directory (child
processes of the suite write their own).  A kernel the suite never launched is either unreachable or untested: the table
says which kernels those are, so that they can be dropped or get a test."""
import csv
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(anonymous namespace)::", "") for o in out[:len(names)]]


def code_object_kernels():
    rows = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "custrings_amd", "csrc", "_build", "*.o"))):
        with tempfile.TemporaryDirectory() as t:
            fat, dev = os.path.join(t, "fat.bin"), os.path.join(t, "dev.o")
            r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], capture_output=True)
            if r.returncode != 0 or not os.path.exists(fat):
                continue
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev], check=True, capture_output=True)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", dev], capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda k: int((re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "0"])[1])
            rows.append({"file": os.path.basename(obj)[:-2], "mangled": name.group(1), "vgpr": g("vgpr_count"), "spill": g("vgpr_spill_count"),
                         "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size")})
    for r, d in zip(rows, demangle([r["mangled"] for r in rows])):
        r["name"] = d
    return rows


def launches(trace_dir):
    """kernel name (demangled, as rocprofv3 prints it) -> calls"""
    calls = {}
    for path in glob.glob(os.path.join(trace_dir, "**", "*kernel_stats.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                n = row.get("Name") or row.get("KernelName") or ""
                c = int(float(row.get("Calls") or row.get("Count") or 0))
                calls[n] = calls.get(n, 0) + c
    if not calls:
        for path in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    n = row.get("Kernel_Name") or row.get("Name") or ""
                    calls[n] = calls.get(n, 0) + 1
    return calls


def norm(s):
    return re.sub(r"\s+", "", s.replace("(anonymous namespace)::", "").replace("[clone .kd]", "")).replace("void", "", 1) if s else s


def main():
    trace = sys.argv[1] if len(sys.argv) > 1 else None
    kernels = code_object_kernels()
    calls = {norm(k): v for k, v in launches(trace).items()} if trace else {}
    print("# %d kernels in the code object; launches: %s" % (len(kernels), trace or "(no trace given)"))
    print("# %-8s %6s %5s %5s %7s  %s" % ("file", "calls", "vgpr", "spill", "scratch", "kernel"))
    never = 0
    for r in sorted(kernels, key=lambda r: (r["file"], r["name"])):
        key = norm(r["name"])
        c = calls.get(key)
        if c is None:  # rocprofv3 prints the name without the argument list
            c = calls.get(norm(r["name"].split("(")[0]), 0)
        never += c == 0
        print("%-10s %6d %5d %5d %7d  %s" % (r["file"], c, r["vgpr"], r["spill"], r["scratch"], r["name"][:200]))
    print("# never launched by the suite: %d of %d" % (never, len(kernels)))


if __name__ == "__main__":
    main()
