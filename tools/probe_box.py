#!/usr/bin/env python3
"""Does a box's emit time follow its raw HBM rates?  fill / copy / read of 4 GiB with torch, then the headline kernels
(dev tool; one line per run, compare across gpurun boxes): python tools/probe_box.py"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

n = 1 << 30  # int32 elements: 4 GiB
x = torch.empty(n, dtype=torch.int32, device="cuda")
y = torch.empty(n, dtype=torch.int32, device="cuda")


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


fill = 4 * n / t(lambda: x.fill_(1)) / 1e12
copy = 8 * n / t(lambda: y.copy_(x)) / 1e12
read = 4 * n / t(lambda: x.sum()) / 1e12
del x, y
out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "3", "--cold-steps", "0"], capture_output=True, text=True).stdout.strip().splitlines()[-1]
b = json.loads(out)
k = b.get("kernels_avg_ms") or b.get("kernels")
smi = subprocess.run(["rocm-smi", "--showclocks", "--showtemp", "--showpower"], capture_output=True, text=True).stdout
keep = [l.strip() for l in smi.splitlines() if any(w in l for w in ("mclk", "sclk", "fclk", "Temperature (Sensor mem", "Temperature (Sensor junction", "Average Graphics Package Power", "Current Socket Graphics Package Power"))]
print(json.dumps({"fill_TBps": round(fill, 3), "copy_TBps": round(copy, 3), "read_TBps": round(read, 3), "ms_per_step": b["ms_per_step"],
                  "kernels": {n_: v["avg_ms"] for n_, v in k.items()}, "smi": keep}))
