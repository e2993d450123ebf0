import sys, os
sys.path.insert(0, '/root/repo')
import torch
import tools.bench_ops as B
col = B.synth(3, 100_000_000)
for i in range(3):
    r = col.findall(sys.argv[1]); print(len(r)); del r
torch.cuda.synchronize()
