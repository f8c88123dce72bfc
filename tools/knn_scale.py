import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops
be = ops.backend()
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e) * 1e3)
    return statistics.median(v)
g = torch.Generator().manual_seed(1)
for b in (1, 4, 16, 64):
    x = ((torch.rand(b, 8192, 3, generator=g) * 2 - 1) * torch.tensor([40.0, 40.0, 3.0])).cuda().contiguous()
    us = t(lambda: be.knn(x, x, 32)); ub = t(lambda: be.knn_bruteforce(x, x, 32))
    print(f"B={b:3d}: pruned {us:8.1f} us ({us/b:6.1f} us/batch)   brute {ub:8.1f} us ({ub/b:6.1f} us/batch)")
