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
for b in (1, 16, 256):
    for n, m in ((64, 33), (128, 65), (1024, 513), (2048, 513), (4096, 513), (8192, 513), (16384, 513)):
        x = (torch.rand(b, n, 3, device="cuda") * 80 - 40).contiguous()
        us = t(lambda: be.fps(x, m))
        print(f"B={b:4d} N={n:6d} M={m:4d}: {us:9.1f} us  -> {us/(m-1)*1000:7.1f} ns/iter")
