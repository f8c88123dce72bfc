"""Pruned vs exhaustive KNN at the decoder's small levels (48 x 2048 x 2048, 48 x 512 x 512)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, synth
be = ops.backend()
def t(fn, reps=7):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e) * 1e3)
    return statistics.median(v)
x1, x2, _ = synth.make_batch(2, 24, 8192, device="cuda")
for n in (2048, 512):
    a = x1.transpose(1, 2)[:, :n].contiguous(); b = (x2.transpose(1, 2)[:, :n] ).contiguous()
    a = torch.cat([a, b]).contiguous(); b = torch.cat([b, a[:24]]).contiguous()
    for k in (16, 32, 3):
        print(f"n={n} k={k}: brute {t(lambda: be.knn_bruteforce(a, b, k)):8.1f} us", end="")
        if k > 4:
            be.PRUNE_MIN_REFS = 64; be.PRUNE_MIN_QUERIES = 64
            tot = t(lambda: be.knn(a.clone(), b.clone(), k))      # clones defeat the sorted-cloud cache: full cost
            cached = t(lambda: be.knn(a, b, k))
            print(f"   pruned incl. 2 builds {tot:8.1f} us   search only {cached:8.1f} us", end="")
        print()
    print(f"n={n} build_cloud {t(lambda: be._sorted_cloud(a.clone())):8.1f} us (incl clone)")
for n in (8192, 16384, 1000):
    c = (torch.rand(24, n, 3, device="cuda") * 80 - 40).contiguous()
    print(f"n={n} build_cloud x24 {t(lambda: be._sorted_cloud(c.clone())):8.1f} us (incl clone)")
