"""Timings at the BASELINE configs[3]/[4] shapes (parity is covered by tests; these are records for DESIGN.md)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops, pointnet2_utils as pu
be = ops.backend()
def t(fn, reps=3):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e))
    return statistics.median(v)
g = torch.Generator().manual_seed(0)
def cloud(b, n, ext): return ((torch.rand(b, n, 3, generator=g) * 2 - 1) * torch.tensor(ext)).cuda().contiguous()
x16 = cloud(8, 16384, [50.0, 50.0, 4.0]); c16 = x16[:, :2048].contiguous()
for r, ns in ((0.5, 16), (1.0, 16), (2.0, 8), (4.0, 8)):
    print(f"config4 ball_query N=16384 M=2048 r={r} nsample={ns}: {t(lambda: pu.ball_query(r, ns, x16, c16)):.3f} ms")
print(f"config4 fps 8x16384->2048: {t(lambda: be.fps(x16, 2048)):.3f} ms;  knn 8x16384x16384 k32: {t(lambda: be.knn(x16, x16, 32)):.3f} ms")
x64 = cloud(8, 65536, [80.0, 80.0, 6.0]); q64 = x64[:, :2048].contiguous()
print(f"config5 fps 8x65536->2048 (tiled kernel): {t(lambda: be.fps(x64, 2048), 2):.3f} ms")
print(f"config5 knn Q=2048 N=65536 k32: {t(lambda: be.knn(q64, x64, 32)):.3f} ms   brute: {t(lambda: be.knn_bruteforce(q64, x64, 32)):.3f} ms")
print(f"config5 knn Q=65536 N=65536 k32: {t(lambda: be.knn(x64, x64, 32), 2):.3f} ms   brute: {t(lambda: be.knn_bruteforce(x64, x64, 32), 1):.3f} ms")
