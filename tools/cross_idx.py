import os, sys, statistics, torch
sys.path.insert(0, '/root/repo')
from mocopci_amd import ops
be = ops.backend(); dev="cuda"
def t(fn, reps=9):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e) * 1e3)
    return statistics.median(v)
w = lambda *s: torch.randn(*s, device=dev) * 0.1
for B, N, D in ((16, 2048, 64), (40, 2048, 64), (16, 512, 128), (48, 512, 128)):
    xyz = torch.randn(B, N, 3, device=dev); f1 = torch.randn(B, N, D, device=dev); f2 = torch.randn(B, N, D, device=dev)
    pk = be.cross_pack(w(D, 3), w(D), w(D, D), w(D))
    rnd = torch.randint(0, N, (B, N, 32), device=dev, dtype=torch.int32)
    seq = ((torch.arange(N, device=dev)[:, None] + torch.arange(32, device=dev)) % N).int().expand(B, N, 32).contiguous()
    same = torch.zeros(B, N, 32, device=dev, dtype=torch.int32)
    flops = B * N * 32 * (D * D + 4 * D) * 2
    for name, idx in (("random", rnd), ("sequential", seq), ("all-zero", same)):
        us = t(lambda: be.cross_volume(xyz, xyz, f1, f2, idx, pk))
        print(f"B={B} N={N} D={D} idx {name:10s}: {us:8.1f} us  {flops/us/1e6:6.1f} TFLOP/s")
