"""One launch of cross_kernel per shape of the step (D = 64: 40 x 2048 points, D = 128: 48 x 512, D = 256: 16 x 256), for counter passes:
    rocprofv3 --pmc ... -- python3 tools/cross_pmc.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops
be = ops.backend()
dev = "cuda"
w = lambda *s: torch.randn(*s, device=dev) * 0.1
for b, n, d in ((40, 2048, 64), (48, 512, 128), (16, 256, 256)):
    xyz1, xyz2 = torch.randn(b, n, 3, device=dev) * 10, torch.randn(b, n, 3, device=dev) * 10
    f1, f2 = torch.randn(b, n, d, device=dev), torch.randn(b, n, d, device=dev)
    base = torch.arange(n, device=dev).view(1, n, 1)
    idx = ((base + torch.randint(-64, 64, (b, n, 32), device=dev)) % n).int().contiguous()
    pk = be.cross_pack(w(d, 3), w(d), w(d, d), w(d))
    for _ in range(3):
        be.cross_volume(xyz1, xyz2, f1, f2, idx, pk)
    torch.cuda.synchronize()
