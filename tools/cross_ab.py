"""cross_kernel at the shapes of the N=8192, B=8 pipeline: device time per call (A/B two builds with MCP_HIP_LIB=<other .so>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops
be = ops.backend()
dev = "cuda"
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
torch.manual_seed(0)
w = lambda *s: torch.randn(*s, device=dev) * 0.1
for name, b, n, d in (("level 1 (D=64)", 40, 2048, 64), ("level 2 (D=128)", 48, 512, 128), ("level 3 (D=256)", 16, 256, 256)):
    xyz1, xyz2 = torch.randn(b, n, 3, device=dev) * 10, torch.randn(b, n, 3, device=dev) * 10
    f1, f2 = torch.randn(b, n, d, device=dev), torch.randn(b, n, d, device=dev)
    base = torch.arange(n, device=dev).view(1, n, 1)
    idx = ((base + torch.randint(-64, 64, (b, n, 32), device=dev)) % n).int().contiguous()
    pk = be.cross_pack(w(d, 3), w(d), w(d, d) * (8.0 / d ** 0.5), w(d))
    us = t(lambda: be.cross_volume(xyz1, xyz2, f1, f2, idx, pk))
    out = be.cross_volume(xyz1, xyz2, f1, f2, idx, pk)
    flop = b * n * 32 * (8 * d + 2 * d * d)
    print(f"{name:18s} B={b} N={n}: {us:7.1f} us  {flop / us / 1e6:7.1f} TFLOP/s   checksum {float(out.double().sum()):.6f}", flush=True)
