"""pointconv_agg at the shapes of the N=8192, B=8 pipeline: device time per call (A/B two builds with MCP_HIP_LIB=<other .so>)."""
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
w = lambda *s: torch.randn(*s, device=dev) * 0.3
wn = [w(8, 3), w(8), w(8, 8), w(8), w(8, 8), w(8)]
for name, b, n, s, d in (("enc level0", 16, 8192, 8192, 32), ("enc level1", 16, 8192, 2048, 32), ("enc level2", 16, 2048, 512, 64), ("enc level3", 16, 512, 256, 128),
                         ("enc level4", 16, 256, 64, 256), ("refine level1 (sampled)", 24, 8192, 2048, 64), ("refine level1 (all)", 24, 8192, 8192, 64)):
    xyz = torch.randn(b, n, 3, device=dev) * 10
    # neighbours that are actually near (sorted-ish gathers, as in the pipeline): indices around the query's own index
    base = (torch.arange(s, device=dev) * (n // s)).view(1, s, 1)
    idx = ((base + torch.randint(-64, 64, (b, s, 32), device=dev)) % n).int().contiguous()
    new_xyz = xyz[:, :: n // s].contiguous()
    f = torch.randn(b, n, d, device=dev)
    us = t(lambda: be.pointconv_agg(xyz, new_xyz, f, idx, *wn))
    out = be.pointconv_agg(xyz, new_xyz, f, idx, *wn)
    print(f"{name:26s} B={b} N={n} S={s} D={d}: {us:7.1f} us   checksum {float(out.double().sum()):.6f}", flush=True)
