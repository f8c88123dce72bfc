"""The fused Linear kernel (mcp_linear) against the library chain (cat, F.linear, activation, add) at the caller graph's shapes."""
import os, sys, statistics, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mocopci_amd import ops
be = ops.backend(); torch.manual_seed(0)
def t(fn, reps=7):
    fn(); torch.cuda.synchronize(); v = []
    for _ in range(reps):
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); v.append(s.elapsed_time(e) * 1e3)
    return statistics.median(v)
cases = [  # rows, piece widths, n, slope, residual
    (196608, (536,), 64, 0.1, False), (131072, (536,), 64, 0.1, False), (131072, (280,), 32, 0.1, False), (196608, (64,), 192, 1.0, False),
    (196608, (32,), 64, 0.1, False), (131072, (32,), 64, 0.1, False), (131072, (64,), 64, 0.1, False), (196608, (64,), 32, 0.0, False),
    (32768, (64, 64, 64), 64, 1.0, False), (8192, (128, 128, 128), 128, 1.0, False), (81920, (64,), 64, 1.0, False),
    (49152, (64,), 64, 1.0, True), (49152, (64,), 128, 1.0, False), (24576, (128,), 128, 1.0, True), (24576, (128,), 256, 1.0, False),
    (24576, (128,), 512, 0.25, False), (4096, (128,), 256, 1.0, False), (2048, (256,), 256, 1.0, False), (4096, (256, 256, 64), 256, 1.0, False),
    (32768, (64,), 128, 0.1, False), (16384, (64,), 64, 1.0, False), (4096, (2072,), 256, 0.1, False),
    (12288, (256,), 256, 1.0, False), (8192, (256,), 256, 1.0, False), (8192, (384,), 128, 1.0, False), (12288, (256,), 192, 1.0, False),
    (8192, (256,), 1536, 1.0, False), (12288, (256,), 1024, 1.0, False), (8192, (256,), 768, 1.0, False), (12288, (256,), 1024, 0.25, False),
    (24576, (128,), 512, 0.25, False), (2048, (256,), 512, 1.0, False), (4096, (256,), 512, 1.0, False), (49152, (64,), 256, 0.25, False),
]
tot_l = tot_f = 0.0
for rows, ks, n, slope, with_res in cases:
    xs = [torch.randn(rows, k, device="cuda") for k in ks]
    w, b = torch.randn(n, sum(ks), device="cuda") / sum(ks) ** 0.5, torch.randn(n, device="cuda") * 0.1
    res = torch.randn(rows, n, device="cuda") if with_res else None
    def lib():
        y = F.linear(torch.cat(xs, -1) if len(xs) > 1 else xs[0], w, b)
        if slope != 1.0: y = F.leaky_relu(y, slope)
        return y if res is None else y + res
    if not be.linear_supported(xs, n):
        print(f"rows {rows:6d} {ks} -> {n}: unsupported"); continue
    pk = be.linear_pack(w, b, list(ks))
    fused = lambda: be.linear(xs if len(xs) > 1 else xs[0], w, b, slope, res, packed=pk)
    err = (fused() - lib()).abs().max().item()
    tl, tf = t(lib), t(fused)
    tot_l += tl; tot_f += tf
    print(f"rows {rows:6d} {str(ks):16s} -> {n:3d} slope {slope:4.2f} res {int(with_res)}: library {tl:7.1f} us  fused {tf:7.1f} us  max err {err:.1e}")
print(f"sum: library {tot_l:.0f} us, fused {tot_f:.0f} us")
