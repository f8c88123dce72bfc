"""PointConv at the shapes of the N=8192, B=8 pipeline: device time per call of mcp_pointconv_agg, of the Linear + LeakyReLU behind it
(mcp_linear, or the library GEMM + activation where the model uses those) and of the one-launch form mcp_pointconv_linear where it is
built for the shape (A/B two builds with MCP_HIP_LIB=<other .so>)."""
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
F = torch.nn.functional
for name, b, n, s, d, co in (("enc level0", 16, 8192, 8192, 32, 32), ("enc level1", 16, 8192, 2048, 64, 64), ("enc level2", 16, 2048, 512, 128, 128),
                             ("enc level3", 16, 512, 256, 256, 256), ("enc level4", 16, 256, 64, 512, 256), ("refine level1 (sampled)", 24, 8192, 2048, 64, 64),
                             ("refine level1 (all)", 24, 8192, 8192, 64, 64)):
    xyz = torch.randn(b, n, 3, device=dev) * 10
    # neighbours that are actually near (sorted-ish gathers, as in the pipeline): indices around the query's own index
    base = (torch.arange(s, device=dev) * (n // s)).view(1, s, 1)
    idx = ((base + torch.randint(-64, 64, (b, s, 32), device=dev)) % n).int().contiguous()
    new_xyz = xyz[:, :: n // s].contiguous()
    f = torch.randn(b, n, d, device=dev)
    us = t(lambda: be.pointconv_agg(xyz, new_xyz, f, idx, *wn))
    out = be.pointconv_agg(xyz, new_xyz, f, idx, *wn)
    wl, bl = torch.randn(co, (d + 3) * 8, device=dev) * ((d + 3) * 8) ** -0.5, torch.randn(co, device=dev) * 0.1
    if be.linear_supported(out, co):
        pk = be.linear_pack(wl, bl, [out.shape[-1]])
        lin = lambda: be.linear(out, wl, bl, 0.1, packed=pk)
    else:
        lin = lambda: F.leaky_relu(F.linear(out, wl, bl), 0.1)
    us_lin = t(lin)
    line = f"{name:26s} B={b} N={n} S={s} D={d}->{co}: agg {us:7.1f} us + linear {us_lin:6.1f} us"
    if be.pointconv_linear_supported(d, co):
        pk2 = be.pointconv_linear_pack(wl, bl)
        us_f = t(lambda: be.pointconv_linear(xyz, new_xyz, f, idx, *wn, wl, bl, 0.1, packed=pk2))
        err = float((be.pointconv_linear(xyz, new_xyz, f, idx, *wn, wl, bl, 0.1, packed=pk2) - lin()).abs().max())
        line += f" = {us + us_lin:7.1f} us;  one launch {us_f:7.1f} us  (max |diff| {err:.2e})"
    print(line + f"   checksum {float(out.double().sum()):.6f}", flush=True)
