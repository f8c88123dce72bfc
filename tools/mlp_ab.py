"""The fused two-layer MLP kernel against the library chain (Linear, PReLU, Linear, add) at the shapes of Multi_Frame_Att."""
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
for rows, c, h, co in ((49152, 64, 256, 64), (49152, 64, 256, 3), (24576, 128, 512, 128), (24576, 128, 512, 3), (12288, 256, 1024, 3)):
    x = torch.randn(rows, c, device="cuda"); res = torch.randn(rows, co, device="cuda")
    w1, b1 = torch.randn(h, c, device="cuda") / c ** 0.5, torch.randn(h, device="cuda") * 0.1
    w2, b2 = torch.randn(co, h, device="cuda") / h ** 0.5, torch.randn(co, device="cuda") * 0.1
    a = torch.tensor([0.25], device="cuda")
    if not be.mlp2_supported(c, h, co):
        print(f"rows {rows:6d} {c:3d}->{h:4d}->{co:3d}: unsupported"); continue
    pk = be.mlp2_pack(w1, b1, w2, b2)
    lib = lambda: F.linear(F.prelu(F.linear(x, w1, b1), a), w2, b2) + res
    fused = lambda: be.mlp2(x, w1, b1, w2, b2, 0.25, res=res, packed=pk)
    gf = 2.0 * rows * (c * h + h * co)
    tl, tf = t(lib), t(fused)
    print(f"rows {rows:6d} {c:3d}->{h:4d}->{co:3d}: library chain {tl:7.1f} us   fused {tf:7.1f} us ({gf / tf / 1e6:6.1f} TFLOP/s)")
